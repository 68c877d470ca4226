#!/usr/bin/env python3
"""Time of one encoder-attention backward (dK/dV + dQ kernels) at the benchmark shape, head-major operands, by HIP events."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pacednegatives_amd import kernels as K
N, S, H = 544, 256, 12
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
M = N * S
qkv = (torch.randn(3 * H, M, 64, device=dev, generator=g) * 0.5).to(torch.bfloat16)
bias = torch.randn(H, 2 * S - 1, device=dev, generator=g) * 0.5
ctx = torch.empty(H, M, 64, device=dev, dtype=torch.bfloat16)
lse = torch.empty(N, H, S, device=dev)
q, k, v = qkv[:H], qkv[H:2 * H], qkv[2 * H:]
do = (torch.randn(H, M, 64, device=dev, generator=g) * 0.5).to(torch.bfloat16)
dqkv = torch.empty_like(qkv)
dq, dk, dv = dqkv[:H], dqkv[H:2 * H], dqkv[2 * H:]
dbias = torch.zeros(64, H, 2 * S - 1, device=dev)
K.enc_attn_fwd(q, k, v, ctx, lse, bias, N, S, H, drop_p=0.1, drop_seed=3)
def run():
    K.enc_attn_bwd(q, k, v, ctx, do, lse, bias, dq, dk, dv, dbias, N, S, H, drop_p=0.1, drop_seed=3)
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): run()
e1.record(); torch.cuda.synchronize()
print(f"PACED_ATTN_SPLIT={os.environ.get('PACED_ATTN_SPLIT', '0')}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per backward (dK/dV + dQ)")
