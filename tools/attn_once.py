"""Two forward + backward launches of the encoder attention at the benchmark shape, head-major operands as the engine
passes them (target of tools/pmc_sq_attention.sh and tools/pmc_mem_attention.sh)."""
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
p = float(os.environ.get("ATTN_DROP", "0.1"))
for _ in range(2):
    K.enc_attn_fwd(q, k, v, ctx, lse, bias, N, S, H, drop_p=p, drop_seed=3)
    K.enc_attn_bwd(q, k, v, ctx, do, lse, bias, dq, dk, dv, dbias, N, S, H, drop_p=p, drop_seed=3)
torch.cuda.synchronize()
