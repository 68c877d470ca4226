"""Two forward + backward launches of the encoder attention at the benchmark shape (target of tools/pmc_sq_attention.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pacednegatives_amd import kernels as K
N, S, H = 544, 256, 12
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
qkv = (torch.randn(N * S, 3 * H * 64, device=dev, generator=g) * 0.5).to(torch.bfloat16)
bias = torch.randn(H, 2 * S - 1, device=dev, generator=g) * 0.5
ctx = torch.empty(N * S, H * 64, device=dev, dtype=torch.bfloat16)
lse = torch.empty(N, H, S, device=dev)
I = H * 64
q, k, v = qkv[:, :I], qkv[:, I:2 * I], qkv[:, 2 * I:]
do = (torch.randn(N * S, H * 64, device=dev, generator=g) * 0.5).to(torch.bfloat16)
dqkv = torch.empty_like(qkv)
dq, dk, dv = dqkv[:, :I], dqkv[:, I:2 * I], dqkv[:, 2 * I:]
dbias = torch.zeros(64, H, 2 * S - 1, device=dev)
for _ in range(2):
    K.enc_attn_fwd(q, k, v, ctx, lse, bias, N, S, H, drop_p=0.1, drop_seed=3)
    K.enc_attn_bwd(q, k, v, ctx, do, lse, bias, dq, dk, dv, dbias, N, S, H, drop_p=0.1, drop_seed=3)
torch.cuda.synchronize()
