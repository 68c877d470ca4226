#!/usr/bin/env python3
"""Where the dK/dV kernel's prologue goes (a -DDBG_STAMPS -DDBG_STAMPS_DKDV build with extra stamps 7..10 inside the prologue:
7 loads issued, 8 bias table in LDS, 9 sums cleared, 10 tile 0 staged; 2 = top of tile 0)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pacednegatives_amd import kernels as K
N, S, H = 544, 256, 12
I = H * 64
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
qkv = (torch.randn(N * S, 3 * I, device=dev, generator=g) * 0.3).to(torch.bfloat16)
dctx = torch.randn(N * S, I, device=dev, generator=g).to(torch.bfloat16)
tbl = torch.randn(H, 2 * S - 1, device=dev, generator=g) * 0.3
ctx = torch.empty(N * S, I, device=dev, dtype=torch.bfloat16)
lse = torch.empty(N * H * S, device=dev)
dqkv = torch.empty_like(qkv)
dtbl = torch.zeros(64, H, 2 * S - 1, device=dev)
q, k, v = qkv[:, :I], qkv[:, I:2 * I], qkv[:, 2 * I:]
K.enc_attn_fwd(q, k, v, ctx, lse, tbl, N, S, H, drop_p=0.1, drop_seed=1)
for _ in range(3):
    K.enc_attn_bwd(q, k, v, ctx, dctx, lse, tbl, dqkv[:, :I], dqkv[:, I:2 * I], dqkv[:, 2 * I:], dtbl, N, S, H, drop_p=0.1, drop_seed=1)
torch.cuda.synchronize()
nb = 2 * H * N
st = dqkv.view(torch.int64)[:nb, :15].cpu().double()
for name, a, b in (("entry -> loads issued", 1, 7), ("bias table staged", 7, 8), ("sums cleared", 8, 9), ("tile 0 staged (waits for its rows)", 9, 10),
                   ("fetch tile 1 + barrier", 10, 2), ("whole prologue", 1, 2)):
    d = st[:, b] - st[:, a]
    print(f"  {name:40s} mean {d.mean():8.0f} cycles  median {d.median():8.0f}  p90 {d.quantile(0.9):8.0f}")
