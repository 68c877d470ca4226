#!/usr/bin/env python3
"""Cycle stamps from inside the encoder attention's backward kernels: where a workgroup's time goes.

Needs a build of csrc/attention.hip with the stamps compiled in, given by PACED_HIP_LIB:
  WHICH=dq   (default): -DDBG_STAMPS -DDBG_STAMPS_DQ    (the dQ kernel leaves its stamps in the bias-gradient buffer)
  WHICH=dkdv          : -DDBG_STAMPS -DDBG_STAMPS_DKDV  (the dK/dV kernel leaves them in the dQ output; the dQ kernel returns at once)
e.g.  hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -fno-gpu-rdc -DDBG_STAMPS -DDBG_STAMPS_DQ -c pacednegatives_amd/csrc/attention.hip -o attn.o
      hipcc --offload-arch=gfx950 -shared -fPIC pacednegatives_amd/lib/obj/{api,gemm,ops,mining}.o attn.o -o ab/libpaced_stamps.so
Stamp 0 / 14: 100 MHz wall clock at entry / at the end; 1: cycle counter at entry.  dQ, per key tile kt < 4: 2+3kt top of
the tile, 3+3kt after its second barrier, 4+3kt after its first 32 keys.  dK/dV: 2 top of query tile 0; per query tile
qt < 4: 3+2qt after its second barrier, 4+2qt at its end; 11 after the last tile, 12 after the dK/dV stores, 13 after the
bias gradient's flush."""
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
p = float(os.environ.get("P", 0.1))
K.enc_attn_fwd(q, k, v, ctx, lse, tbl, N, S, H, drop_p=p, drop_seed=1)
for _ in range(3):
    K.enc_attn_bwd(q, k, v, ctx, dctx, lse, tbl, dqkv[:, :I], dqkv[:, I:2 * I], dqkv[:, 2 * I:],
                   None if os.environ.get("NODBIAS") else dtbl, N, S, H, drop_p=p, drop_seed=1)       # NODBIAS=1: without the table gradient
torch.cuda.synchronize()
nb = 2 * H * N
which = os.environ.get("WHICH", "dq")
if which == "dq":
    st = dtbl.view(-1).view(torch.int64)[: nb * 15].view(nb, 15).cpu().double()
else:
    st = dqkv.view(torch.int64)[:nb, :15].cpu().double()          # row b of the dQ output holds workgroup b's stamps
wall = (st[:, 14] - st[:, 0]) * 10.0          # ns
t0 = st[:, 0].min()
print(f"{which}: workgroups {nb}; kernel span {(st[:, 14].max() - t0) * 10 / 1e3:.1f} us; workgroup lifetime mean {wall.mean() / 1e3:.2f} us "
      f"(min {wall.min() / 1e3:.2f}, max {wall.max() / 1e3:.2f})")
if which == "dq":
    names = ["prologue (entry -> top of tile 0)"]
    d = [st[:, 2] - st[:, 1]]
    for kt in range(4):
        names += [f"tile {kt}: barrier + stage + barrier", f"tile {kt}: first 32 keys"]
        d += [st[:, 3 + 3 * kt] - st[:, 2 + 3 * kt], st[:, 4 + 3 * kt] - st[:, 3 + 3 * kt]]
        if kt < 3:
            names += [f"tile {kt}: second 32 keys"]
            d += [st[:, 5 + 3 * kt] - st[:, 4 + 3 * kt]]
else:
    names = ["prologue (entry -> top of tile 0)"]
    d = [st[:, 2] - st[:, 1]]
    for qt in range(4):
        names += [f"tile {qt}: barrier + stage + barrier", f"tile {qt}: arithmetic"]
        d += [st[:, 3 + 2 * qt] - st[:, 2 + 2 * qt], st[:, 4 + 2 * qt] - st[:, 3 + 2 * qt]]
    names += ["tiles 4..7", "dK/dV stores", "bias-gradient flush"]
    d += [st[:, 11] - st[:, 10], st[:, 12] - st[:, 11], st[:, 13] - st[:, 12]]
    print(f"cycles entry -> end: mean {(st[:, 13] - st[:, 1]).mean():.0f}; implied clock {((st[:, 13] - st[:, 1]) / wall).mean():.2f} GHz")
for nme, x in zip(names, d):
    print(f"  {nme:40s} mean {x.mean():8.0f} cycles  median {x.median():8.0f}  p90 {x.quantile(0.9):8.0f}")
ev = torch.cat([torch.stack([st[:, 0], torch.ones(nb, dtype=torch.float64)], 1), torch.stack([st[:, 14], -torch.ones(nb, dtype=torch.float64)], 1)])
ev = ev[ev[:, 0].argsort()]
alive = ev[:, 1].cumsum(0)
print(f"workgroups alive at once: max {alive.max():.0f}, mean over events {alive.mean():.0f}")
