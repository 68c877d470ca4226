#!/usr/bin/env python3
"""Micro-benchmark of the encoder attention kernels at the benchmark shape (N=544, S=256, H=12)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pacednegatives_amd import kernels as K

N, S, H = int(os.environ.get("N", 544)), int(os.environ.get("S", 256)), 12
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


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

fl = 4.0 * S * S * 64 * H * N
for p in (0.0, 0.1):
    t = timeit(lambda: K.enc_attn_fwd(q, k, v, ctx, lse, tbl, N, S, H, drop_p=p, drop_seed=1))
    print(f"fwd  p={p}: {t:8.1f} us  {fl/t/1e6:7.1f} TFLOP/s", flush=True)
    for db in (dtbl, None):
        t = timeit(lambda: K.enc_attn_bwd(q, k, v, ctx, dctx, lse, tbl, dqkv[:, :I], dqkv[:, I:2 * I], dqkv[:, 2 * I:], db, N, S, H, drop_p=p, drop_seed=1))
        print(f"bwd  p={p} dbias={'yes' if db is not None else 'no '}: {t:8.1f} us  {2.5*fl/t/1e6:7.1f} TFLOP/s (algorithmic 2.5x fwd)", flush=True)
