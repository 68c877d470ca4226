#!/usr/bin/env python3
"""Does the [token, 3*H*64] layout (a head's slice = 128-byte pieces at a 4608-byte stride) cost the encoder attention
kernels time?  The same arithmetic as 544 sequences x 12 heads, run as 6528 one-head sequences whose Q, K, V, dO, O are
contiguous [token, 64] matrices (what a head-major layout would give)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pacednegatives_amd import kernels as K

S = 256
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for name, N, H in (("token-major, 12 heads", 544, 12), ("head-major (one-head sequences)", 544 * 12, 1)):
    I = H * 64
    if H == 1:      # three separate contiguous matrices
        q, k, v = [(torch.randn(N * S, 64, device=dev, generator=g) * 0.3).to(torch.bfloat16) for _ in range(3)]
        dq, dk, dv = [torch.empty_like(q) for _ in range(3)]
    else:
        qkv = (torch.randn(N * S, 3 * I, device=dev, generator=g) * 0.3).to(torch.bfloat16)
        dqkv = torch.empty_like(qkv)
        q, k, v = qkv[:, :I], qkv[:, I:2 * I], qkv[:, 2 * I:]
        dq, dk, dv = dqkv[:, :I], dqkv[:, I:2 * I], dqkv[:, 2 * I:]
    dctx = torch.randn(N * S, I, device=dev, generator=g).to(torch.bfloat16)
    tbl = torch.randn(H, 2 * S - 1, device=dev, generator=g) * 0.3
    ctx = torch.empty(N * S, I, device=dev, dtype=torch.bfloat16)
    lse = torch.empty(N * H * S, device=dev)
    for p in (0.0, 0.1):
        t = timeit(lambda: K.enc_attn_fwd(q, k, v, ctx, lse, tbl, N, S, H, drop_p=p, drop_seed=1))
        tb = timeit(lambda: K.enc_attn_bwd(q, k, v, ctx, dctx, lse, tbl, dq, dk, dv, None, N, S, H, drop_p=p, drop_seed=1))
        print(f"{name:34s} p={p}: forward {t:7.1f} us   backward without bias gradient {tb:7.1f} us", flush=True)
    del q, k, v, dq, dk, dv, dctx, ctx
