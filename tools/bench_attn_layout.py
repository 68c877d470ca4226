#!/usr/bin/env python3
"""Encoder attention at the benchmark shape with q, k, v as column slices of a [token, 3*H*64] matrix (a head's slice =
128-byte pieces at a 4608-byte stride) against the head-major [3H, token, 64] output of gemm_heads (contiguous blocks);
context, its gradient and dq/dk/dv stay token-major in both.  Also the projection itself, written either way."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pacednegatives_amd import kernels as K

N, S, H = 544, 256, 12
I, M = H * 64, N * S
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


x = torch.randn(M, I, device=dev, generator=g).to(torch.bfloat16)
w = (torch.randn(3 * I, I, device=dev, generator=g) * 0.03).to(torch.bfloat16)
qkv = torch.empty(M, 3 * I, device=dev, dtype=torch.bfloat16)
heads = torch.empty(3 * H, M, 64, device=dev, dtype=torch.bfloat16)
for rnd in range(2):
    print(f"projection, rows of 3*H*64: {timeit(lambda: K.gemm(x, w, qkv)):7.1f} us   head-major: {timeit(lambda: K.gemm(x, w, heads)):7.1f} us", flush=True)
dctx = torch.randn(M, I, device=dev, generator=g).to(torch.bfloat16)
tbl = torch.randn(H, 2 * S - 1, device=dev, generator=g) * 0.3
ctx = torch.empty(M, I, device=dev, dtype=torch.bfloat16)
lse = torch.empty(N * H * S, device=dev)
dqkv = torch.empty(M, 3 * I, device=dev, dtype=torch.bfloat16)
dtbl = torch.zeros(64, H, 2 * S - 1, device=dev)
lay = {"token-major": (qkv[:, :I], qkv[:, I:2 * I], qkv[:, 2 * I:]), "head-major": (heads[:H], heads[H:2 * H], heads[2 * H:])}
for rnd in range(2):
    for name, (q, k, v) in lay.items():
        for p in (0.0, 0.1):
            t = timeit(lambda: K.enc_attn_fwd(q, k, v, ctx, lse, tbl, N, S, H, drop_p=p, drop_seed=1))
            tb = timeit(lambda: K.enc_attn_bwd(q, k, v, ctx, dctx, lse, tbl, dqkv[:, :I], dqkv[:, I:2 * I], dqkv[:, 2 * I:], dtbl, N, S, H, drop_p=p, drop_seed=1))
            print(f"{name:12s} p={p}: forward {t:7.1f} us   backward {tb:7.1f} us", flush=True)
