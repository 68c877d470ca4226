#!/usr/bin/env python3
"""Does the row pitch of the operands matter to the phased GEMM (L2 set / channel mapping)?  Products of the benchmark
with the weight matrix (and optionally the activations) stored at a pitch of K + pad elements instead of K."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pacednegatives_amd import kernels as K

M = 139264
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


def padded(rows, cols, pad, scale=1.0):
    buf = (torch.randn(rows, cols + pad, device=dev, generator=g) * scale).to(torch.bfloat16)
    return buf[:, :cols]


PADS = [int(x) for x in os.environ.get("PADS", "0,64,128,192,256").split(",")]
# forward y = x W^T: W [N, K]; input gradient dx = dy W: W [K, N] read transposed
for name, N, Kc, tb in (("qkv fwd", 2304, 768, False), ("wi fwd", 3072, 768, False), ("o fwd", 768, 768, False), ("wo fwd", 768, 3072, False),
                        ("qkv dx", 768, 2304, True), ("wi dx", 768, 3072, True), ("wo dx (plain)", 3072, 768, True)):
    a = padded(M, Kc, 0)
    c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    line = f"{name:14s}"
    for pad in PADS:
        b = padded(Kc, N, pad, 0.05) if tb else padded(N, Kc, pad, 0.05)
        t = min(timeit(lambda: K.gemm(a, b, c, trans_b=tb)) for _ in range(3))
        line += f"  W+{pad}: {t:6.1f}"
        del b
    print(line + "  us", flush=True)
    del a, c
