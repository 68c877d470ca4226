#!/usr/bin/env python3
"""What the last, partly filled round of a 256x256 product costs: the N = 768 products of the encoder at token counts that
give 6.0, 6.375 (the benchmark's 139264 rows) and 6.75 rounds of 256 tiles."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from pacednegatives_amd import kernels as K  # noqa: E402

dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for name, Kc, N, epi in (("o-proj+res", 768, 768, K.EPI_RES_DROP), ("wo+res", 3072, 768, K.EPI_RES_DROP), ("qkv", 768, 2304, K.EPI_BF16),
                         ("wi+relu", 768, 3072, K.EPI_RELU_DROP)):
    for M in (131072, 139264, 147456):
        a = torch.randn(M, Kc, device=dev, generator=g).to(torch.bfloat16)
        w = (torch.randn(N, Kc, device=dev, generator=g) * 0.05).to(torch.bfloat16)
        c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        aux = torch.randn(M, N, device=dev, generator=g).to(torch.bfloat16) if epi == K.EPI_RES_DROP else None
        us = timed(lambda: K.gemm(a, w, c, epilogue=epi, aux=aux, drop_p=0.1, drop_seed=5))
        tiles = (M // 256) * (N // 256)
        print(f"{name:12s} M={M:7d} tiles={tiles:5d} rounds={tiles / 256:6.3f}  {us:8.1f} us  {us / (tiles / 256):7.2f} us per round-equivalent  "
              f"{2.0 * M * N * Kc / us / 1e6:7.1f} TFLOP/s")
        del a, w, c, aux
