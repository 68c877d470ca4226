#!/usr/bin/env python3
"""Yardstick only (not part of the product): the library's plain bf16 products next to the vendor GEMM
(torch.matmul, i.e. hipBLASLt / rocBLAS) on the benchmark workload's shapes, same operands, same process.
The vendor call has no fused epilogue, so only plain-epilogue products are compared."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pacednegatives_amd import kernels as K

M = int(os.environ.get("M", 139264))
SHAPES = [  # name, ta, tb, rows, cols, contraction
    ("qkv  y=xW^T", 0, 0, M, 2304, 768),
    ("o    y=xW^T", 0, 0, M, 768, 768),
    ("wi   y=xW^T", 0, 0, M, 3072, 768),
    ("wo   y=xW^T", 0, 0, M, 768, 3072),
    ("dact dx=dyW", 0, 1, M, 3072, 768),
    ("dxn  dx=dyW k3072", 0, 1, M, 768, 3072),
    ("dWi  dW=dy^Tx", 1, 1, 3072, 768, M),
]


def timed(fn, n=5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    rn = lambda *s: torch.randn(*s, device=dev, generator=g).to(torch.bfloat16)
    for name, ta, tb, R, C, Kc in SHAPES:
        a = rn(Kc, R) if ta else rn(R, Kc)
        b = rn(Kc, C) if tb else rn(C, Kc)
        atomic = bool(ta)
        out = torch.zeros(R, C, device=dev, dtype=torch.float32 if atomic else torch.bfloat16)
        A = a.t() if ta else a
        B = b if tb else b.t()
        ref = torch.empty(R, C, device=dev, dtype=torch.bfloat16)
        ours, blas = [], []
        for _ in range(3):
            ours.append(timed(lambda: K.gemm(a, b, out, trans_a=bool(ta), trans_b=bool(tb),
                                             epilogue=K.EPI_ATOMIC_F32 if atomic else K.EPI_BF16,
                                             splits=0 if atomic else 1)))
            blas.append(timed(lambda: torch.matmul(A, B, out=ref)))
        fl = 2.0 * R * C * Kc
        print(f"{name:20s} ours {min(ours)*1e3:8.1f} us {fl/min(ours)/1e9:7.1f} TFLOP/s | vendor {min(blas)*1e3:8.1f} us "
              f"{fl/min(blas)/1e9:7.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
