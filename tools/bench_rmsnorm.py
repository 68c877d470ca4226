"""Isolated timings of the RMSNorm kernels on the encoder shape of the bench workload (rows x d bf16)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from pacednegatives_amd import kernels as K

M = int(os.environ.get("ROWS", 139264))
d = int(os.environ.get("D", 768))
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s: torch.randn(*s, device=dev, generator=g).to(torch.bfloat16)
x, dy, dres, w = rn(M, d), rn(M, d), rn(M, d), rn(d)
y, dx, dx2 = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
rstd = torch.empty(M, device=dev)
dw = torch.zeros(d, device=dev)
big = torch.zeros(1 << 28, device=dev)   # 1 GiB, read between repetitions: evicts the operands without leaving dirty lines


def run(name, fn, nbytes):
    best = 1e9
    for _ in range(5):
        big.sum()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    print(f"{name:34s} {best * 1e3:8.1f} us {nbytes / best / 1e9:7.2f} TB/s", flush=True)


B = M * d * 2
K.rmsnorm_fwd(x, w, y, rstd)
run("fwd", lambda: K.rmsnorm_fwd(x, w, y, rstd), 2 * B)
run("fwd drop", lambda: K.rmsnorm_fwd(x, w, y, rstd, drop_p=0.1, drop_seed=3), 2 * B)
run("bwd dres dw", lambda: K.rmsnorm_bwd(dy, x, w, rstd, dx, dres=dres, dw=dw), 4 * B)
run("bwd dres dw dx2 drop2", lambda: K.rmsnorm_bwd(dy, x, w, rstd, dx, dres=dres, dx2=dx2, dw=dw, drop2_p=0.1, drop2_seed=5), 5 * B)
run("bwd dw dx2 drop drop2", lambda: K.rmsnorm_bwd(dy, x, w, rstd, dx, dx2=dx2, dw=dw, drop_p=0.1, drop_seed=4, drop2_p=0.1, drop2_seed=5), 4 * B)
run("bwd dres (no dw)", lambda: K.rmsnorm_bwd(dy, x, w, rstd, dx, dres=dres), 4 * B)
