import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pacednegatives_amd import kernels as K
M = 139264
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s: torch.randn(*s, device=dev, generator=g).to(torch.bfloat16)
K.gemm_set_tile(256)
def run(name, fn, flops):
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 5)
    print(f"{name:28s} {best*1e3:8.1f} us {flops/best/1e9:8.1f} TF", flush=True)
for (N, Kc) in ((768, 768), (768, 3072), (3072, 768)):
    a, b = rn(M, Kc), rn(N, Kc)
    out = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
    aux = rn(M, N)
    row = rn(1, N).expand(M, N)
    fl = 2.0 * M * N * Kc
    print(f"--- N={N} K={Kc}")
    run("plain", lambda: K.gemm(a, b, out), fl)
    run("res p=0", lambda: K.gemm(a, b, out, epilogue=K.EPI_RES_DROP, aux=aux), fl)
    run("res p=0 aux=one row", lambda: K.gemm(a, b, out, epilogue=K.EPI_RES_DROP, aux=row), fl)
    run("res p=0 in place", lambda: K.gemm(a, b, out, epilogue=K.EPI_RES_DROP, aux=out), fl)
    run("res p=.1", lambda: K.gemm(a, b, out, epilogue=K.EPI_RES_DROP, aux=aux, drop_p=0.1, drop_seed=1), fl)
    run("res p=.1 aux=one row", lambda: K.gemm(a, b, out, epilogue=K.EPI_RES_DROP, aux=row, drop_p=0.1, drop_seed=1), fl)
    run("relu p=0", lambda: K.gemm(a, b, out, epilogue=K.EPI_RELU_DROP), fl)
    run("relu p=.1", lambda: K.gemm(a, b, out, epilogue=K.EPI_RELU_DROP, drop_p=0.1, drop_seed=1), fl)
    run("f32 out", lambda: K.gemm(a, b, torch.empty(M, N, device=dev), epilogue=K.EPI_F32), fl) if N == 768 else None
