#!/usr/bin/env python3
"""Micro-benchmark of paced_gemm_bf16 on the T5-base shapes of the benchmark workload, per tile
configuration.  Random N(0,1) operands; HIP-event timing; interleaved rounds in one process."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pacednegatives_amd import kernels as K

M = int(os.environ.get("M", 139264))
TILES = tuple(int(t) for t in os.environ.get("TILES", "128,256").split(","))
SHAPES = [  # (name, ta, tb, epi, rows_out, cols_out, contraction)
    ("qkv   NT", 0, 0, K.EPI_BF16, M, 2304, 768),
    ("o     NT res", 0, 0, K.EPI_RES_DROP, M, 768, 768),
    ("wi    NT relu", 0, 0, K.EPI_RELU_DROP, M, 3072, 768),
    ("wo    NT res", 0, 0, K.EPI_RES_DROP, M, 768, 3072),
    ("dact  NN mask", 0, 1, K.EPI_MASK, M, 3072, 768),
    ("dxn   NN k3072", 0, 1, K.EPI_BF16, M, 768, 3072),
    ("dxn   NN k2304", 0, 1, K.EPI_BF16, M, 768, 2304),
    ("dWi   TN", 1, 1, K.EPI_ATOMIC_F32, 3072, 768, M),
    ("dWqkv TN", 1, 1, K.EPI_ATOMIC_F32, 2304, 768, M),
    ("dWo   TN", 1, 1, K.EPI_ATOMIC_F32, 768, 768, M),
]
if os.environ.get("EXTRA"):     # the same products with cheaper epilogues: what each epilogue costs
    SHAPES += [
        ("o     NT plain", 0, 0, K.EPI_BF16, M, 768, 768),
        ("o     NT res p=0", 0, 0, K.EPI_RES_DROP, M, 768, 768),
        ("wi    NT plain", 0, 0, K.EPI_BF16, M, 3072, 768),
        ("wi    NT relu p=0", 0, 0, K.EPI_RELU_DROP, M, 3072, 768),
        ("dact  NN bits", 0, 1, K.EPI_MASK_BITS, M, 3072, 768),
        ("dact  NN tilebits", 0, 1, K.EPI_MASK_BITS, M, 3072, 768),      # tile-ordered bit matrix (256x256 kernel only)
        ("wi    NT relu+bits", 0, 0, K.EPI_RELU_DROP, M, 3072, 768),
        ("wi    NT relu+tilebits", 0, 0, K.EPI_RELU_DROP, M, 3072, 768),
        ("dact  NN plain", 0, 1, K.EPI_BF16, M, 3072, 768),
    ]


def splits_for(rows, cols, contraction, tile):
    tiles = ((rows + tile - 1) // tile) * ((cols + tile - 1) // tile)
    want = max(1, ((1024 if tile == 128 else 256) + tiles - 1) // tiles)
    return max(1, min(want, (contraction + 63) // 64))


def main():
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    rn = lambda *s: torch.randn(*s, device=dev, generator=g).to(torch.bfloat16)
    res = {}
    only = os.environ.get("ONLY")
    for name, ta, tb, epi, R, C, Kc in SHAPES:
        if only and only not in name:
            continue
        a = rn(Kc, R) if ta else rn(R, Kc)
        b = rn(Kc, C) if tb else rn(C, Kc)
        out = torch.zeros(R, C, device=dev, dtype=torch.float32 if epi in (K.EPI_F32, K.EPI_ATOMIC_F32) else torch.bfloat16)
        aux = rn(R, C) if epi in (K.EPI_RES_DROP, K.EPI_MASK) else None
        if epi == K.EPI_MASK_BITS or "bits" in name:
            aux = torch.randint(0, 256, (R, C // 8), device=dev, dtype=torch.uint8)
        tiled = "tilebits" in name
        if tiled:
            K.gemm_set_tile(256)
            aux = torch.randint(0, 256, (K._lib.load().paced_gemm_tile_bits_bytes(R, C),), device=dev, dtype=torch.uint8)
        drop = 0.1 if epi in (K.EPI_RES_DROP, K.EPI_RELU_DROP) and "p=0" not in name else 0.0
        for rnd in range(3):
            for tile in TILES:
                if tiled and tile != 256:
                    res.setdefault((name, tile), []).append(float("nan"))
                    continue
                K.gemm_set_tile(tile)
                sp = 0 if epi == K.EPI_ATOMIC_F32 else 1
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    K.gemm(a, b, out, trans_a=bool(ta), trans_b=bool(tb), epilogue=epi, aux=aux, drop_p=drop, drop_seed=1, splits=sp)
                e1.record()
                torch.cuda.synchronize()
                res.setdefault((name, tile), []).append(e0.elapsed_time(e1) / 5)
        for tile in TILES:
            ms = min(res[(name, tile)])
            print(f"{name:16s} tile {tile}: {ms*1e3:8.1f} us  {2.0*R*C*Kc/ms/1e9:8.1f} TFLOP/s   (rounds {[round(x*1e3) for x in res[(name, tile)]]})", flush=True)
    K.gemm_set_tile(0)


if __name__ == "__main__":
    main()
