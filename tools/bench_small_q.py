#!/usr/bin/env python3
"""The pool sweep where HBM binds: few queries per pass (Q FLOP per pool byte against a ridge of ~312).

10 M x 768 bf16 pool (15.36 GB, read once per sweep), top-1000, Q = 64 / 128 / 256 through paced_pool_topk and Q = 64
through paced_pool_topk_static.  Reports ms per sweep and both roofline fractions per Q as one JSON line.
usage: bench_small_q.py [--pool N] [--iters I] [--qs 64,128,256]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from pacednegatives_amd import kernels as K  # noqa: E402

HBM_PEAK, MFMA_PEAK = 8.0e12, 2.5e15


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pool", type=int, default=10_000_000)
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--k", type=int, default=1000)
    ap.add_argument("--qs", default="64,128,256")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    D, k, Np = 768, args.k, args.pool
    g = torch.Generator(device=dev).manual_seed(7)
    pool = torch.empty(Np, D, dtype=torch.bfloat16, device=dev)
    for lo in range(0, Np, 1 << 20):
        hi = min(Np, lo + (1 << 20))
        pool[lo:hi] = (torch.randn(hi - lo, D, device=dev, generator=g) * 0.05).to(torch.bfloat16)
    out = {"pool_rows": Np, "k": k, "runs": []}
    cases = [("paced_pool_topk", int(q)) for q in args.qs.split(",")] + [("paced_pool_topk_static", 64)]
    for entry, Q in cases:
        queries = (torch.randn(Q, D, device=dev, generator=g) * 0.05).to(torch.bfloat16)
        if entry == "paced_pool_topk":
            ws = torch.empty(K._lib.load().paced_pool_topk_workspace_bytes(Q, k), dtype=torch.uint8, device=dev)
            run = lambda: K.pool_topk(queries, pool, k, workspace=ws)  # noqa: E731
        else:
            s, i = torch.empty(Q, k, device=dev), torch.empty(Q, k, dtype=torch.int64, device=dev)
            run = lambda: K.pool_topk_static(queries, pool, k, s, i)  # noqa: E731
        run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.iters):
            run()
        torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / args.iters
        bytes_alg = Np * D * 2 + Q * D * 2 + Q * k * 12
        flops = 2.0 * Q * Np * D
        out["runs"].append({"entry": entry, "Q": Q, "ms_per_sweep": el * 1e3, "scores_per_s": Q * Np / el,
                            "hbm_GBps": bytes_alg / el / 1e9, "hbm_frac": bytes_alg / el / HBM_PEAK,
                            "mfma_frac": flops / el / MFMA_PEAK, "bound": "hbm" if Q < MFMA_PEAK / HBM_PEAK else "mfma"})
    print(json.dumps(out))


if __name__ == "__main__":
    main()
