#!/usr/bin/env python3
"""Benchmark of the lceT5 training step (and, as a secondary figure, the negative-pool scorer).

  python bench.py --gpus N --steps K --warmup W

A "step" is one reference-faithful ``LCEModel.training_step`` (two no-grad T5 forwards, the LCE loss,
the eta meta-update, two forwards with state, backward, AdamW; dropout 0.1 active as under Lightning's
train mode) over one synthetic batch that is already resident in HBM.  Workload at every N: the
per-GPU shard of the configuration the metric is quoted on -- T5-base, K=16 negatives, seq_len 256,
global batch 256 over 8 GPUs = 32 tuples per GPU (weak scaling: per-GPU work is fixed).  With N > 1
the driver starts one process per GPU (torch.distributed.run); gradients are all-reduced over RCCL.

Rank 0 prints ONE JSON line.  ``roofline`` refers to the dominant kernel (the phased 256x256 bf16 MFMA
contraction ``gemm_phased_kernel``): algorithmic FLOPs of its launches divided by their summed duration,
measured with HIP events on the launch stream during one extra instrumented step run right after the
timed region (instrumenting the timed steps themselves would add event overhead to ``value``);
``roofline.all_gemm_launches`` pools the small-tile kernels' launches (the decoder's rows) with them.
``cpu_baseline`` times the CPU oracle (fp32 torch restatement of the reference step) on the box's
host cores on a bounded sample of the same workload.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
import warnings

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import pacednegatives_amd  # noqa: E402,F401  (first: it sets HIP_FORCE_DEV_KERNARG before the HIP runtime initialises)
import torch  # noqa: E402

BF16_DENSE_PEAK = 2.5e15     # FLOP/s, MI355X dense bf16 MFMA
HBM_PEAK = 8.0e12            # B/s


TRAFFIC_STEP = "profiles/r3_pmc_traffic.json"            # tools/measure_r3.sh: two --pmc passes of `bench.py --steps 1 --warmup 0`
TRAFFIC_MINING = "profiles/r3_pmc_traffic_mining.json"   # ... of tools/mine_once.py


def committed_traffic(rel_path, keys, applies):
    """(bytes, source) from a committed counter summary: the value is a constant of the repository, not a measurement of
    this run, and `source` says which file and for which kernel sources it was taken (their digest is stored in the file
    by tools/pmc_summary.py; when the kernels have changed since, the value is withheld)."""
    path = os.path.join(ROOT, rel_path)
    if not applies or not os.path.exists(path):
        return None, None
    try:
        doc = json.load(open(path))
        value = doc[keys[0]][keys[1]]
    except Exception:
        return None, None
    want, have = doc.get("kernel_sources_sha256"), kernel_sources_digest()
    if want is not None and want != have:
        return None, f"{rel_path} was taken with other kernel sources ({want[:12]} != {have[:12]}): withheld"
    return value, f"committed profile {rel_path} (rocprofv3 --pmc, separate passes; kernel sources {have[:12]}), not this run"


def kernel_sources_digest():
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "pacednegatives_amd", "csrc")
    for f in sorted(os.listdir(d)):
        with open(os.path.join(d, f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read())
    return h.hexdigest()


def t5_forward_flops(dims, S, T=2):
    """Algorithmic FLOPs of one sequence's forward (2 per MAC; GEMMs, attention products, LM head)."""
    d, I, ff, V = dims.d_model, dims.inner, dims.d_ff, dims.vocab_size
    enc = dims.num_layers * (8 * S * d * I + 4 * S * S * I + 4 * S * d * ff)
    dec = dims.num_decoder_layers * (8 * T * d * I + 4 * T * T * I + 2 * T * d * I + 4 * S * d * I + 4 * T * S * I
                                     + 2 * T * I * d + 4 * T * d * ff)
    return enc + dec + 2 * T * d * V


def run_train(args, rank, world, device):
    from pacednegatives_amd import kernels as K, parallel
    from pacednegatives_amd.config import resolve
    from pacednegatives_amd.lceT5 import LCEModel
    from pacednegatives_amd.synthetic import token_batch
    import torch.distributed as dist

    dims = resolve(args.model)
    B, nneg, S = args.batch, args.n_neg, args.seq
    hp = {"model_name": args.model, "total_steps": 100000 // B, "eta": 0.3466, "batch_size": B, "lr": 1e-3, "meta_lr": 1e-3,
          "n": nneg, "warmup_steps": 10000 // B, "use_mean": False, "ignore_index": -100, "dropout": args.dropout, "seed": 0}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model = LCEModel(hp, device=device)
    batch = token_batch(B, nneg, S, dims.vocab_size, seed=1234 + rank, device=device)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        model.training_step(batch, i)
    sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        model.training_step(batch, args.warmup + i)
    sync()
    elapsed = time.perf_counter() - t0
    elapsed = parallel.max_over_ranks(elapsed, device)

    # one instrumented step: per-launch HIP events around the contraction kernel
    roof = None
    K.PROFILE = []
    was_overlapped = bool(model.overlap_forwards)
    model.overlap_forwards = False      # per-launch durations: no second stream sharing the GPU with the timed kernel
    model.training_step(batch, args.warmup + args.steps)
    torch.cuda.synchronize()
    prof, K.PROFILE = K.PROFILE, None
    by_sym = {}
    for key, flops, e0, e1 in prof:
        ms = e0.elapsed_time(e1)
        s = by_sym.setdefault(key, [0.0, 0.0, 0])
        s[0] += flops; s[1] += ms * 1e-3; s[2] += 1
    # HBM-side counters cannot be collected inside this run (rocprofv3 --pmc wraps the whole process): `traffic` quotes the
    # committed summary of two --pmc passes of this same command, and says so; null for any other workload
    traffic, traffic_source = committed_traffic(TRAFFIC_STEP, ("gemm_phased_kernel", "hbm_bytes_per_launch"),
                                                (args.model, B, nneg, S) == ("t5-base", 32, 16, 256))

    def pooled(items):
        f, t, n = (sum(v[i] for _, v in items) for i in range(3))
        return f, t, n
    all_f, all_t, all_n = pooled(list(by_sym.items()))
    # the dominant kernel: the phased 256x256 kernel (key[3] == 256); the small-tile kernels carry the decoder's
    # two-token rows, a few percent of the time and next to none of the FLOPs
    phased = [(k, v) for k, v in by_sym.items() if k[3] == 256]
    tot_f, tot_t, tot_n = pooled(phased)
    if tot_t > 0:
        dom = max(phased, key=lambda kv: kv[1][1])
        roof = {"bound": "mfma", "kernel": "gemm_phased_kernel (every launch of one training step)", "achieved": tot_f / tot_t / 1e12,
                "peak": BF16_DENSE_PEAK / 1e12, "unit": "TFLOP/s", "frac": tot_f / tot_t / BF16_DENSE_PEAK, "traffic": traffic,
                "traffic_unit": "HBM bytes per launch (FETCH_SIZE x2 + WRITE_SIZE, separate --pmc passes)",
                "traffic_source": traffic_source,
                "flops_per_launch": tot_f / tot_n, "launches": tot_n, "avg_launch_us": tot_t / tot_n * 1e6,
                "largest_share_variant": {"trans_a,trans_b,epilogue": list(dom[0][:3]), "TFLOP/s": dom[1][0] / dom[1][1] / 1e12,
                                    "launches": dom[1][2], "avg_launch_us": dom[1][1] / dom[1][2] * 1e6},
                "share_of_step": tot_t / (elapsed / args.steps),
                "all_gemm_launches": {"kernels": "gemm_phased_kernel + gemm_bf16_kernel + gemm_ring_kernel", "launches": all_n,
                                      "achieved": all_f / all_t / 1e12, "frac": all_f / all_t / BF16_DENSE_PEAK,
                                      "share_of_step": all_t / (elapsed / args.steps)}}
    F = t5_forward_flops(dims, S)
    tuples_per_s = world * B * args.steps / elapsed
    flops_per_tuple = 4 * (1 + nneg) * F
    out = {
        "metric": "training triples/sec (q + 1+K docs) at T5-base, K=16", "value": tuples_per_s, "unit": "triples/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": f"lceT5 training_step, {args.model} random-init, one triple = query + positive + K={nneg} negatives "
                               f"(1+K documents), seq_len={S}, "
                               f"{B} tuples per GPU (global batch {B * world}), dropout {args.dropout}, "
                               "reference-faithful 4 forwards + backward + 2 AdamW",
                   "parallelism": f"dp{world}", "global_batch": B * world, "seq_len": S},
        "ranks": {"world_size": dist.get_world_size() if dist.is_initialized() else 1,
                  "backend": dist.get_backend() if dist.is_initialized() else None,
                  "early_update": bool(model.early_update), "overlap_forwards": was_overlapped},
        "roofline": roof,
        "step_mfma_frac": tuples_per_s * flops_per_tuple / (world * BF16_DENSE_PEAK),
        "algorithmic_flops_per_tuple": flops_per_tuple,
    }
    return out, model


def cpu_baseline(args, dims):
    """Oracle (fp32 CPU restatement of the reference step) on a bounded sample of the same workload: B=2 tuples of the same
    shape on every core the job may use, and the same step at ONE thread on a tuple cut to 1+1 sequences (a step's cost is
    linear in its sequences: nothing but the loss couples them), scaled to 1+K."""
    from oracle import lce_ref
    from oracle.t5_ref import T5Dims as RefDims, init_params
    from pacednegatives_amd.synthetic import token_batch
    rd = RefDims(dims.vocab_size, dims.d_model, dims.d_kv, dims.num_heads, dims.d_ff, dims.num_layers, dims.num_decoder_layers)
    threads = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16")))

    def timed(B, n_neg, n_threads, max_steps, warm, budget_s):
        torch.set_num_threads(n_threads)
        P = init_params(rd, 0)
        eta = torch.tensor([0.3466])
        opt, meta = lce_ref.HFAdamW(P, 1e-3), lce_ref.HFAdamW({"eta": eta}, 1e-3)
        p, n = token_batch(B, n_neg, args.seq, dims.vocab_size, seed=1234)
        steps, used = 0, 0.0
        for i in range(warm + max_steps):
            t0 = time.perf_counter()
            lce_ref.training_step(P, rd, eta, opt, meta, i, p["input_ids"], p["labels"], n["input_ids"], n["labels"], n_neg,
                                  False, 10000, 100000)
            dt = time.perf_counter() - t0
            if i >= warm:
                used += dt; steps += 1
            if used > budget_s:
                break
        return steps, used

    B = args.cpu_batch
    steps, used = timed(B, args.n_neg, threads, args.cpu_steps, 1, 25.0)
    out = {"value": B * steps / used, "unit": "triples/s", "cores": threads, "kind": "port",
           "sample": f"{steps} timed steps (after 1 warm-up) of B={B} tuples (1+{args.n_neg} sequences x {args.seq} tokens each), "
                     f"{args.model} dims, fp32, dropout 0, torch CPU"}
    if args.cpu_one_thread:
        s1, u1 = timed(1, 1, 1, 1, 0, 0.0)
        out["one_thread"] = {"value": s1 / u1 * 2.0 / (1 + args.n_neg), "unit": "triples/s", "cores": 1,
                             "sample": f"1 step (no warm-up) of 1 tuple cut to 1+1 sequences x {args.seq} tokens, {u1:.1f} s, "
                                       f"scaled by 2/(1+{args.n_neg}) to whole tuples"}
    torch.set_num_threads(threads)
    return out


def cpu_baseline_mining(pool_rows, queries, k, budget_s=20.0):
    """The negative-pool selection on the host cores: fp32 ``torch.matmul`` of 64-query chunks against a 1 M-row slice
    of the same pool, ``torch.topk`` per chunk, until ``budget_s`` seconds are used; scores/s over the chunks finished."""
    threads = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16")))
    torch.set_num_threads(threads)
    pool = pool_rows.float().cpu()
    q = queries.float().cpu()
    done, used = 0, 0.0
    for lo in range(0, q.shape[0], 64):
        t0 = time.perf_counter()
        sc = q[lo:lo + 64] @ pool.t()
        torch.topk(sc, min(k, pool.shape[0]), dim=1)
        dt = time.perf_counter() - t0
        if lo > 0:                   # the first chunk warms the thread pool up
            used += dt; done += sc.shape[0]
        if used > budget_s:
            break
    return {"value": done * pool.shape[0] / used, "unit": "scores/s", "cores": threads, "kind": "port",
            "sample": f"{done} queries (chunks of 64, after one warm-up chunk) x {pool.shape[0]} pool rows x {pool.shape[1]} dims: "
                      f"fp32 torch.matmul + torch.topk({k}) on the host, {used:.1f} s"}


def run_mining(args, rank, world, device):
    """Secondary figure: pool rows are sharded over ranks, queries replicated, shard results merged."""
    from pacednegatives_amd import kernels as K, mining, parallel
    import torch.distributed as dist
    Q, D, k = args.mine_queries, 768, args.mine_k
    Np = args.mine_pool // world
    g = torch.Generator(device=device).manual_seed(7 + rank)
    pool = torch.empty(Np, D, dtype=torch.bfloat16, device=device)
    step = 1 << 20
    for lo in range(0, Np, step):
        hi = min(Np, lo + step)
        pool[lo:hi] = (torch.randn(hi - lo, D, device=device, generator=g) * 0.05).to(torch.bfloat16)
    gq = torch.Generator(device=device).manual_seed(7)
    queries = (torch.randn(Q, D, device=device, generator=gq) * 0.05).to(torch.bfloat16)
    ws = torch.empty(K._lib.load().paced_pool_topk_workspace_bytes(Q, k), dtype=torch.uint8, device=device)

    def once():
        return mining.pool_topk_sharded(queries, pool, k, rank * Np, workspace=ws)
    once()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.mine_iters):
        once()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    el = parallel.max_over_ranks((time.perf_counter() - t0) / args.mine_iters, device)
    total_rows = Np * world
    flops = 2.0 * Q * total_rows * D
    bytes_alg = total_rows * D * 2 + Q * D * 2 + Q * k * 12
    traffic, traffic_source = committed_traffic(TRAFFIC_MINING, ("sweep", "hbm_bytes_per_sweep"),
                                                (Q, total_rows, k, world) == (4096, 10_000_000, 1000, 1))
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline_mining(pool[:min(Np, 1_000_000)], queries, k)
    # MFMA binds: all Q queries share one pass over the pool, so the intensity is Q FLOP per pool byte (4096 at the
    # benchmark shape) against a ridge of peak FLOP/s / peak B/s = 312; HBM would bind for fewer than ~312 queries per pass.
    out = {"metric": "negative-pool scores/sec", "value": Q * total_rows / el, "unit": "scores/s", "ms_per_sweep": el * 1e3,
            "config": {"workload": f"top-{k} of {Q} queries x {total_rows} pool rows x {D} dims, bf16, pool sharded over {world} GPU(s)"},
            "roofline": {"bound": "mfma", "achieved": flops / el / 1e12, "peak": BF16_DENSE_PEAK / 1e12, "unit": "TFLOP/s",
                         "frac": flops / el / (world * BF16_DENSE_PEAK), "hbm_frac": bytes_alg / el / (world * HBM_PEAK),
                         "algorithmic_bytes": bytes_alg, "traffic": traffic, "traffic_source": traffic_source,
                         "binding": f"mfma: {Q} FLOP per pool byte against a ridge of {BF16_DENSE_PEAK / HBM_PEAK:.0f}; "
                                    "HBM binds below that many queries per pass"}}
    if cpu is not None:
        out["cpu_baseline"] = cpu
    if world == 1 and not args.skip_small_q:
        # The same sweep where the pool's bytes bind (Q below the ridge): the HBM-streaming kernel takes it up to 128 queries.
        regime = []
        for entry, Qs in (("paced_pool_topk", 64), ("paced_pool_topk", 128), ("paced_pool_topk", 256), ("paced_pool_topk_static", 64)):
            qs = queries[:Qs].contiguous()
            if entry == "paced_pool_topk":
                ws_s = torch.empty(K._lib.load().paced_pool_topk_workspace_bytes(Qs, k), dtype=torch.uint8, device=device)
                run = lambda: K.pool_topk(qs, pool, k, workspace=ws_s)  # noqa: E731
            else:
                os_, oi_ = torch.empty(Qs, k, device=device), torch.empty(Qs, k, dtype=torch.int64, device=device)
                run = lambda: K.pool_topk_static(qs, pool, k, os_, oi_)  # noqa: E731
            run()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            el_s = (time.perf_counter() - t0) / 3
            b_alg = Np * D * 2 + Qs * D * 2 + Qs * k * 12
            regime.append({"entry": entry, "queries": Qs, "ms_per_sweep": el_s * 1e3, "scores_per_s": Qs * Np / el_s,
                           "roofline": {"bound": "hbm", "achieved": b_alg / el_s / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                                        "frac": b_alg / el_s / HBM_PEAK, "algorithmic_bytes": b_alg,
                                        "mfma_frac": 2.0 * Qs * Np * D / el_s / BF16_DENSE_PEAK}})
            del run
        out["hbm_regime"] = regime
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--model", default="t5-base")
    ap.add_argument("--batch", type=int, default=32, help="tuples per GPU")
    ap.add_argument("--n_neg", type=int, default=16)
    ap.add_argument("--seq", type=int, default=256)
    ap.add_argument("--dropout", type=float, default=0.1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", dest="cpu_steps", type=int, default=3)
    ap.add_argument("--cpu-batch", dest="cpu_batch", type=int, default=2, help="tuples per step of the CPU baseline")
    ap.add_argument("--no-cpu-one-thread", dest="cpu_one_thread", action="store_false", help="skip the 1-thread CPU figure")
    ap.add_argument("--skip-mining", action="store_true")
    ap.add_argument("--mine_pool", type=int, default=10_000_000)
    ap.add_argument("--mine_queries", type=int, default=4096)
    ap.add_argument("--mine_k", type=int, default=1000)
    ap.add_argument("--mine_iters", type=int, default=2)
    ap.add_argument("--skip-small-q", dest="skip_small_q", action="store_true", help="skip the pool sweeps at 64 / 256 queries")
    ap.add_argument("--launch-timeout", dest="launch_timeout", type=float, default=1500.0,
                    help="seconds after which a self-launched multi-rank run is stopped")
    args = ap.parse_args()

    if os.environ.get("PACED_HANG_DUMP"):        # debugging aid: dump every thread's stack and exit after that many seconds
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["PACED_HANG_DUMP"]), exit=True)
    if args.gpus > 1 and "RANK" not in os.environ:
        # `python bench.py --gpus N` without a torchrun environment: N fresh rank processes of this script, rank 0's JSON
        # line relayed; all ranks are stopped (exit code 124) if the run exceeds --launch-timeout seconds.  This
        # process never touches the GPU.
        from pacednegatives_amd.launcher import launch_ranks
        raise SystemExit(launch_ranks(args.gpus, timeout_s=args.launch_timeout, json_only_stdout=True))
    # stdout carries the JSON line and nothing else: libraries write notices there (RCCL prints the path of its
    # shared object when the first communicator is made, gloo its connections), so file descriptor 1 points at
    # stderr until the line is ready
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    from pacednegatives_amd.trainer import init_distributed
    rank, world, local = init_distributed()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local)
    device = torch.device(f"cuda:{local}")

    out, model = run_train(args, rank, world, device)
    dims = model.model.config
    del model
    torch.cuda.empty_cache()
    if not args.skip_mining:
        out["secondary"] = run_mining(args, rank, world, device)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args, dims)
    sys.stdout.flush()
    if rank == 0:
        os.dup2(real_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
