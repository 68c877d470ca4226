#!/usr/bin/env python3
"""One warm-up and a few timed sweeps of the pool scorer at the benchmark shape (for rocprofv3 kernel traces)."""
import sys, os, time, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--mine_pool", type=int, default=10_000_000)
ap.add_argument("--mine_queries", type=int, default=4096)
ap.add_argument("--mine_k", type=int, default=1000)
ap.add_argument("--mine_iters", type=int, default=2)
args = ap.parse_args()
args.no_cpu_baseline = True
args.skip_small_q = True
torch.cuda.set_device(0)
print(bench.run_mining(args, 0, 1, torch.device("cuda:0")))
