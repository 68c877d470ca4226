#!/usr/bin/env python3
"""Shader clock (rocm-smi, polled from a thread) while the encoder attention kernels run back to back, on random and on
all-zero operands: does the part clock down under these kernels, and do they follow the clock?"""
import os, re, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pacednegatives_amd import kernels as K

N, S, H = 544, 256, 12
M = N * S
dev = "cuda"
SECONDS = float(os.environ.get("SECONDS_PER_CASE", 3))


def poll(stop, out):
    while not stop.is_set():
        try:
            txt = subprocess.run(["rocm-smi", "--showclocks"], capture_output=True, text=True, timeout=10).stdout
        except Exception as e:
            txt = f"ERR {e}"
        sclk = re.findall(r"sclk clock level: \S+ \((\d+)Mhz\)", txt)
        out.append(int(sclk[0]) if sclk else None)
        time.sleep(0.2)


def case(name, fn):
    fn(); torch.cuda.synchronize()
    stop, samples = threading.Event(), []
    th = threading.Thread(target=poll, args=(stop, samples)); th.start()
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < SECONDS:
        for _ in range(20):
            fn()
        torch.cuda.synchronize(); n += 20
    el = time.perf_counter() - t0
    stop.set(); th.join()
    clk = [c for c in samples[1:-1] if c]
    print(f"{name:44s} {el / n * 1e6:8.1f} us per call   sclk MHz {min(clk) if clk else '?'}-{max(clk) if clk else '?'} (mean {sum(clk) / max(1, len(clk)):.0f})", flush=True)


for zero in (False, True):
    g = torch.Generator(device=dev).manual_seed(0)
    scale = 0.0 if zero else 0.5
    qkv = (torch.randn(3 * H, M, 64, device=dev, generator=g) * scale).to(torch.bfloat16)
    bias = torch.randn(H, 2 * S - 1, device=dev, generator=g) * scale
    ctx = torch.empty(H, M, 64, device=dev, dtype=torch.bfloat16)
    lse = torch.empty(N, H, S, device=dev)
    q, k, v = qkv[:H], qkv[H:2 * H], qkv[2 * H:]
    do = (torch.randn(H, M, 64, device=dev, generator=g) * scale).to(torch.bfloat16)
    dqkv = torch.empty_like(qkv)
    dq, dk, dv = dqkv[:H], dqkv[H:2 * H], dqkv[2 * H:]
    dbias = torch.zeros(64, H, 2 * S - 1, device=dev)
    tag = "all-zero operands" if zero else "random operands"
    case(f"forward, {tag}", lambda: K.enc_attn_fwd(q, k, v, ctx, lse, bias, N, S, H, drop_p=0.1, drop_seed=3))
    case(f"backward (dK/dV + dQ), {tag}", lambda: K.enc_attn_bwd(q, k, v, ctx, do, lse, bias, dq, dk, dv, dbias, N, S, H, drop_p=0.1, drop_seed=3))
