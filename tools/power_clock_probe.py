#!/usr/bin/env python3
"""Measurement aid: shader clock and package power while one product runs back to back for a few seconds.
The figures come from `rocm-smi` (a child process polled from a thread); the products are the library's own and,
as a yardstick, the vendor GEMM (torch.matmul) on the same operands."""
import sys, os, re, subprocess, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pacednegatives_amd import kernels as K

M = 139264
SECONDS = float(os.environ.get("SECONDS_PER_CASE", 4))


def poll(stop, out):
    while not stop.is_set():
        try:
            txt = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=10).stdout
        except Exception as e:          # the tool may be missing or slow; the probe then reports nothing
            txt = f"ERR {e}"
        sclk = re.findall(r"sclk clock level: \S+ \((\d+)Mhz\)", txt)
        pw = re.findall(r"Power \(W\): ([\d.]+)", txt)
        out.append((sclk[0] if sclk else None, pw[0] if pw else None, txt if not (sclk and pw) else ""))
        time.sleep(0.2)


def case(name, fn, flops):
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
    mid = samples[1:-1] or samples
    clk = [int(s[0]) for s in mid if s[0]]
    pw = [float(s[1]) for s in mid if s[1]]
    print(f"{name:28s} {flops*n/el/1e12:7.1f} TFLOP/s  sclk MHz {min(clk) if clk else '?'}-{max(clk) if clk else '?'} "
          f"(mean {sum(clk)/len(clk) if clk else 0:.0f})  power W {min(pw) if pw else '?'}-{max(pw) if pw else '?'} "
          f"(mean {sum(pw)/len(pw) if pw else 0:.0f})  samples {len(mid)}", flush=True)
    if not clk or not pw:
        print("raw sample:\n" + (samples[0][2] if samples else "none"), flush=True)


def main():
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    rn = lambda *s: torch.randn(*s, device=dev, generator=g).to(torch.bfloat16)
    idle_stop, idle = threading.Event(), []
    th = threading.Thread(target=poll, args=(idle_stop, idle)); th.start(); time.sleep(1.5); idle_stop.set(); th.join()
    print("idle:", [(s[0], s[1]) for s in idle], flush=True)
    if idle and idle[0][2]:
        print(idle[0][2])
    for name, N, Kc in (("wo  [M,3072]x[768,3072]^T", 768, 3072), ("qkv [M,768]x[2304,768]^T", 2304, 768)):
        a, b = rn(M, Kc), rn(N, Kc)
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        fl = 2.0 * M * N * Kc
        case("ours   " + name[:4], lambda: K.gemm(a, b, out), fl)
        case("vendor " + name[:4], lambda: torch.matmul(a, b.t(), out=out), fl)
        z = torch.zeros_like(a); zb = torch.zeros_like(b)
        case("ours, zero operands " + name[:4], lambda: K.gemm(z, zb, out), fl)
    x = rn(M, 768); y = torch.empty_like(x); w = rn(768)
    case("rmsnorm fwd (HBM-bound)", lambda: K.rmsnorm_fwd(x, w, y), 0.0)


if __name__ == "__main__":
    main()
