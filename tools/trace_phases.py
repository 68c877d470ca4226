#!/usr/bin/env python3
"""Per step of a rocprofv3 kernel trace of bench.py: when the main chain's phases start and end (encoder forward,
decoder forward + backward, encoder backward, update), found from marker kernels, and how much kernel time of other
queues ran inside each phase.  usage: trace_phases.py kernel_trace.csv"""
import csv, sys, collections
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
t0 = rows[0][0]
ad = [r for r in rows if "adamw_kernel" in r[2]]
# the model update (the large adamw launch) ends a step
big = [r for r in ad if r[1] - r[0] > 300_000]
print("steps found:", len(big))
for i in range(1, len(big)):
    a, b = big[i - 1][1], big[i][1]
    step = [r for r in rows if r[0] >= a and r[1] <= b]
    save_fwd = [r for r in step if "gemm_phased_kernel<false, false, 2>" in r[2]]       # wi products: 24 no-grad + 24... both passes
    dself_b = [r for r in step if "dec_self_attn_bwd" in r[2]]
    dself_f = [r for r in step if "dec_self_attn_fwd" in r[2]]
    ebwd = [r for r in step if "enc_attn_bwd_dkdv" in r[2]]
    print(f"step {i}: {(b - a) / 1e6:.2f} ms; kernels {len(step)}")
    if dself_f and dself_b and ebwd:
        q_main = dself_b[0][3]
        dec_f0 = min(r[0] for r in dself_f if r[3] == q_main) if any(r[3] == q_main for r in dself_f) else dself_f[0][0]
        dec_b1 = max(r[1] for r in dself_b)
        print(f"   decoder phases of the chain that keeps state: start +{(dec_f0 - a) / 1e6:.2f} ms, end +{(dec_b1 - a) / 1e6:.2f} ms "
              f"({(dec_b1 - dec_f0) / 1e6:.2f} ms); encoder backward first attention +{(ebwd[0][0] - a) / 1e6:.2f}, last +{(ebwd[-1][1] - a) / 1e6:.2f}")
        byq = collections.Counter()
        for r in step:
            s, e = max(r[0], dec_f0), min(r[1], dec_b1)
            if e > s:
                byq[r[3]] += e - s
        print("   kernel time inside the decoder window by queue (ms):", {k: round(v / 1e6, 2) for k, v in byq.items()})
