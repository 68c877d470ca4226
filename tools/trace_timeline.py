#!/usr/bin/env python3
"""From a rocprofv3 kernel trace (csv): per millisecond of the last `last_ms` ms, the share of time in which at least one
LARGE kernel (>= 60 us) was running, only small kernels were running, or nothing was; and which large kernels started in
that millisecond.  Shows where the latency-bound chains of the step run without cover.
usage: trace_timeline.py kernel_trace.csv last_ms [skip_ms]   (skip_ms: leave out that much at the end of the trace,
e.g. bench.py's instrumented in-order step)"""
import csv, sys
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))]
t_end = max(r[1] for r in rows) - (int(float(sys.argv[3]) * 1e6) if len(sys.argv) > 3 else 0)
rows = [r for r in rows if r[0] < t_end]
t0 = t_end - int(float(sys.argv[2]) * 1e6)
rows = sorted(r for r in rows if r[1] > t0)
nb = int((t_end - t0) / 1e6) + 1
big, any_ = [0.0] * nb, [0.0] * nb
names = [dict() for _ in range(nb)]


def short(n):
    n = n.split("(")[0]
    for k in ("gemm_phased", "gemm_ring", "gemm_bf16", "enc_attn_fwd", "enc_attn_bwd_dkdv", "enc_attn_bwd_dq", "rmsnorm_fwd", "rmsnorm_bwd",
              "cross2_fwd", "cross2_bwd", "adamw", "embedding", "ce_kernel"):
        if k in n:
            return k
    return n[-24:]


def cover(rs):
    """union length per ms bin of the intervals rs"""
    out = [0.0] * nb
    end = t0
    for s, e, _ in rs:
        s = max(s, end, t0)
        if e <= s:
            continue
        end = e
        b = int((s - t0) / 1e6)
        while s < e and b < nb:
            lim = t0 + (b + 1) * 1000000
            out[b] += min(e, lim) - s
            s = min(e, lim); b += 1
    return out


bigs = [r for r in rows if r[1] - r[0] >= 60000]
cb, ca = cover(bigs), cover(rows)
for s, e, n in bigs:
    b = int((max(s, t0) - t0) / 1e6)
    names[b][short(n)] = names[b].get(short(n), 0) + 1
smalls = [dict() for _ in range(nb)]
for s, e, n in rows:
    if e - s < 60000 and s >= t0:
        b = int((s - t0) / 1e6)
        smalls[b]["n"] = smalls[b].get("n", 0) + 1
print(" ms  large%  small-only%  idle%  small-launches  large kernels started")
for b in range(nb):
    print(f"{b:4d} {cb[b]/1e4:6.1f} {(ca[b]-cb[b])/1e4:8.1f} {100-ca[b]/1e4:8.1f} {smalls[b].get('n',0):8d}     "
          + " ".join(f"{k}x{v}" for k, v in sorted(names[b].items())))
