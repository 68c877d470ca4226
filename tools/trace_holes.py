#!/usr/bin/env python3
"""From a rocprofv3 kernel trace (csv), the share of wall time in which 0, 1, 2.. kernels were running, and the large
kernels that ran alone -- where a second stream could still add work.  usage: trace_holes.py kernel_trace.csv [ms_before_end_start ms_before_end_stop]"""
import csv, sys
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
lo, hi = rows[0][0], max(r[1] for r in rows)
w0, w1 = (hi - float(sys.argv[2]) * 1e6, hi - float(sys.argv[3]) * 1e6) if len(sys.argv) > 3 else (lo, hi)
ev = []
for s, e, _ in rows:
    s, e = max(s, w0), min(e, w1)
    if e > s:
        ev += [(s, 1), (e, -1)]
ev.sort()
depth, last, hist = 0, w0, {}
for t, d in ev:
    hist[depth] = hist.get(depth, 0) + (t - last)
    depth += d
    last = t
hist[0] = hist.get(0, 0) + (w1 - last)
tot = w1 - w0
print(f"window {tot/1e6:.2f} ms")
for k in sorted(hist):
    print(f"  {k} kernels running: {100*hist[k]/tot:5.1f}%  {hist[k]/1e6:8.2f} ms")
big = {}
alone = {}
act = []
for s, e, n in rows:
    if e < w0 or s > w1:
        continue
    act.append((s, e, n))
# time each kernel family spends as the only running kernel
pts = sorted(set([a for a, _, _ in act] + [b for _, b, _ in act]))
import bisect
starts = sorted(act)
for i in range(len(pts) - 1):
    a, b = pts[i], pts[i + 1]
    live = [n for s, e, n in act if s <= a and e >= b]
    if len(live) == 1:
        k = live[0].split("(")[0][:60]
        alone[k] = alone.get(k, 0) + (b - a)
print("running alone (ms):")
for k, v in sorted(alone.items(), key=lambda kv: -kv[1])[:14]:
    print(f"  {v/1e6:8.2f}  {k}")
