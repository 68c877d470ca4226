#!/usr/bin/env python3
"""From a rocprofv3 kernel trace (csv): the intervals in which NO kernel was running, largest first, with the kernel
that ended before and the one that started after each; and the total per kind of boundary.
usage: trace_gaps.py kernel_trace.csv [min_gap_us [last_ms]]   (last_ms: only the last so many ms of the trace)"""
import csv, sys
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-60:]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
if len(sys.argv) > 3:
    t_end = max(r[1] for r in rows)
    rows = [r for r in rows if r[0] >= t_end - float(sys.argv[3]) * 1e6]
min_gap = float(sys.argv[2]) * 1e3 if len(sys.argv) > 2 else 5e3
gaps, end, last = [], rows[0][1], rows[0][2]
for s, e, n in rows[1:]:
    if s > end:
        gaps.append((s - end, end - rows[0][0], last, n))
    if e > end:
        end, last = e, n
tot = end - rows[0][0]
print(f"trace {tot/1e6:.2f} ms, {len(rows)} kernels, idle {sum(g[0] for g in gaps)/1e6:.3f} ms in {len(gaps)} gaps "
      f"({100*sum(g[0] for g in gaps)/tot:.2f} %)")
big = [g for g in gaps if g[0] >= min_gap]
print(f"gaps >= {min_gap/1e3:.0f} us: {len(big)}, {sum(g[0] for g in big)/1e6:.3f} ms")
for g in sorted(big, key=lambda g: -g[0])[:40]:
    print(f"  {g[0]/1e3:9.1f} us at {g[1]/1e6:9.3f} ms   after {g[2]:45s} before {g[3]}")
kinds = {}
for g in gaps:
    k = (g[2], g[3])
    v = kinds.setdefault(k, [0, 0]); v[0] += g[0]; v[1] += 1
print("by boundary (total us, count):")
for k, v in sorted(kinds.items(), key=lambda kv: -kv[1][0])[:25]:
    print(f"  {v[0]/1e3:9.1f} {v[1]:5d}   {k[0]} -> {k[1]}")
