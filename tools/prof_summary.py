#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel-trace CSV: time per kernel symbol and per (symbol, grid) group."""
import collections, csv, glob, re, sys
d = sys.argv[1]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
sym = collections.defaultdict(lambda: [0, 0.0])
grp = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    n = re.sub(r"\(.*", "", n)[:70]
    us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    sym[n][0] += 1; sym[n][1] += us
    k = (n, int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), r["Grid_Size_Y"], r["Grid_Size_Z"])
    grp[k][0] += 1; grp[k][1] += us
tot = sum(v[1] for v in sym.values())
print(f"total kernel time {tot/1e3:.1f} ms over {steps:g} steps = {tot/1e3/steps:.1f} ms/step")
for n, (c, us) in sorted(sym.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"{n:72s} calls/step={c/steps:7.1f} ms/step={us/1e3/steps:8.2f} avg_us={us/c:9.1f} {100*us/tot:5.1f}%")
print("--- by grid")
for k, (c, us) in sorted(grp.items(), key=lambda kv: -kv[1][1])[:24]:
    print(f"{k[0][:48]:48s} wgs={k[1]:7d} y={k[2]:>3s} z={k[3]:>4s} n/step={c/steps:6.1f} avg_us={us/c:9.1f} ms/step={us/1e3/steps:7.2f}")
