#!/bin/bash
# per-launch durations of the pool sweep's kernels for one Q (kernel trace of tools/bench_small_q.py --qs Q --iters 1)
Q=${1:-64}; O=gpurun_out/${2:-smallq_trace}; mkdir -p $O; R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/pst -o pst -- python3 $R/tools/bench_small_q.py --qs $Q --iters 1 > /dev/null 2> $R/$O/trace.err || { echo trace failed; tail -n 5 $R/$O/trace.err; exit 1; }
cd $R; python - <<'PY'
import csv, glob
f = glob.glob("/tmp/pst/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "pool_stream" in r["Kernel_Name"] or "topk_" in r["Kernel_Name"] or "gemm" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
prev_end = None
for r in rows[:140]:
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (st - prev_end) / 1e3 if prev_end else 0.0
    print(f"{(st - t0) / 1e3:10.1f} us  +{gap:6.1f} gap  {(en - st) / 1e3:8.1f} us  grid {r['Grid_Size_X']:>8s}  {r['Kernel_Name'][:60]}")
    prev_end = en
PY
