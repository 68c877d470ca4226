#!/bin/bash
# pool sweep at small Q: timings, then a kernel trace of the same program
O=gpurun_out/${1:-smallq}; mkdir -p $O; R=$PWD
timeout -k 10 300 python tools/bench_small_q.py > $O/small_q.json 2> $O/small_q.err || { tail -n 20 $O/small_q.err; exit 1; }
cat $O/small_q.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace -o sq -- python3 $R/tools/bench_small_q.py --iters 1 > /dev/null 2> $R/$O/trace.err || { echo trace failed; tail -n 5 $R/$O/trace.err; exit 1; }
rm -f $R/$O/trace/*/*kernel_trace.csv $R/$O/trace/*kernel_trace.csv
cd $R; python - $O <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/trace/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:12]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>6s} total {float(r['TotalDurationNs'])/1e6:9.2f} ms avg {float(r['AverageNs'])/1e3:8.1f} us")
PY
