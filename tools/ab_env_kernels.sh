#!/bin/bash
# per-kernel averages of the in-order step (one stream) under two settings of one environment variable: ab_env_kernels.sh VAR a b
var=$1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export PACED_OVERLAP_FORWARDS=0
for val in $2 $3; do
  export $var=$val
  rm -rf /tmp/prof_$val
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$val -o t -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --skip-mining > /dev/null 2>&1 || exit 1
  echo "== $var=$val"
  python3 - <<PY
import csv, glob
f = glob.glob("/tmp/prof_$val/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"  all kernels {tot/3e6:.2f} ms per step")
for r in rows[:14]:
    print(f"  {r['Name'][:60]:60s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:8.1f} us  total/step {float(r['TotalDurationNs'])/3e6:7.2f} ms")
PY
done
