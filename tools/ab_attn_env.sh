#!/bin/bash
# per-kernel durations of tools/bench_attn.py under values of one environment switch: ab_attn_env.sh NAME v1 v2 ...
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
name=$1; shift
for v in "$@" "$@"; do
  export $name=$v
  rm -rf /tmp/prof_env
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_env -o t -- python3 $R/tools/bench_attn.py > /dev/null 2>&1 || exit 1
  echo "== $name=$v"
  python - <<PY
import csv, glob
f = glob.glob("/tmp/prof_env/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "enc_attn" in r["Name"] and "true, false" in r["Name"]:
        print(f"  {r['Name'][:56]:56s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:8.1f} us  min {float(r['MinNs'])/1e3:8.1f}")
PY
done
