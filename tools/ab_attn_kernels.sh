#!/bin/bash
# per-kernel durations of tools/bench_attn.py for ab/libpaced_<name>.so (every name given) and the in-tree build ("new")
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for lib in "$@" new; do
  if [ $lib = new ]; then unset PACED_HIP_LIB; else export PACED_HIP_LIB=$R/ab/libpaced_$lib.so; fi
  rm -rf /tmp/prof_$lib
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$lib -o t -- python3 $R/tools/bench_attn.py > /dev/null 2>&1 || exit 1
  echo "== $lib"
  python - <<PY
import csv, glob
f = glob.glob("/tmp/prof_$lib/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "enc_attn" in r["Name"]:
        print(f"  {r['Name'][:56]:56s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:8.1f} us  min {float(r['MinNs'])/1e3:8.1f}")
PY
done
