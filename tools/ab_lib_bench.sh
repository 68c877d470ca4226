#!/bin/bash
# One job, alternating: the RMSNorm micro-benchmark and the training step with several builds of the library
#   usage: ab_lib_bench.sh NAME...      (ab/libpaced_NAME.so; "new" = the in-tree build is always included)
for lib in "$@" new; do
  if [ $lib = new ]; then unset PACED_HIP_LIB; else export PACED_HIP_LIB=$PWD/ab/libpaced_$lib.so; fi
  echo "== $lib"; timeout -k 10 120 python tools/bench_rmsnorm.py 2>/dev/null | cut -c1-64
done
for r in 1 2 3; do for lib in "$@" new; do
  if [ $lib = new ]; then unset PACED_HIP_LIB; else export PACED_HIP_LIB=$PWD/ab/libpaced_$lib.so; fi
  timeout -k 10 200 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --skip-mining 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$lib round $r', round(d['ms_per_step'],2))" || exit 1
done; done
