#!/bin/bash
# Q=64 pool sweep with several builds of the library, alternating, in one job: ab_small_q.sh NAME...   ("new" = in-tree build)
for r in 1 2; do for lib in "$@" new; do
  if [ $lib = new ]; then unset PACED_HIP_LIB; else export PACED_HIP_LIB=$PWD/ab/libpaced_$lib.so; fi
  timeout -k 10 200 python tools/bench_small_q.py --qs 64 --iters 5 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); r=d['runs'][0]; print('$lib round $r: Q=64 %.3f ms/sweep  hbm_frac %.3f' % (r['ms_per_sweep'], r['hbm_frac']))" || exit 1
done; done
