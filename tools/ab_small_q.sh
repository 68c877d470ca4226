#!/bin/bash
# pool sweep at few queries with several builds of the library, alternating, in one job: [QS=64,128] ab_small_q.sh NAME...   ("new" = in-tree build)
QS=${QS:-64}
for r in 1 2; do for lib in "$@" new; do
  if [ $lib = new ]; then unset PACED_HIP_LIB; else export PACED_HIP_LIB=$PWD/ab/libpaced_$lib.so; fi
  timeout -k 10 200 python tools/bench_small_q.py --qs $QS --iters 5 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)
for r in d['runs']: print('$lib round $r: %s Q=%d %.3f ms/sweep  hbm_frac %.3f' % (r['entry'], r['Q'], r['ms_per_sweep'], r['hbm_frac']))" || exit 1
done; done
