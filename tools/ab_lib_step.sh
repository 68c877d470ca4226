#!/bin/bash
# The training step with several builds of the library, alternating, in one job: ab_lib_step.sh NAME...   ("new" = in-tree build)
for r in 1 2 3 4; do for lib in "$@" new; do
  if [ $lib = new ]; then unset PACED_HIP_LIB; else export PACED_HIP_LIB=$PWD/ab/libpaced_$lib.so; fi
  timeout -k 10 200 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --skip-mining 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$lib round $r', round(d['ms_per_step'],2))" || exit 1
done; done
