#!/bin/bash
# the headline step under a few settings of one environment switch, alternating, in ONE job: knob_sweep.sh NAME v1 v2 ...
name=$1; shift
for r in 1 2 3; do for v in "$@"; do
  env $name=$v timeout -k 10 200 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --skip-mining 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$name=$v round $r', round(d['ms_per_step'],2), flush=True)" || exit 1
done; done
