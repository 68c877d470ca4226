#!/bin/bash
# the headline step with an environment switch at given values and UNSET (the package's default), alternating, ONE job: knob_default.sh NAME v1 ...
name=$1; shift
for r in 1 2 3; do for v in "$@" unset; do
  if [ $v = unset ]; then unset $name; else export $name=$v; fi
  timeout -k 10 200 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --skip-mining 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$name=$v round $r', round(d['ms_per_step'],2), flush=True)" || exit 1
done; done
