#!/bin/bash
# step time of bench.py under several settings of one environment variable, alternating, in ONE job:
#   ab_env_multi.sh VAR rounds value1 value2 ...   (extra fixed settings: EXTRA_ENV="A=1 B=2")
var=$1; rounds=$2; shift 2
for r in $(seq 1 $rounds); do for val in "$@"; do
  env $EXTRA_ENV $var=$val timeout -k 10 200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --skip-mining 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$var=$val round $r', round(d['ms_per_step'],2), flush=True)" || exit 1
done; done
