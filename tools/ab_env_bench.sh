#!/bin/bash
# step time of bench.py under two settings of one environment variable, alternating, in ONE job: ab_env_bench.sh VAR value_a value_b
var=$1; a=$2; b=$3
for r in 1 2 3; do for val in $a $b; do
  env $var=$val timeout -k 10 200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --skip-mining 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$var=$val round $r', round(d['ms_per_step'],2))" || exit 1
done; done
