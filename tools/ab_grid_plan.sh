#!/bin/bash
# step time of bench.py for several (grid limit, capped layers) plans of the no-grad forwards, in ONE job:
#   ab_grid_plan.sh rounds grid:layers grid:layers ...
rounds=$1; shift
for r in $(seq 1 $rounds); do for plan in "$@"; do
  g=${plan%%:*}; l=${plan##*:}
  PACED_NOGRAD_GRID=$g PACED_NOGRAD_GRID_LAYERS=$l timeout -k 10 200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --skip-mining 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('grid $g layers $l round $r', round(d['ms_per_step'],2), flush=True)" || exit 1
done; done
