#!/bin/bash
# the no-grad forwards' grid limit (PACED_NOGRAD_GRID: 0 = none, 224 = the default on 256 compute units) on the other
# configurations of the baseline, in ONE job
run() { # name, args...
  name=$1; shift
  for r in 1 2; do for g in 0 224; do
    PACED_NOGRAD_GRID=$g timeout -k 10 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --skip-mining "$@" 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$name grid $g round $r', round(d['ms_per_step'],2), 'ms', round(d['value'],1), 'tuples/s', round(d['step_mfma_frac'],3), flush=True)" || exit 1
  done; done
}
run "t5-base K=8 S=128 B=64" --model t5-base --batch 64 --n_neg 8 --seq 128
run "t5-large K=32 S=256 B=32" --model t5-large --batch 32 --n_neg 32 --seq 256
run "t5-small K=16 S=256 B=32" --model t5-small --batch 32 --n_neg 16 --seq 256
