#!/bin/bash
# the other configurations of the baseline through bench.py (not bench lines), then the headline one, and a 100-step run of it; ONE job
run() { # name, args...
  name=$1; shift
  timeout -k 10 400 python bench.py --no-cpu-baseline --skip-mining "$@" 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$name:', round(d['ms_per_step'],2), 'ms/step', round(d['value'],1), 'tuples/s', 'step_mfma_frac', round(d['step_mfma_frac'],3), flush=True)" || exit 1
}
run "t5-base K=8 S=128 B=64" --steps 6 --warmup 2 --model t5-base --batch 64 --n_neg 8 --seq 128
run "t5-large K=32 S=256 B=32" --steps 4 --warmup 2 --model t5-large --batch 32 --n_neg 32 --seq 256
run "t5-small K=16 S=256 B=32" --steps 6 --warmup 2 --model t5-small --batch 32 --n_neg 16 --seq 256
run "t5-small K=2 S=256 B=4 (configs[0])" --steps 6 --warmup 2 --model t5-small --batch 4 --n_neg 2 --seq 256
run "headline, 5 steps" --steps 5 --warmup 2
run "headline, 100 steps" --steps 100 --warmup 2
