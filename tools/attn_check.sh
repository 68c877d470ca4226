#!/bin/bash
# the encoder attention tests, then per-kernel durations of tools/bench_attn.py for ab/libpaced_<name>.so (every name given) and the in-tree build ("new"), twice
timeout -k 10 600 python -u -m pytest tests/test_kernels_gpu.py -x -q -k "enc_attention" --timeout 300 --timeout-method thread 2>&1 | tail -5 || exit 1
bash tools/ab_attn_kernels.sh "$@" && bash tools/ab_attn_kernels.sh "$@"
