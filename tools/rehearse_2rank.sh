#!/bin/bash
# Rehearsal of bench.py's multi-rank path on ONE GPU: bench.py starts two rank processes itself; they share cuda:0 and talk
# over gloo (RCCL refuses two ranks on one device).  Covers the launcher, rendezvous, sharding, both all-reduces of the
# step, the sharded pool sweep with its gather + merge, the barrier/MAX timing and the rank-0 JSON line.
PACED_DIST_BACKEND=gloo timeout -k 10 200 python bench.py --gpus 2 --steps 2 --warmup 1 --model t5-tiny --batch 4 --n_neg 4 --seq 64 \
    --mine_pool 400000 --mine_queries 64 --mine_k 100 --no-cpu-baseline > gpurun_out/b_2rank.json 2> gpurun_out/b_2rank.err
echo rc $?
head -c 600 gpurun_out/b_2rank.json
