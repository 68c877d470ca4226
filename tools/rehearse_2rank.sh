#!/bin/bash
# Rehearsal of bench.py's multi-rank path on ONE GPU: two ranks share cuda:0 and talk over gloo (RCCL refuses two
# ranks on one device).  Covers rendezvous, sharding, both all-reduces of the step, the barrier/MAX timing and the
# rank-0 JSON line; the pool-scorer leg is skipped (gloo's all_gather on device tensors does not return here).
export PACED_DIST_BACKEND=gloo WORLD_SIZE=2 MASTER_ADDR=127.0.0.1 MASTER_PORT=29512
RANK=0 LOCAL_RANK=0 timeout -s ABRT 100 python -X faulthandler bench.py --gpus 2 --steps 2 --warmup 1 --model t5-tiny --batch 4 --n_neg 4 --seq 64 --skip-mining > gpurun_out/r0.json 2> gpurun_out/r0.err &
P0=$!
RANK=1 LOCAL_RANK=1 timeout -s ABRT 100 python -X faulthandler bench.py --gpus 2 --steps 2 --warmup 1 --model t5-tiny --batch 4 --n_neg 4 --seq 64 --skip-mining > gpurun_out/r1.json 2> gpurun_out/r1.err &
P1=$!
for i in 1 2 3 4 5 6 7 8 9 10 11 12 13 14 15 16; do sleep 10; echo tick $i; if ! kill -0 $P0 2>/dev/null && ! kill -0 $P1 2>/dev/null; then break; fi; done
wait $P0; echo rc0 $?; wait $P1; echo rc1 $?
tail -25 gpurun_out/r0.err; cat gpurun_out/r0.json | head -c 400
