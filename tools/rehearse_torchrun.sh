#!/bin/bash
# The driver's N > 1 command line (torch.distributed.run) on ONE GPU: two ranks share cuda:0 and talk over gloo.
# Run three times in a row; a hang dumps every thread's stack after 60 s (PACED_HANG_DUMP) and fails the script.
for i in 1 2 3; do
  PACED_HANG_DUMP=60 PACED_DIST_BACKEND=gloo timeout -k 10 150 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
      --master-addr 127.0.0.1 --master-port $((29540 + i)) bench.py --gpus 2 --steps 2 --warmup 1 --model t5-tiny --batch 4 --n_neg 4 --seq 64 \
      --mine_pool 400000 --mine_queries 64 --mine_k 100 --no-cpu-baseline > gpurun_out/torchrun_$i.json 2> gpurun_out/torchrun_$i.err
  rc=$?
  echo "run $i rc $rc json lines $(grep -c '^{' gpurun_out/torchrun_$i.json)"
  if [ $rc -ne 0 ]; then grep -v "^$" gpurun_out/torchrun_$i.err | tail -40; exit 1; fi
done
