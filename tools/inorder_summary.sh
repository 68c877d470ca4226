#!/bin/bash
# Kernel trace of the one-stream schedule, summarised per kernel symbol (4 traced steps: 1 warm-up + 3 timed).
R=$PWD
mkdir -p $R/gpurun_out/inorder
cd /tmp && export TMPDIR=/tmp
export PACED_OVERLAP_FORWARDS=0
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/inorder -o io -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --skip-mining > $R/gpurun_out/inorder/bench.json 2> $R/gpurun_out/inorder/err.txt || { tail -5 $R/gpurun_out/inorder/err.txt; exit 1; }
python3 $R/tools/prof_summary.py $R/gpurun_out/inorder 4 > $R/gpurun_out/inorder/summary.txt
rm -f $R/gpurun_out/inorder/*kernel_trace.csv
cat $R/gpurun_out/inorder/summary.txt
