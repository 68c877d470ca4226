#!/bin/bash
# kernel trace of a short bench run, then the idle intervals of the GPU (tools/trace_gaps.py)
set -o pipefail
R=$PWD; O=$R/gpurun_out/gaps; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o g -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --skip-mining > $O/bench.json 2> $O/err.txt || { tail -n 20 $O/err.txt; exit 1; }
F=$(find $O/trace -name '*kernel_trace.csv' | head -n 1)
python3 $R/tools/trace_gaps.py $F 5 ${LAST_MS:-370} > $O/gaps.txt
python3 $R/tools/trace_timeline.py $F ${TL_MS:-130} ${TL_SKIP:-135} > $O/timeline.txt
rm -rf $O/trace
cat $O/gaps.txt
