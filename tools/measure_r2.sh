#!/bin/bash
# Full measurement on one GPU box (the profiles/r2_* set): GPU tests, smoke, HBM-traffic counters for the training step and for the
# pool sweep (two separate --pmc passes each), kernel traces (default schedule, one-stream schedule, pool sweep), and
# the default bench line.  Every GPU step runs under its own time limit; the script stops at the first failure.
set -o pipefail
O=gpurun_out/r2final
mkdir -p $O
R=$PWD
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $O/tests.log 2>&1 || { tail -n 30 $O/tests.log; exit 1; }
tail -n 2 $O/tests.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1 || { tail -n 30 $O/smoke.log; exit 1; }
tail -n 1 $O/smoke.log
cd /tmp && export TMPDIR=/tmp
# ---- training step: HBM traffic per kernel
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$O/pmc_fetch -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --skip-mining > $R/$O/pmc_fetch.json 2> $R/$O/pmc_fetch.err || { echo "pmc fetch failed"; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/$O/pmc_write -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --skip-mining > $R/$O/pmc_write.json 2> $R/$O/pmc_write.err || { echo "pmc write failed"; exit 1; }
python3 $R/tools/pmc_summary.py $R/$O/pmc_fetch $R/$O/pmc_write $R/$O/pmc_traffic.json > $R/$O/pmc_summary.txt || exit 1
rm -rf $R/$O/pmc_fetch $R/$O/pmc_write
# ---- pool sweep: HBM traffic (1 warm-up + 1 timed sweep = 2 sweeps)
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$O/mpmc_fetch -- python3 $R/tools/mine_once.py --mine_iters 1 > $R/$O/mpmc_fetch.txt 2> $R/$O/mpmc_fetch.err || { echo "mining pmc fetch failed"; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/$O/mpmc_write -- python3 $R/tools/mine_once.py --mine_iters 1 > $R/$O/mpmc_write.txt 2> $R/$O/mpmc_write.err || { echo "mining pmc write failed"; exit 1; }
python3 $R/tools/pmc_mining.py $R/$O/mpmc_fetch $R/$O/mpmc_write $R/$O/pmc_traffic_mining.json 2 > $R/$O/pmc_summary_mining.txt || exit 1
rm -rf $R/$O/mpmc_fetch $R/$O/mpmc_write
# ---- kernel traces
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace -o r2 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --skip-mining > $R/$O/trace_bench.json 2> $R/$O/trace.err || { echo "kernel trace failed"; exit 1; }
rm -f $R/$O/trace/r2_kernel_trace.csv
PACED_OVERLAP_FORWARDS=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace_inorder -o r2io -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --skip-mining > $R/$O/trace_inorder_bench.json 2> $R/$O/trace_inorder.err || { echo "in-order kernel trace failed"; exit 1; }
rm -f $R/$O/trace_inorder/r2io_kernel_trace.csv
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace_mining -o r2m -- python3 $R/tools/mine_once.py --mine_iters 3 > $R/$O/trace_mining.txt 2> $R/$O/trace_mining.err || { echo "mining kernel trace failed"; exit 1; }
rm -f $R/$O/trace_mining/r2m_kernel_trace.csv
cd $R
cp $O/pmc_traffic.json profiles/r2_pmc_traffic.json
cp $O/pmc_traffic_mining.json profiles/r2_pmc_traffic_mining.json
timeout -k 10 900 python bench.py > $O/bench.json 2> $O/bench.err || { tail -n 20 $O/bench.err; exit 1; }
cat $O/bench.json
# ---- encoder attention: per-kernel durations and memory-path counters (two --pmc passes)
bash tools/ab_attn_kernels.sh > $O/attn_kernels.txt 2>&1 || { tail -n 5 $O/attn_kernels.txt; exit 1; }
cd $R
bash tools/pmc_mem_attention.sh > /dev/null 2>&1 || { echo "attention counters failed"; exit 1; }
cd $R
cp gpurun_out/pmc_mem_attention/table.txt $O/pmc_mem_attention.txt
