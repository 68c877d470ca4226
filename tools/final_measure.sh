#!/bin/bash
# Full measurement on one GPU box: GPU tests, smoke, HBM-traffic counters (two separate --pmc passes), kernel trace,
# and the default bench line.  Every GPU step runs under its own time limit; the script stops at the first failure.
set -o pipefail
mkdir -p gpurun_out/final
O=gpurun_out/final
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > $O/tests.log 2>&1 || { tail -n 30 $O/tests.log; exit 1; }
tail -n 2 $O/tests.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1 || { tail -n 30 $O/smoke.log; exit 1; }
tail -n 1 $O/smoke.log
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$O/pmc_fetch -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --skip-mining > $R/$O/pmc_fetch.json 2> $R/$O/pmc_fetch.err || { echo "pmc fetch failed"; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/$O/pmc_write -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --skip-mining > $R/$O/pmc_write.json 2> $R/$O/pmc_write.err || { echo "pmc write failed"; exit 1; }
python3 $R/tools/pmc_summary.py $R/$O/pmc_fetch $R/$O/pmc_write $R/$O/pmc_traffic.json > $R/$O/pmc_summary.txt || exit 1
cp $R/$O/pmc_traffic.json $R/profiles/r1_pmc_traffic.json
rm -rf $R/$O/pmc_fetch $R/$O/pmc_write
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace -o r1 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --skip-mining > $R/$O/trace_bench.json 2> $R/$O/trace.err || { echo "kernel trace failed"; exit 1; }
rm -f $R/$O/trace/r1_kernel_trace.csv
cd $R
timeout -k 10 900 python bench.py > $O/bench.json 2> $O/bench.err || { tail -n 20 $O/bench.err; exit 1; }
cat $O/bench.json
