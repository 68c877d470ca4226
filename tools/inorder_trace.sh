# rocprofv3 kernel trace of bench.py with the one-stream schedule (PACED_OVERLAP_FORWARDS=0): per-launch durations of kernels running alone.
R=$PWD
mkdir -p $R/gpurun_out/inorder
cd /tmp && export TMPDIR=/tmp
export PACED_OVERLAP_FORWARDS=0
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/inorder -o io -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --skip-mining > $R/gpurun_out/inorder/bench.json 2> $R/gpurun_out/inorder/err.txt || { tail -5 $R/gpurun_out/inorder/err.txt; exit 1; }
rm -f $R/gpurun_out/inorder/io_kernel_trace.csv
ls $R/gpurun_out/inorder
