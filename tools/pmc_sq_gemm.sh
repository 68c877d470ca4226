# SQ-block counters of the phased GEMM on the forward shapes of the benchmark: two rocprofv3 --pmc passes (run from the repository root).
R=$PWD
mkdir -p $R/gpurun_out/pmc_sq_gemm
cd /tmp && export TMPDIR=/tmp
export TILES=256 ONLY=NT
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INSTS_VALU --output-format csv -d $R/gpurun_out/pmc_sq_gemm/p1 -- python3 $R/tools/bench_gemm.py > /dev/null 2> $R/gpurun_out/pmc_sq_gemm/p1.err || { tail -5 $R/gpurun_out/pmc_sq_gemm/p1.err; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM --output-format csv -d $R/gpurun_out/pmc_sq_gemm/p2 -- python3 $R/tools/bench_gemm.py > /dev/null 2> $R/gpurun_out/pmc_sq_gemm/p2.err || { tail -5 $R/gpurun_out/pmc_sq_gemm/p2.err; exit 1; }
echo done
