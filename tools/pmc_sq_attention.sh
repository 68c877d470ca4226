# SQ-block counters of the encoder attention kernels: two rocprofv3 --pmc passes of eight counters each (run from the repository root).
R=$PWD
mkdir -p $R/gpurun_out/pmc_sq_attention
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 90 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INSTS_VALU --output-format csv -d $R/gpurun_out/pmc_sq_attention/p1 -- python3 $R/tools/attn_once.py > /dev/null 2> $R/gpurun_out/pmc_sq_attention/p1.err || { tail -5 $R/gpurun_out/pmc_sq_attention/p1.err; exit 1; }
timeout -s KILL 90 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_CYCLES SQ_INSTS_SALU --output-format csv -d $R/gpurun_out/pmc_sq_attention/p2 -- python3 $R/tools/attn_once.py > /dev/null 2> $R/gpurun_out/pmc_sq_attention/p2.err || { tail -5 $R/gpurun_out/pmc_sq_attention/p2.err; exit 1; }
find $R/gpurun_out/pmc_sq_attention -name "*counter_collection.csv" | head
