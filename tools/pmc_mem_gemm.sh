# Memory-path counters of the phased GEMM on the forward shapes of the benchmark: two rocprofv3 --pmc passes (run from the repository root).
R=$PWD
O=$R/gpurun_out/pmc_mem_gemm
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export TILES=256 ONLY=${ONLY:-NT}
timeout -s KILL 150 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ_LATENCY TCP_TCC_READ_REQ TCP_TOTAL_CACHE_ACCESSES TA_TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES TD_TD_BUSY TD_TC_STALL --output-format csv -d $O/p1 -- python3 $R/tools/bench_gemm.py > /dev/null 2> $O/p1.err || { tail -5 $O/p1.err; exit 1; }
timeout -s KILL 150 rocprofv3 --pmc SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_CYCLES SQ_INSTS_VMEM_WR TCP_TCR_TCP_STALL_CYCLES TCP_RFIFO_STALL_CYCLES TCP_LFIFO_STALL_CYCLES TCP_TCP_LATENCY TA_DATA_STALLED_BY_TC_CYCLES TA_BUFFER_TOTAL_CYCLES TD_LOAD_WAVEFRONT TD_SPI_STALL --output-format csv -d $O/p2 -- python3 $R/tools/bench_gemm.py > /dev/null 2> $O/p2.err || { tail -5 $O/p2.err; exit 1; }
python3 $R/tools/pmc_table.py $O/p1 $O/p2 > $O/table.txt
cat $O/table.txt
