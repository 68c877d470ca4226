#!/bin/bash
# SQ counters of the encoder attention kernels at the benchmark shape -> gpurun_out/pmc_sq_attention/table.txt
bash tools/pmc_sq_attention.sh > /dev/null 2>&1 || { echo "counter passes failed"; cat gpurun_out/pmc_sq_attention/p1.err | tail -5; exit 1; }
python tools/pmc_table.py gpurun_out/pmc_sq_attention/p1 gpurun_out/pmc_sq_attention/p2 > gpurun_out/pmc_sq_attention/table.txt
rm -rf gpurun_out/pmc_sq_attention/p1 gpurun_out/pmc_sq_attention/p2
cat gpurun_out/pmc_sq_attention/table.txt
