#!/bin/bash
# A/B of two builds of the library in ONE job (boxes differ by a few percent): ab/libpaced_base.so against the in-tree build.
for r in 1 2; do
  for lib in base new; do
    if [ $lib = base ]; then export PACED_HIP_LIB=$PWD/ab/libpaced_base.so; else unset PACED_HIP_LIB; fi
    echo "== $lib round $r"
    timeout -k 10 120 python tools/bench_attn.py 2>&1 | grep -v amdgpu.ids | grep "p=0.1" || exit 1
  done
done
