#!/bin/bash
# A/B of builds of the library in ONE job (boxes differ by a few percent): ab/libpaced_<name>.so for every name given
# (default: base) against the in-tree build ("new").
names="${@:-base}"
for r in 1 2; do
  for lib in $names new; do
    if [ $lib = new ]; then unset PACED_HIP_LIB; else export PACED_HIP_LIB=$PWD/ab/libpaced_$lib.so; fi
    echo "== $lib round $r"
    timeout -k 10 120 python tools/bench_attn.py 2>&1 | grep -v amdgpu.ids | grep "p=0.1" || exit 1
  done
done
