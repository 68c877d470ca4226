#!/bin/bash
# ab/libpaced_<name>.so from the in-tree objects with attention.hip rebuilt under extra flags: build_attn_variant.sh NAME [-DFOO=1 ...]
name=$1; shift
mkdir -p ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -fno-gpu-rdc -Wno-unused-result "$@" -c pacednegatives_amd/csrc/attention.hip -o ab/attn_$name.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC pacednegatives_amd/lib/obj/{api,gemm,ops,mining}.o ab/attn_$name.o -o ab/libpaced_$name.so && rm ab/attn_$name.o && echo built ab/libpaced_$name.so
