#!/bin/bash
# Build a variant of the library for A/B runs in one job: ab/libpaced_<name>.so with extra compiler flags
#   usage: tools/build_variant.sh NAME [-DFOO=1 ...]      then: PACED_HIP_LIB=$PWD/ab/libpaced_NAME.so python ...
name=$1; shift
mkdir -p ab/obj_$name
for f in api gemm ops attention mining; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -fno-gpu-rdc -Wno-unused-result "$@" -c pacednegatives_amd/csrc/$f.hip -o ab/obj_$name/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC ab/obj_$name/*.o -o ab/libpaced_$name.so && rm -rf ab/obj_$name && echo built ab/libpaced_$name.so
