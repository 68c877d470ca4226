for r in 1 2; do
  for lib in old new; do
    if [ $lib = old ]; then export PACED_HIP_LIB=${OLD_LIB:-$PWD/gpurun_ab/libpaced_old.so}; else unset PACED_HIP_LIB; fi
    echo "== $lib round $r"
    TILES=256 timeout -k 10 120 python tools/bench_gemm.py 2>&1 | grep -v amdgpu.ids | cut -c1-64 || exit 1
  done
done
