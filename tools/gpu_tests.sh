#!/bin/bash
# GPU test suite on the box, output under gpurun_out/$1 (default: tests): usage  gpurun -- bash tools/gpu_tests.sh NAME [pytest args]
O=gpurun_out/${1:-tests}; shift
mkdir -p $O
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 400 --timeout-method thread -s "$@" > $O/tests.log 2>&1
rc=$?
tail -n 30 $O/tests.log | cut -c1-300
exit $rc
