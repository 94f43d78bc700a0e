#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
python3 -m pytest tests/test_mg_gpu.py -x -q -k "verify or hierarchy or block" > gpurun_out/pytest_call13.log 2>&1; rc=$?; echo "pytest mg rc=$rc"; tail -3 gpurun_out/pytest_call13.log
[ $rc -eq 0 ] || exit 1
( time QUDA_AMD_MG_PROFILE=1 python3 bench.py ) > gpurun_out/bench_call13.log 2>&1; echo "bench rc=$?"; grep "MG level 1\|^real" gpurun_out/bench_call13.log | cut -c1-300
