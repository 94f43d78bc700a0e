#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
python3 -m pytest tests/test_mg_gpu.py -x -q --durations=8 > gpurun_out/pytest_call14.log 2>&1; rc=$?; echo "pytest mg rc=$rc"; tail -16 gpurun_out/pytest_call14.log
[ $rc -eq 0 ] || exit 1
QUDA_AMD_MG_PROFILE=1 python3 tools/c5_single_gpu.py --no-extras > gpurun_out/c5_call14.log 2>&1; echo "c5 rc=$?"; grep "MG level\|^{" gpurun_out/c5_call14.log | cut -c1-1400
QUDA_AMD_NULL_FULL=1 QUDA_AMD_MG_PROFILE=1 python3 tools/c5_single_gpu.py --no-extras > gpurun_out/c5_call14_full.log 2>&1; echo "c5 (full-operator null vectors) rc=$?"; grep "MG level 1\|^{" gpurun_out/c5_call14_full.log | cut -c1-700
