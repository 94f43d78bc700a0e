#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
python3 -m pytest tests/test_mg_gpu.py -x -q > gpurun_out/pytest_call6.log 2>&1; rc=$?; echo "pytest mg rc=$rc"; tail -15 gpurun_out/pytest_call6.log
[ $rc -eq 0 ] || exit 1
QUDA_AMD_MG_PROFILE=1 python3 tools/c5_single_gpu.py --no-extras > gpurun_out/c5_call6.log 2>&1; echo "c5 rc=$?"; grep -v "^MG profile" gpurun_out/c5_call6.log | tail -14 | cut -c1-1500
