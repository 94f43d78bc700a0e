#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
python3 -m pytest tests/test_mg_gpu.py -x -q -k "block or hierarchy" > gpurun_out/pytest_call5a.log 2>&1; rc=$?; echo "pytest block rc=$rc"; tail -5 gpurun_out/pytest_call5a.log
[ $rc -eq 0 ] || exit 1
python3 -m pytest tests/test_qkxtm_gpu.py tests/test_dropin_gpu.py tests/test_layout_gpu.py -q > gpurun_out/pytest_call5b.log 2>&1; rc=$?; echo "pytest qkxtm/dropin rc=$rc"; tail -15 gpurun_out/pytest_call5b.log
QUDA_AMD_MG_PROFILE=1 python3 tools/c5_single_gpu.py --no-extras > gpurun_out/c5_call5.log 2>&1; echo "c5 rc=$?"; tail -4 gpurun_out/c5_call5.log | cut -c1-2500
