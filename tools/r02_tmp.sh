#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
python3 -m pytest tests/test_mg_gpu.py tests/test_qkxtm_gpu.py -x -q > gpurun_out/pytest_tmp.log 2>&1; rc=$?; echo "pytest mg+qkxtm rc=$rc"; tail -3 gpurun_out/pytest_tmp.log
[ $rc -eq 0 ] || exit 1
QUDA_AMD_MG_PROFILE=1 python3 tools/c5_single_gpu.py --no-extras > gpurun_out/c5_tmp.log 2>&1; echo "rc=$?"; grep "transfer (fill\|^{" gpurun_out/c5_tmp.log | cut -c1-260
