#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
python3 -m pytest tests/test_mg_gpu.py -x -q > gpurun_out/pytest_tmp.log 2>&1; rc=$?; echo "pytest mg rc=$rc"; tail -3 gpurun_out/pytest_tmp.log
[ $rc -eq 0 ] || exit 1
QUDA_AMD_MG_PROFILE=1 python3 tools/c5_single_gpu.py --no-extras > gpurun_out/c5_tmp.log 2>&1; echo "rc=$?"; grep "coarse operator 48^2 x 9 per site on 12\|^{" gpurun_out/c5_tmp.log | cut -c1-260
