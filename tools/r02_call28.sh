#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
python3 -m pytest tests/test_mg_gpu.py -x -q > gpurun_out/pytest_call28.log 2>&1; rc=$?; echo "pytest mg rc=$rc"; tail -4 gpurun_out/pytest_call28.log
[ $rc -eq 0 ] || exit 1
for v in 0 1 2 3; do QUDA_AMD_PROLONG_VAR=$v python3 tools/transfer_timing.py 32,32,32,32 2>&1 | grep "^[RP] \|setup"; done
QUDA_AMD_MG_PROFILE=1 python3 tools/c5_single_gpu.py --no-extras > gpurun_out/c5_call28.log 2>&1; echo "c5 rc=$?"; grep "MG level\|^{" gpurun_out/c5_call28.log | cut -c1-600
python3 -m pytest tests/test_dslash_gpu.py -x -q -k "full_size" > gpurun_out/pytest_call28b.log 2>&1; echo "pytest dslash full size rc=$?"; tail -4 gpurun_out/pytest_call28b.log
