#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
python3 -m pytest tests/test_mg_gpu.py -x -q -k "verify or hierarchy or block" > gpurun_out/pytest_call10.log 2>&1; rc=$?; echo "pytest mg rc=$rc"; tail -5 gpurun_out/pytest_call10.log
[ $rc -eq 0 ] || exit 1
QUDA_AMD_MG_PROFILE=1 python3 tools/c5_single_gpu.py --no-extras > gpurun_out/c5_call10.log 2>&1; echo "c5 rc=$?"; grep "MG level\|setup_secs" gpurun_out/c5_call10.log | cut -c1-330
python3 tools/mg_kappa_scan.py 32,32,32,32 0.35 0.131,0.134,0.137,0.140,0.143 0.002 30000 gpurun_out/r02_mg_kappa_scan_32x4_b.json > gpurun_out/kappa_scan_b.log 2>&1; echo "scan rc=$?"; cat gpurun_out/kappa_scan_b.log | cut -c1-600
