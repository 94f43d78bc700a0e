#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
QUDA_AMD_MG_PROFILE=1 python3 tools/c5_single_gpu.py --no-extras > gpurun_out/c5_call9.log 2>&1; echo "c5 rc=$?"; grep "MG level\|setup_secs" gpurun_out/c5_call9.log | cut -c1-330
python3 tools/mg_kappa_scan.py 32,32,32,32 0.35 0.124,0.1255,0.1265,0.1275,0.1285 0.005 30000 gpurun_out/r02_mg_kappa_scan_32x4.json > gpurun_out/kappa_scan.log 2>&1; echo "scan rc=$?"; cat gpurun_out/kappa_scan.log | cut -c1-600
