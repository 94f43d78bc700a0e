#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
V="ghost_pipe=0,ghost_pipe=1"
for prec in 8 4 2; do
  python3 tools/subvolume_timing.py $prec 32,16,16,16 14 "$V" 3 > gpurun_out/sub21_$prec.log 2>&1 || { tail -5 gpurun_out/sub21_$prec.log; exit 1; }
  tail -1 gpurun_out/sub21_$prec.log
done
QUDA_AMD_TIMELINE=1 python3 tools/subvolume_timing.py 8 > gpurun_out/tl_call21_8.log 2>&1 && grep "ghost\|block end" gpurun_out/tl_call21_8.log
python3 -m pytest tests/test_dslash_gpu.py -x -q > gpurun_out/pytest_call21.log 2>&1; rc=$?; echo "pytest dslash rc=$rc"; tail -5 gpurun_out/pytest_call21.log
