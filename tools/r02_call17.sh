#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
for prec in 8 4 2; do
  for fold in 1 0; do
    QUDA_AMD_P2P_FOLD=$fold python3 tools/subvolume_timing.py $prec > gpurun_out/sub17.log 2>&1 || { tail -5 gpurun_out/sub17.log; exit 1; }
    echo "fold $fold: $(tail -1 gpurun_out/sub17.log)"
  done
done
QUDA_AMD_TIMELINE=1 python3 tools/subvolume_timing.py 8 > gpurun_out/tl_call17_8.log 2>&1 && cat gpurun_out/tl_call17_8.log
python3 -m pytest tests/test_dslash_gpu.py -x -q > gpurun_out/pytest_call17.log 2>&1; rc=$?; echo "pytest dslash rc=$rc"; tail -5 gpurun_out/pytest_call17.log
