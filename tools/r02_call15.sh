#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
QUDA_AMD_MG_PROFILE=1 python3 tools/c5_single_gpu.py --no-extras > gpurun_out/c5_call15.log 2>&1; rc=$?; echo "c5 rc=$rc"; grep "MG level\|^{" gpurun_out/c5_call15.log | cut -c1-900
[ $rc -eq 0 ] || exit 1
python3 -m pytest tests/test_mg_gpu.py -x -q > gpurun_out/pytest_call15.log 2>&1; rc=$?; echo "pytest mg rc=$rc"; tail -4 gpurun_out/pytest_call15.log
[ $rc -eq 0 ] || exit 1
for prec in 8 4; do
  QUDA_AMD_TIMELINE=1 python3 tools/subvolume_timing.py $prec > gpurun_out/tl_call15_$prec.log 2>&1 && cat gpurun_out/tl_call15_$prec.log || exit 1
  python3 tools/subvolume_timing.py $prec > gpurun_out/sub_call15_$prec.log 2>&1 && tail -1 gpurun_out/sub_call15_$prec.log || exit 1
done
