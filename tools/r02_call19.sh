#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
V="p2p_fold=0+edge_first=0,p2p_fold=0+edge_first=1,p2p_fold=1+edge_first=1"
for prec in 8 4 2; do
  python3 tools/subvolume_timing.py $prec 32,16,16,16 14 "$V" 4 > gpurun_out/sub19_$prec.log 2>&1 || { tail -5 gpurun_out/sub19_$prec.log; exit 1; }
  tail -1 gpurun_out/sub19_$prec.log
done
QUDA_AMD_TIMELINE=1 QUDA_AMD_P2P_FOLD=0 python3 tools/subvolume_timing.py 8 > gpurun_out/tl_call19_8.log 2>&1 && cat gpurun_out/tl_call19_8.log
python3 -m pytest tests/test_dslash_gpu.py -x -q > gpurun_out/pytest_call19.log 2>&1; rc=$?; echo "pytest dslash rc=$rc"; tail -5 gpurun_out/pytest_call19.log
