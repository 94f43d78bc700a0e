#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
V="p2p_fold=1,p2p_fold=0,p2p_fold=0+pack_prio=1,p2p_fold=0+site_delay=2"
for prec in 8 4 2; do
  python3 tools/subvolume_timing.py $prec 32,16,16,16 14 "$V" 4 > gpurun_out/sub18_$prec.log 2>&1 || { tail -5 gpurun_out/sub18_$prec.log; exit 1; }
  tail -1 gpurun_out/sub18_$prec.log
done
QUDA_AMD_TIMELINE=1 QUDA_AMD_P2P_FOLD=0 python3 tools/subvolume_timing.py 8 > gpurun_out/tl_call18_8.log 2>&1 && cat gpurun_out/tl_call18_8.log
