#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
for prio in 0 1; do
for d in 0 2 4 6 8 12; do
  QUDA_AMD_P2P_PACK_PRIO=$prio QUDA_AMD_P2P_SITE_DELAY=$d python3 tools/subvolume_timing.py 8 > gpurun_out/sub16.log 2>&1 || { tail -5 gpurun_out/sub16.log; exit 1; }
  echo "prio $prio delay $d: $(tail -1 gpurun_out/sub16.log)"
done
done
QUDA_AMD_TIMELINE=1 QUDA_AMD_P2P_SITE_DELAY=6 python3 tools/subvolume_timing.py 8 > gpurun_out/tl_call16_8.log 2>&1 && cat gpurun_out/tl_call16_8.log
QUDA_AMD_P2P_SITE_DELAY=4 python3 tools/subvolume_timing.py 4 2>&1 | tail -1
QUDA_AMD_P2P_SITE_DELAY=8 python3 tools/subvolume_timing.py 4 2>&1 | tail -1
