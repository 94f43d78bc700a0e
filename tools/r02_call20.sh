#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
V="edge_first=0,edge_first=1"
for prec in 8 4 2; do
  python3 tools/subvolume_timing.py $prec 32,16,16,16 14 "$V" 3 > gpurun_out/sub20_$prec.log 2>&1 || { tail -5 gpurun_out/sub20_$prec.log; exit 1; }
  tail -1 gpurun_out/sub20_$prec.log
done
