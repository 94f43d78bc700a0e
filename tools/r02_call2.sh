#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
python3 -m pytest tests/test_dslash_gpu.py -x -q > gpurun_out/pytest_call2.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 gpurun_out/pytest_call2.log
[ $rc -eq 0 ] || exit 1
S="order=1;tiled=1,nxz=8,tt=1;tiled=2,nxz=8,tt=1;tiled=1,nxz=8,tt=1,store_aux=2;tiled=2,nxz=8,tt=1,store_aux=2"
S="$S;order=1,hop_order=1;tiled=1,nxz=8,tt=1,hop_order=1;tiled=2,nxz=8,tt=1,hop_order=1;tiled=1,nxz=8,tt=1,store_aux=2,hop_order=1;tiled=2,nxz=8,tt=1,store_aux=2,hop_order=1"
S="$S;tiled=1,nxz=8,tt=2,store_aux=2,hop_order=1;tiled=2,nxz=8,tt=2,store_aux=2,hop_order=1;tiled=2,nxz=4,tt=1,store_aux=2,hop_order=1;tiled=2,nxz=1,tt=1,store_aux=2,hop_order=1;tiled=2,nxz=2,tt=1,store_aux=2,hop_order=1"
S="$S;tiled=2,nxz=8,tt=1,store_aux=2,hop_order=1,block=128;tiled=2,nxz=8,tt=1,store_aux=2,hop_order=1,block=64;order=1,store_aux=2,hop_order=1;remap=0,hop_order=1,store_aux=2;order=0,hop_order=1,store_aux=2"
python3 tools/dslash_sweep.py 48,48,48,96 "4:tm,8:tm,2:tm" "$S" 20 > gpurun_out/sweep48b.log 2>&1 || { tail -20 gpurun_out/sweep48b.log; exit 1; }
python3 tools/dslash_sweep.py 32,32,32,32 "8:tm,4:tm,2:tm,8:tmc,4:tmc,2:tmc" "$S" 50 > gpurun_out/sweep32b.log 2>&1 || { tail -20 gpurun_out/sweep32b.log; exit 1; }
python3 tools/dslash_sweep.py 32,16,16,16 "8:tm,4:tm,2:tm" "$S" 200 > gpurun_out/sweep_sub.log 2>&1 || { tail -20 gpurun_out/sweep_sub.log; exit 1; }
tail -4 gpurun_out/sweep_sub.log
