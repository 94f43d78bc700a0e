#!/bin/bash
# round-2 GPU call 1: block-order sweeps + PMC evidence of the "before" state
set -o pipefail
mkdir -p gpurun_out
S48="block=256,order=1;block=256,order=2;block=256,order=3;block=256,order=4;block=256,order=6;order=1;order=2;order=3;order=4;order=6;block=128,order=2;block=128,order=4"
S48="$S48;tiled=1,nxz=1,tz=48,tt=2;tiled=1,nxz=1,tz=8,tt=2;tiled=1,nxz=1,tz=4,tt=4;tiled=1,nxz=1,tz=6,tt=3;tiled=1,nxz=1,tz=2,tt=6;tiled=1,nxz=1,tz=12,tt=12;tiled=1,nxz=1,tz=4,tt=2;tiled=1,nxz=1,tz=2,tt=2"
S48="$S48;tiled=1,nxz=8,tz=6,tt=1;tiled=1,nxz=8,tz=6,tt=2;tiled=1,nxz=8,tz=6,tt=4;tiled=1,nxz=8,tz=3,tt=2;tiled=1,nxz=8,tz=2,tt=4;tiled=1,nxz=8,tz=6,tt=8"
S48="$S48;tiled=1,nxz=2,tz=6,tt=2;tiled=1,nxz=2,tz=8,tt=4;tiled=1,nxz=2,tz=4,tt=3;tiled=1,nxz=4,tz=12,tt=2;tiled=1,nxz=4,tz=6,tt=2;tiled=1,nxz=4,tz=4,tt=4;tiled=1,nxz=4,tz=3,tt=3"
S48="$S48;block=128,tiled=1,nxz=1,tz=4,tt=4;block=128,tiled=1,nxz=8,tz=6,tt=2;block=64,tiled=1,nxz=1,tz=4,tt=4"
S48="$S48;order=4,lds_pad=40000;order=4,lds_pad=54000;order=4,lds_pad=81000;tiled=1,nxz=1,tz=4,tt=4,lds_pad=40000;tiled=1,nxz=1,tz=4,tt=4,lds_pad=54000;order=4,store_aux=2"
python3 tools/dslash_sweep.py 48,48,48,96 "4:tm,8:tm,2:tm" "$S48" 20 > gpurun_out/sweep48.log 2>&1 || { tail -20 gpurun_out/sweep48.log; exit 1; }
S32="order=1;order=2;order=0;remap=0;tiled=1,nxz=1,tz=32,tt=2;tiled=1,nxz=1,tz=32,tt=4;tiled=1,nxz=1,tz=8,tt=2;tiled=1,nxz=1,tz=8,tt=4;tiled=1,nxz=1,tz=4,tt=4;tiled=1,nxz=1,tz=4,tt=2;tiled=1,nxz=1,tz=16,tt=4"
S32="$S32;tiled=1,nxz=8,tz=4,tt=2;tiled=1,nxz=8,tz=4,tt=4;tiled=1,nxz=8,tz=4,tt=1;tiled=1,nxz=2,tz=4,tt=4;tiled=1,nxz=2,tz=8,tt=2;tiled=1,nxz=4,tz=8,tt=2;tiled=1,nxz=4,tz=4,tt=4;block=128,order=1;block=128,tiled=1,nxz=1,tz=4,tt=4"
S32="$S32;order=1,lds_pad=40000;order=1,lds_pad=54000;order=1,lds_pad=81000;order=1,store_aux=2"
python3 tools/dslash_sweep.py 32,32,32,32 "8:tm,4:tm,2:tm,8:tmc,4:tmc,2:tmc" "$S32" 50 > gpurun_out/sweep32.log 2>&1 || { tail -20 gpurun_out/sweep32.log; exit 1; }
tail -5 gpurun_out/sweep32.log
export QUDA_AMD_DSLASH_BLOCK=256
tools/profile_case.sh r02a_before_tm_f32_48x48x48x96 5308416 --lattice 48,48,48,96 --prec 4 --fast-gauge && \
tools/profile_case.sh r02a_before_tm_f64_48x48x48x96 5308416 --lattice 48,48,48,96 --prec 8 --fast-gauge && \
tools/profile_case.sh r02a_before_tmc_i16_32x4 524288 --prec 2 --dslash tmc --fast-gauge && \
tools/profile_case.sh r02a_before_tmc_f32_32x4 524288 --prec 4 --dslash tmc --fast-gauge
unset QUDA_AMD_DSLASH_BLOCK
python3 -m pytest tests -m gpu -x -q > gpurun_out/pytest_call1.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_call1.log
