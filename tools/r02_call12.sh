#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
python3 -m pytest tests/test_mg_gpu.py -x -q -k "stalls" -s > gpurun_out/pytest_call12.log 2>&1; rc=$?; echo "pytest rc=$rc"; grep "critical kappa\|passed\|failed" gpurun_out/pytest_call12.log
S="tiled=2;tiled=2,lds_pad=40000;tiled=2,lds_pad=54000;tiled=2,block=128;tiled=2,block=64;tiled=2,block=128,lds_pad=27000;tiled=2,block=128,lds_pad=40000;tiled=2,block=64,lds_pad=20000;tiled=2,block=64,lds_pad=13000;tiled=2,nxz=8,tz=3;tiled=2,nxz=8,tz=2;tiled=2,nxz=4"
python3 tools/dslash_sweep.py 48,48,48,96 "8:tm" "$S" 20 > gpurun_out/sweep48c.log 2>&1; cat gpurun_out/sweep48c.log | cut -c1-150
( time python3 bench.py ) > gpurun_out/bench_call12.log 2>&1; echo "bench rc=$?"; tail -4 gpurun_out/bench_call12.log | cut -c1-1200
