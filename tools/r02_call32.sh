#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
python3 tools/dslash_sweep.py 48,48,48,96 "8:tm,4:tm,2:tm,8:tmc" "ygroups=0;ygroups=2;ygroups=3;ygroups=6;ygroups=-1" 30 > gpurun_out/sweep_call32.log 2>&1 || { tail -5 gpurun_out/sweep_call32.log; exit 1; }
cat gpurun_out/sweep_call32.log | tail -30
python3 tools/dslash_sweep.py 32,32,32,32 "8:tm,4:tm" "ygroups=0;ygroups=2;ygroups=-1" 50 > gpurun_out/sweep_call32b.log 2>&1; tail -8 gpurun_out/sweep_call32b.log
QUDA_AMD_DSLASH_YGROUPS=2 python3 -m pytest tests/test_dslash_gpu.py -x -q -k "golden or seeded or full_size" > gpurun_out/pytest_call32.log 2>&1; echo "pytest (ygroups=2) rc=$?"; tail -3 gpurun_out/pytest_call32.log
