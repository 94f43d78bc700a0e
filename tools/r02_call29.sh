#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
python3 tools/transfer_timing.py 32,32,32,32 2>&1 | grep "^[RP] \|setup"
python3 tools/transfer_timing.py 48,48,48,96 2>&1 | grep "^[RP] \|setup"
python3 -m pytest tests/test_mg_gpu.py -x -q > gpurun_out/pytest_call29.log 2>&1; rc=$?; echo "pytest mg rc=$rc"; tail -4 gpurun_out/pytest_call29.log
