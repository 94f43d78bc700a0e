#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
python3 -m pytest tests/test_mg_gpu.py -x -q -k "block or mfma or hierarchy" > gpurun_out/pytest_call23.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 gpurun_out/pytest_call23.log
[ $rc -eq 0 ] || exit 1
python3 tools/c5_single_gpu.py --no-extras > gpurun_out/c5_call23.log 2>&1; echo "c5 rc=$?"; grep "^{" gpurun_out/c5_call23.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['setup_secs_all'], d['iters'], d['true_res']); print(json.dumps(d['coarse_block_mfma']))"
