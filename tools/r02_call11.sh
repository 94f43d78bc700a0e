#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
python3 tools/mg_kappa_scan.py 32,32,32,32 0.35 0.145,0.147,0.149,0.151 0.001 40000 gpurun_out/r02_mg_kappa_scan_32x4_c.json > gpurun_out/kappa_scan_c.log 2>&1; echo "scan rc=$?"; cat gpurun_out/kappa_scan_c.log | cut -c1-600
( time python3 bench.py ) > gpurun_out/bench_call11.log 2>&1; echo "bench rc=$?"; tail -5 gpurun_out/bench_call11.log | cut -c1-6000
