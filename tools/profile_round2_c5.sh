#!/bin/bash
# kernel stats of the C5 run (two hierarchy builds, 4 plain-GCR and 4 MG-GCR solves at 48^3 x 96 on one GPU) and of the transfer timing
set -o pipefail
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
out=$PWD/gpurun_out/prof_c5g; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o s -- python3 tools/c5_single_gpu.py --no-extras > gpurun_out/prof_c5g.log 2>&1 || { tail -20 gpurun_out/prof_c5g.log; exit 1; }
cp $(find $out -name "*kernel_stats.csv" | head -1) gpurun_out/r02g_c5_48x48x48x96_one_gpu_kernel_stats.csv
grep "^{" gpurun_out/prof_c5g.log | cut -c1-330
python3 tools/transfer_timing.py 48,48,48,96 2>&1 | grep "^[RP] \|setup" | tee gpurun_out/r02g_transfer_timing_48x48x48x96.txt
