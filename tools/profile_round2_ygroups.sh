#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
export QUDA_AMD_DSLASH_YGROUPS=2
bash tools/profile_case.sh r02g_yg2_tm_f64_48x48x48x96 5308416 --prec 8 --lattice 48,48,48,96 --fast-gauge > gpurun_out/prof33_a.log 2>&1 || { tail -20 gpurun_out/prof33_a.log; exit 1; }
tail -30 gpurun_out/prof33_a.log | grep -i "bytes_per_launch\|l2_hit\|mean_KiB" | head
ls gpurun_out | grep r02g
