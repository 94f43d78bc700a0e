#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
bash tools/bench_rehearsal.sh 2 gpurun_out/bench_reh2.log --steps 50 --warmup 5; echo "rehearsal 2 exit $?"; grep "^{" gpurun_out/bench_reh2.log | cut -c1-900; grep "rehearsal rc" gpurun_out/bench_reh2.log
bash tools/bench_rehearsal.sh 4 gpurun_out/bench_reh4.log --steps 50 --warmup 5; echo "rehearsal 4 exit $?"; grep "^{" gpurun_out/bench_reh4.log | cut -c1-900; grep "rehearsal rc" gpurun_out/bench_reh4.log
bash tools/bench_rehearsal.sh 4 gpurun_out/bench_reh4c3.log --steps 50 --warmup 5 --lattice 32,32,32,64; echo "rehearsal 4 configs[3] exit $?"; grep "^{" gpurun_out/bench_reh4c3.log | cut -c1-700; grep "rehearsal rc" gpurun_out/bench_reh4c3.log
bash tools/mgpu_rehearsal.sh 2 > gpurun_out/mgpu_reh2.log 2>&1; echo "mgpu rehearsal 2 exit $?"; tail -3 gpurun_out/mgpu_reh2.log | cut -c1-200
