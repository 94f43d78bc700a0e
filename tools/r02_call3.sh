#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
tools/profile_case.sh r02b_tm_f32_48x48x48x96 5308416 --lattice 48,48,48,96 --prec 4 --fast-gauge && \
tools/profile_case.sh r02b_tm_f64_48x48x48x96 5308416 --lattice 48,48,48,96 --prec 8 --fast-gauge && \
tools/profile_case.sh r02b_tmc_i16_32x4 524288 --prec 2 --dslash tmc --fast-gauge && \
tools/profile_case.sh r02b_tmc_f32_32x4 524288 --prec 4 --dslash tmc --fast-gauge && \
tools/profile_case.sh r02b_bench_fp64_tm_32x4 524288 --fast-gauge
python3 -m pytest tests -m gpu -x -q > gpurun_out/pytest_call3.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_call3.log
