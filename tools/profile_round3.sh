#!/bin/bash
# Round-3 profile set (GPU box, repo root): kernel stats + PMC traffic of the bench configuration and of the 48^3 x 96 stencils, and the
# kernel-by-kernel table of one warmed C5 MG-GCR solve (fp32 V-cycle; fp16 storage with QUDA_AMD_MG_HALF=1).  Summaries land in profiles/ and are
# copied to gpurun_out/ so that they travel back.
set -e
bash tools/profile_case.sh r03_bench_fp64_tm_32x4 524288 > gpurun_out/r03_prof_bench.log 2>&1
bash tools/profile_case.sh r03_tm_f64_48x48x48x96 5308416 --lattice 48,48,48,96 --fast-gauge --prec 8 > gpurun_out/r03_prof_48_f64.log 2>&1
bash tools/profile_case.sh r03_tm_f32_48x48x48x96 5308416 --lattice 48,48,48,96 --fast-gauge --prec 4 > gpurun_out/r03_prof_48_f32.log 2>&1
bash tools/profile_mg_solve.sh r03c_c5_vcycle 48 96 tm V > gpurun_out/r03_prof_c5_v.log 2>&1
QUDA_AMD_MG_HALF=1 bash tools/profile_mg_solve.sh r03d_c5_vcycle_half 48 96 tm V > gpurun_out/r03_prof_c5_vh.log 2>&1
cp profiles/r03* gpurun_out/ 2>/dev/null || true
tail -3 gpurun_out/r03_prof_bench.log; head -3 gpurun_out/r03c_c5_vcycle_mg_solve.log | cut -c1-400; head -3 gpurun_out/r03d_c5_vcycle_half_mg_solve.log | cut -c1-400
