#!/bin/bash
# Round-4 profile set (GPU box, repo root): kernel stats + PMC traffic of the bench configuration (32^4 fp64 twisted-mass stencil, 18 reals) and of the
# 8- and 12-real variants, the fp32 8-real stencil at 48^3 x 96, and the kernel-by-kernel tables of one warmed MG-GCR solve on the sub-lattice of an
# 8-GPU split (fused coarse cycle on / off) and at C5.  Summaries land in profiles/ and are copied to gpurun_out/ so that they travel back.
set -e
export TMPDIR=/tmp
bash tools/profile_case.sh r04_bench_fp64_tm_32x4 524288 > gpurun_out/r04_prof_bench.log 2>&1
bash tools/profile_case.sh r04_tm_f64_r8_32x4 524288 --recon 8 > gpurun_out/r04_prof_f64_r8.log 2>&1
bash tools/profile_case.sh r04_tm_f32_r8_48x48x48x96 5308416 --lattice 48,48,48,96 --fast-gauge --prec 4 --recon 8 > gpurun_out/r04_prof_48_f32_r8.log 2>&1
QA_PROFILE_FUSED=0 QA_PROFILE_LATTICE=32,16,16,16 QA_PROFILE_MASK=14 bash tools/profile_mg_solve.sh r04g_sub8_masked_unfused 32 32 tm V > gpurun_out/r04_prof_sub8_unfused.log 2>&1
QA_PROFILE_LATTICE=32,16,16,16 QA_PROFILE_MASK=14 bash tools/profile_mg_solve.sh r04h_sub8_masked_fused 32 32 tm V > gpurun_out/r04_prof_sub8_fused.log 2>&1
bash tools/profile_mg_solve.sh r04i_c5_vcycle 48 96 tm V > gpurun_out/r04_prof_c5_v.log 2>&1
cp profiles/r04* gpurun_out/ 2>/dev/null || true
tail -3 gpurun_out/r04_prof_bench.log; head -3 gpurun_out/r04h_sub8_masked_fused_mg_solve.log | cut -c1-400; head -3 gpurun_out/r04i_c5_vcycle_mg_solve.log | cut -c1-400
