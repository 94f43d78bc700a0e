#!/bin/bash
# round-2 final profiles: kernel stats + PMC traffic of the bench kernel and of the cases VERDICT item 1 names, C5 kernel stats
set -o pipefail
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
bash tools/profile_case.sh r02f_bench_fp64_tm_32x4 524288 > gpurun_out/prof27_a.log 2>&1 || { tail -20 gpurun_out/prof27_a.log; exit 1; }
echo "a done"; tail -3 gpurun_out/prof27_a.log
bash tools/profile_case.sh r02f_tmc_i16_32x4 524288 --prec 2 --dslash tmc > gpurun_out/prof27_b.log 2>&1 || { tail -20 gpurun_out/prof27_b.log; exit 1; }
echo "b done"
bash tools/profile_case.sh r02f_tmc_f32_32x4 524288 --prec 4 --dslash tmc > gpurun_out/prof27_c.log 2>&1 || { tail -20 gpurun_out/prof27_c.log; exit 1; }
echo "c done"
bash tools/profile_case.sh r02f_tm_i16_32x4 524288 --prec 2 > gpurun_out/prof27_d.log 2>&1 || { tail -20 gpurun_out/prof27_d.log; exit 1; }
echo "d done"
bash tools/profile_case.sh r02f_tm_f32_48x48x48x96 5308416 --prec 4 --lattice 48,48,48,96 --fast-gauge > gpurun_out/prof27_e.log 2>&1 || { tail -20 gpurun_out/prof27_e.log; exit 1; }
echo "e done"
bash tools/profile_case.sh r02f_tm_f64_48x48x48x96 5308416 --prec 8 --lattice 48,48,48,96 --fast-gauge > gpurun_out/prof27_f.log 2>&1 || { tail -20 gpurun_out/prof27_f.log; exit 1; }
echo "f done"
out=$PWD/gpurun_out/prof_c5; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o s -- python3 tools/c5_single_gpu.py --no-extras > gpurun_out/prof27_c5.log 2>&1 || { tail -20 gpurun_out/prof27_c5.log; exit 1; }
cp $(find $out -name "*kernel_stats.csv" | head -1) gpurun_out/r02f_c5_48x48x48x96_one_gpu_kernel_stats.csv
echo "c5 done"; grep "^{" gpurun_out/prof27_c5.log | cut -c1-300
out=$PWD/gpurun_out/prof_sub; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o s -- python3 tools/subvolume_timing.py 8 > gpurun_out/prof27_sub.log 2>&1 || { tail -20 gpurun_out/prof27_sub.log; exit 1; }
cp $(find $out -name "*kernel_stats.csv" | head -1) gpurun_out/r02f_subvolume_32x16x16x16_partitioned_kernel_stats.csv
echo "sub done"; tail -1 gpurun_out/prof27_sub.log
ls gpurun_out | grep r02f
