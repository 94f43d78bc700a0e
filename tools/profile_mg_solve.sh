#!/bin/bash
# usage (GPU box, repo root): tools/profile_mg_solve.sh <tag> [L_s L_t] [tm|tmc] [V|K]  ->  profiles/<tag>_mg_solve_table.json + gpurun_out/<tag>_mg_solve.log
set -e
tag=$1; Ls=${2:-48}; Lt=${3:-96}; act=${4:-tm}; cyc=${5:-V}
export TMPDIR=/tmp
out=$PWD/gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out profiles
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o t -- python3 tools/mg_solve_profile.py $Ls $Lt $act $out/acct.json $cyc > $out/run.log 2>&1
grep "^SOLVE" $out/run.log > gpurun_out/${tag}_mg_solve.log
secs=$(python3 -c "import json,sys; print(json.loads(open('gpurun_out/${tag}_mg_solve.log').read()[6:])['solver_secs'])")
python3 tools/summarize_solve_trace.py $(find $out -name "*kernel_trace.csv" | head -1) $out/acct.json profiles/${tag}_mg_solve_table.json $secs >> gpurun_out/${tag}_mg_solve.log 2>&1
cp $(find $out -name "*kernel_stats.csv" | head -1) profiles/${tag}_mg_solve_whole_run_kernel_stats.csv
cp profiles/${tag}_mg_solve_table.json profiles/${tag}_mg_solve_whole_run_kernel_stats.csv gpurun_out/ 2>/dev/null || true
cat gpurun_out/${tag}_mg_solve.log
