#!/bin/bash
# usage (GPU box, repo root): tools/profile_multisrc.sh <tag> [L_s L_t nsrc outer]  ->  profiles/<tag>_multisrc_table.json + gpurun_out/<tag>_multisrc.log
# kernel-by-kernel table of ONE lockstep invertMultiSrcQuda solve (marker-bracketed; every launch joined with its algorithmic bytes)
set -e
tag=$1; Ls=${2:-32}; Lt=${3:-32}; ns=${4:-12}; outer=${5:-pc}
export TMPDIR=/tmp
out=$PWD/gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out profiles
QA_PROFILE_MARKERS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o t -- python3 tools/multisrc_timing.py $Ls $Lt $ns $outer $out/acct.json > $out/run.log 2>&1
grep "^SOLVE" $out/run.log > gpurun_out/${tag}_multisrc.log
secs=$(python3 -c "import json,sys; print(json.loads(open('gpurun_out/${tag}_multisrc.log').read()[6:])['solver_secs'])")
python3 tools/summarize_solve_trace.py $(find $out -name "*kernel_trace.csv" | head -1) $out/acct.json profiles/${tag}_multisrc_table.json $secs >> gpurun_out/${tag}_multisrc.log 2>&1
cp profiles/${tag}_multisrc_table.json gpurun_out/ 2>/dev/null || true
cat gpurun_out/${tag}_multisrc.log
