#!/bin/bash
# usage: tools/profile_round.sh <tag>   (on the GPU box, from the repo root)
# kernel-trace/stats pass and two separate PMC passes (FETCH_SIZE, WRITE_SIZE) of the default bench command, then the
# committed summaries under profiles/ (tools/summarize_profiles.py).
set -e
tag=$1
export TMPDIR=/tmp
out=$PWD/gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
B="python3 bench.py --no-cpu --no-extra --steps 20 --warmup 2"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o s -- $B > $out/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -o f -- $B > $out/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -o w -- $B > $out/write.log 2>&1
python3 tools/summarize_profiles.py $tag $out/stats $out/fetch $out/write > $out/summary.log 2>&1
cp profiles/${tag}_* gpurun_out/ 2>/dev/null || true
tail -30 $out/summary.log
