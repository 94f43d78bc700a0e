#!/bin/bash
# usage: tools/profile_case.sh <tag> <VH> <bench.py arguments ...>   (on the GPU box, from the repo root)
# kernel-trace/stats pass and separate PMC passes (FETCH_SIZE; WRITE_SIZE; L2 hit/miss) of one bench.py configuration;
# summaries -> profiles/<tag>_* (tools/summarize_profiles.py; VH = checkerboard sites, for the FETCH_SIZE calibration).
set -e
tag=$1; shift
export VH=$1; shift
export TMPDIR=/tmp
out=$PWD/gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o s -- python3 bench.py --no-cpu --no-extra --steps 20 --warmup 2 "$@" > $out/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -o f -- python3 bench.py --no-cpu --no-extra --steps 20 --warmup 2 "$@" > $out/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -o w -- python3 bench.py --no-cpu --no-extra --steps 20 --warmup 2 "$@" > $out/write.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/l2 -o l -- python3 bench.py --no-cpu --no-extra --steps 20 --warmup 2 "$@" > $out/l2.log 2>&1 || true
python3 tools/summarize_profiles.py $tag $out/stats $out/fetch $out/write $out/l2 > $out/summary.log 2>&1
cp profiles/${tag}_* gpurun_out/ 2>/dev/null || true
tail -40 $out/summary.log
