#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
python3 -m pytest tests/test_mg_gpu.py -x -q -k "verify or hierarchy or block" > gpurun_out/pytest_call8.log 2>&1; rc=$?; echo "pytest mg rc=$rc"; tail -5 gpurun_out/pytest_call8.log
[ $rc -eq 0 ] || exit 1
for n in 8 24; do
QUDA_AMD_BLOCK_FINE_NRHS=$n QUDA_AMD_MG_PROFILE=1 python3 tools/c5_single_gpu.py --no-extras > gpurun_out/c5_call8_$n.log 2>&1; echo "c5 nrhs=$n rc=$?"; grep "MG level 1\|setup_secs" gpurun_out/c5_call8_$n.log | cut -c1-400
done
export TMPDIR=/tmp
out=$PWD/gpurun_out/prof_c5_setup
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o s -- python3 tools/c5_single_gpu.py --no-extras > gpurun_out/c5_call8.log 2>&1; echo "rc=$?"
python3 - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/prof_c5_setup/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print("%-100s calls %6s avg %10.1f us tot %8.1f ms %s%%"%(r['Name'][:100],r['Calls'],float(r['AverageNs'])/1e3,float(r['TotalDurationNs'])/1e6,r['Percentage']))
PY
