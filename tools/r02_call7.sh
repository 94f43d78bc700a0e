#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
out=$PWD/gpurun_out/prof_c5_setup
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o s -- python3 tools/c5_single_gpu.py --no-extras > gpurun_out/c5_call7.log 2>&1; echo "rc=$?"
grep "MG level" gpurun_out/c5_call7.log
python3 - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/prof_c5_setup/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:22]:
    print("%-100s calls %6s avg %10.1f us tot %8.1f ms %s%%"%(r['Name'][:100],r['Calls'],float(r['AverageNs'])/1e3,float(r['TotalDurationNs'])/1e6,r['Percentage']))
PY
cp $(ls gpurun_out/prof_c5_setup/*/*kernel_stats.csv | head -1) gpurun_out/r02c_c5_blocksetup_kernel_stats.csv
