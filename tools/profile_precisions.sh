#!/bin/bash
set -e
export TMPDIR=/tmp
for p in 4 2; do
  out=$PWD/gpurun_out/prof_prec$p
  rm -rf $out; mkdir -p $out
  B="python3 bench.py --no-cpu --no-extra --steps 20 --warmup 2 --prec $p"
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o s -- $B > $out/stats.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -o f -- $B > $out/fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -o w -- $B > $out/write.log 2>&1
  python3 - <<PY
import csv,glob,collections
def cnt(d):
    agg=collections.defaultdict(list)
    for f in glob.glob("$out/"+d+"/**/*counter_collection.csv",recursive=True):
        for r in csv.DictReader(open(f)): agg[(r["Kernel_Name"][:60],r["Counter_Name"])].append(float(r["Counter_Value"]))
    return agg
f=cnt("fetch"); w=cnt("write")
for (k,c),v in list(f.items())+list(w.items()):
    if "dslash_kernel" in k or "Norm2" in k: print($p,k,c,len(v),sum(v)/len(v))
for r in csv.DictReader(open(glob.glob("$out/stats/**/*kernel_stats.csv",recursive=True)[0])):
    if "dslash_kernel" in r["Name"]: print($p,"avg ns",r["AverageNs"])
PY
done
