#!/bin/bash
# usage (GPU box, repo root): tools/recon_timing.sh <out.log>  — the twisted-mass stencil with 18 / 12 / 8 real links in every precision at 32^4 and 48^3 x 96
# (bench.py lines: us per application, fraction of the 8 TB/s roofline on the ALGORITHMIC bytes 8 R P + 72 P (+ 4) per site)
out=${1:-gpurun_out/recon_timing.log}
: > $out
for lat in 32,32,32,32 48,48,48,96; do
  for pr in "8 18" "8 12" "8 8" "4 18" "4 12" "4 8" "2 18" "2 12" "2 8"; do
    set -- $pr
    python3 bench.py --lattice $lat --fast-gauge --prec $1 --recon $2 --no-extra --no-cpu --steps 100 --warmup 10 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
r=d['roofline']
print('%-12s prec %d recon %2d  %8.2f us  %6.1f GB/s  frac %.3f  %d B/site  %.0f GFLOP/s' % ('$lat', $1, $2, r['kernel_us'], r['achieved'], r['frac'], r['bytes_per_site'], d['value']))" >> $out
  done
done
cat $out
