#!/bin/bash
# link-load cache policies A/B on one box, interleaved and repeated (the run-to-run spread is as large as the effects): fp64 and fp32, 48^3 x 96 and 32^4
run() { # lib lattice prec
  if [ "$1" = base ]; then unset QUDA_AMD_LIBRARY; else export QUDA_AMD_LIBRARY=$PWD/quda-qkxtm-multigrid_amd/lib/libquda_$1.so; fi
  line=$(timeout -k 10 200 python3 bench.py --no-cpu --no-extra --lattice $2 --fast-gauge --prec $3 --steps 100 --warmup 5 2>/dev/null | tail -1)
  echo "$2 prec=$3 $1 $(python3 -c "import json,sys; d=json.loads(sys.argv[1]); print(d['roofline']['kernel_us'], d['roofline']['frac'])" "$line")"
}
for rep in 1 2 3; do for lat in 48,48,48,96 32,32,32,32; do for prec in 8 4; do for lib in base g18 g17 g19; do run $lib $lat $prec; done; done; done; done
