#!/bin/bash
# grid over spinor / gauge plane paddings at 48^3 x 96 on ONE box, the unpadded case repeated for drift control
run() { # lib spad gpad prec
  if [ "$1" = base ]; then unset QUDA_AMD_LIBRARY; else export QUDA_AMD_LIBRARY=$PWD/quda-qkxtm-multigrid_amd/lib/libquda_$1.so; fi
  export QUDA_AMD_FIELD_PAD=$2 QUDA_AMD_GAUGE_PAD=$3
  line=$(timeout -k 10 200 python3 bench.py --no-cpu --no-extra --lattice ${LAT:-48,48,48,96} --fast-gauge --prec $4 --steps 100 --warmup 5 2>/dev/null | tail -1)
  echo "$1 spad=$2 gpad=$3 prec=$4 $(python3 -c "import json,sys; d=json.loads(sys.argv[1]); print(d['roofline']['kernel_us'], d['roofline']['frac'])" "$line")"
}
for prec in ${PRECS:-8 4 2}; do
  run base 0 0 $prec; run ${ALT:-g18h2} 0 0 $prec
  for sp in ${SPADS:-0 672 1344 2688 5376}; do for gp in ${GPADS:-0 672 1344 2688}; do run ${LIB:-base} $sp $gp $prec; done; done
  run base 0 0 $prec; run ${ALT:-g18h2} 0 0 $prec
done
