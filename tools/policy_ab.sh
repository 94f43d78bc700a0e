#!/bin/bash
# A/B of stencil builds / field paddings at 48^3 x 96 (GPU box, repo root): tools/policy_ab.sh "<lib names>" "<pads>" "<precs>"
libs=${1:-"base"}; pads=${2:-"0"}; precs=${3:-"8 4 2"}
for lib in $libs; do for pad in $pads; do for prec in $precs; do
  if [ "$lib" = base ]; then unset QUDA_AMD_LIBRARY; else export QUDA_AMD_LIBRARY=$PWD/quda-qkxtm-multigrid_amd/lib/libquda_$lib.so; fi
  export QUDA_AMD_FIELD_PAD=$pad
  line=$(timeout -k 10 200 python3 bench.py --no-cpu --no-extra --lattice ${LAT:-48,48,48,96} --fast-gauge --prec $prec --steps 100 --warmup 5 2>/dev/null | tail -1)
  echo "${LAT:-48,48,48,96} $lib pad=$pad prec=$prec $(python3 -c "import json,sys; d=json.loads(sys.argv[1]); print(d['roofline']['kernel_us'], d['roofline']['frac'], d['value'])" "$line")"
done; done; done
