#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
QUDA_AMD_P2P_FOLD=1 python3 -m pytest tests/test_dslash_gpu.py -x -q -k "partitioned or two_process or verified" > gpurun_out/pytest_call35.log 2>&1; echo "pytest (fold=1) rc=$?"; tail -3 gpurun_out/pytest_call35.log
QUDA_AMD_GALERKIN_FULL=1 QUDA_AMD_NULL_ORTHO=gs QUDA_AMD_PROLONG_XGROUP=0 python3 -m pytest tests/test_mg_gpu.py -x -q > gpurun_out/pytest_call35b.log 2>&1; echo "pytest mg (reference-shaped setup switches) rc=$?"; tail -3 gpurun_out/pytest_call35b.log
( time python3 bench.py ) > gpurun_out/bench_call35.log 2>&1; echo "bench rc=$?"; grep "^{" gpurun_out/bench_call35.log > gpurun_out/r02_bench_line.json; python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r02_bench_line.json'))
e=d['extra']
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline']['traffic_source'])
for k,v in e.items():
    if k.startswith('mg_'): print(k, v.get('setup_secs_all'), v.get('solve_secs'), v.get('iters'), v.get('true_res'), (v.get('plain_gcr') or {}).get('iters'))
    else: print(k, json.dumps(v)[:600])
print(json.dumps(d['cpu_baseline'])[:600])
PY
tail -4 gpurun_out/bench_call35.log | grep real
