#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
python3 -m pytest tests/test_dslash_gpu.py -x -q > gpurun_out/pytest_call25.log 2>&1; rc=$?; echo "pytest dslash rc=$rc"; tail -3 gpurun_out/pytest_call25.log
[ $rc -eq 0 ] || exit 1
( time python3 bench.py ) > gpurun_out/bench_call25.log 2>&1; echo "bench rc=$?"; grep "^{" gpurun_out/bench_call25.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
e=d.pop('extra',{}) if 'extra' in d else d.get('config',{}).pop('extra',{})
print(d['value'], d['roofline'])
for k,v in e.items():
    if 'mg_' in k: print(k, v.get('setup_secs_all'), v.get('solve_secs'), v.get('iters'))
    else: print(k, json.dumps(v)[:700])
"; tail -4 gpurun_out/bench_call25.log | grep real
