#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
python3 -m pytest tests/test_dslash_gpu.py -x -q > gpurun_out/pytest_call22.log 2>&1; rc=$?; echo "pytest dslash rc=$rc"; tail -3 gpurun_out/pytest_call22.log
[ $rc -eq 0 ] || exit 1
( time python3 bench.py ) > gpurun_out/bench_call22.log 2>&1; echo "bench rc=$?"; grep "^{" gpurun_out/bench_call22.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
e=d.pop('extra',{}) if 'extra' in d else d.get('config',{}).pop('extra',{})
print(json.dumps(d)[:1500])
for k,v in e.items(): print(k, json.dumps(v)[:900])
"; tail -4 gpurun_out/bench_call22.log | grep real
