#!/bin/bash
# round-end check on a GPU box: the whole GPU suite, then the default bench line
set -o pipefail
mkdir -p gpurun_out
( time python3 -m pytest tests -x -q -m gpu --durations=6 ) > gpurun_out/pytest_final.log 2>&1; rc=$?; echo "pytest gpu rc=$rc"; tail -14 gpurun_out/pytest_final.log
[ $rc -eq 0 ] || exit 1
( time python3 bench.py ) > gpurun_out/bench_final.log 2>&1; echo "bench rc=$?"; grep "^{" gpurun_out/bench_final.log > gpurun_out/bench_line_final.json; python3 - <<'PY'
import json
d=json.load(open('gpurun_out/bench_line_final.json'))
e=d['extra']
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline']['traffic_source'])
for k,v in e.items():
    if k.startswith('mg_'): print(k, v.get('setup_secs_all'), v.get('solve_secs'), v.get('iters'), v.get('true_res'), (v.get('plain_gcr') or {}).get('iters'))
    else: print(k, json.dumps(v)[:500])
PY
tail -4 gpurun_out/bench_final.log | grep real
