#!/usr/bin/env python3
"""MG-GCR on the twisted-clover operator at 32^4 (bench.py extra.mg_gcr_tmc) as a stand-alone run; QUDA_AMD_BLOCK_FINE=0 in the
environment gives the sequential null-vector solves for comparison, QUDA_AMD_MG_PROFILE=1 the stage times of the set-up."""
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
from synth import smooth_gauge  # noqa: E402

qa = importlib.import_module("quda-qkxtm-multigrid_amd")
qa.init(0)
L = int(sys.argv[1]) if len(sys.argv) > 1 else 32
X = (L, L, L, L)
g = smooth_gauge(X, 0.35)
for dslash in (sys.argv[2:] or ["tmc", "tm"]):
    print(dslash, json.dumps(bench.run_mg(qa, X, gauge=g, dslash=dslash, coarse_bench=False, extras=False)), flush=True)
qa.end()
