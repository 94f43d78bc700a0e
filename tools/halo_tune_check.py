#!/usr/bin/env python3
"""The partitioned stencil on the 8-GPU sub-lattice (32x16x16x16, y z t through the self-neighbour emulation, atom wire format) with the
launch-parameter sweep of the tune cache switched off and on (QudaInvertParam.tune): does the sweep find anything the heuristics miss?"""
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402

qa = importlib.import_module("quda-qkxtm-multigrid_amd")
qa.init(0)
Xs = [32, 16, 16, 16]
gs = bench.make_gauge(Xs) if hasattr(bench, "make_gauge") else None
if gs is None:
    from synth import tiled_gauge
    gs = tiled_gauge(Xs)
hs = np.random.default_rng(1).random(int(np.prod(Xs)) // 2 * 24)
out = {}
for prec in (8, 4, 2):
    row = {}
    for tune in (0, 1):
        qa.lib().qudaAmdSetPartitionMask(0b1110)
        qa.lib().qudaAmdSetDslashTune(b"halo_format", 1)
        qa.load_gauge(gs, qa.gauge_param(Xs, cuda_prec=prec))
        ip = qa.invert_param(qa.QUDA_TWISTED_MASS_DSLASH, 0.1, 0.01, +1, "ee", 0, cuda_prec=prec)
        ip.tune = 1 if tune else 0
        src, dst = qa.Spinor(prec), qa.Spinor(prec)
        src.load(hs, ip)
        d = qa.Dirac(ip, pc=True)
        if tune:
            x = qa.dslash(hs, ip, 0)      # dslashQuda: sets the tuning switch from ip.tune and sweeps the key once
        d.time_dslash(dst, src, 0, 50)
        row["tune_%d_us" % tune] = round(1e6 * min(d.time_dslash(dst, src, 0, 500) for _ in range(3)), 2)
        src.free(); dst.free(); d.free()
    out[prec] = row
qa.lib().qudaAmdSetPartitionMask(0)
print("HALOTUNE " + json.dumps(out))
qa.end()
