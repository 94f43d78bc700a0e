#!/usr/bin/env python3
"""Streaming yardstick of the BLAS layer: y += a x on full-lattice fields (2 reads + 1 write), fp64 / fp32, 32^4 and 48^3 x 96, HIP events
(qudaAmdTimeAxpy).  Usage: python3 tools/axpy_timing.py"""
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
qa = importlib.import_module("quda-qkxtm-multigrid_amd")
import numpy as np
from synth import tiled_gauge

qa.init(0)
for X in ((32, 32, 32, 32), (48, 48, 48, 96)):
    qa.load_gauge(tiled_gauge(list(X)), qa.gauge_param(list(X), cuda_prec=8))
    V = int(np.prod(X))
    for prec in (8, 4):
        sx, sy = qa.Spinor(prec, qa.QUDA_FULL_SITE_SUBSET), qa.Spinor(prec, qa.QUDA_FULL_SITE_SUBSET)
        qa.lib().qudaAmdTimeAxpy(0.5, sx.h, sy.h, 5)
        sec = min(qa.lib().qudaAmdTimeAxpy(0.5, sx.h, sy.h, 50) for _ in range(3))
        print("axpy %s prec %d: %.2f us  %.0f GB/s" % ("x".join(map(str, X)), prec, 1e6 * sec, 3 * V * 24 * prec / sec * 1e-9))
        sx.free(); sy.free()
qa.end()
