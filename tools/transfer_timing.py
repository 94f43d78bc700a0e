#!/usr/bin/env python3
"""Seconds per restriction / prolongation of level 0 of a 3-level hierarchy (24 null vectors, 4^4 aggregates), with the HBM rate
on V + fine + coarse vector.  usage: transfer_timing.py [lattice]   (QUDA_AMD_PROLONG_VAR selects a prolongator variant)"""
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from synth import smooth_gauge_cayley

X = tuple(int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "32,32,32,32").split(","))
qa = importlib.import_module("quda-qkxtm-multigrid_amd")
qa.init(0)
gauge = smooth_gauge_cayley(X, 0.35, workers=8)
gp = qa.gauge_param(X, cuda_prec=8, prec_sloppy=4, prec_precondition=4, t_boundary=qa.QUDA_PERIODIC_T)
qa.load_gauge(gauge, gp)
ip = qa.invert_param(qa.QUDA_TWISTED_MASS_DSLASH, 0.124, 0.005, +1, "ee", 0, cuda_prec=8, prec_sloppy=4, prec_precondition=4, solution_type=qa.QUDA_MAT_SOLUTION)
ip.solve_type, ip.inv_type, ip.gcrNkrylov, ip.tol, ip.maxiter = qa.QUDA_DIRECT_SOLVE, qa.QUDA_GCR_INVERTER, 20, 1e-10, 2000
lvl1 = [x // 4 for x in X]
b1 = tuple(4 if (x % 4 == 0 and (x // 4) % 2 == 0) else 2 for x in lvl1)
mp = qa.multigrid_param(ip, n_level=3, geo_block=[(4, 4, 4, 4), b1, (2, 2, 2, 2)], n_vec=24, setup_maxiter=500, setup_tol=5e-6, smoother_pc=True)
mg = qa.Multigrid(mp)
V = int(np.prod(X))
nbytes = V * 12 * 24 * 8 + V * 24 * 4 + V // 256 * 48 * 8
for what in ("R", "P"):
    s = min(mg.time_transfer(0, what, 20) for _ in range(3))
    print("%s lattice %s var %s: %.1f us  %.0f GB/s  %.3f of 8 TB/s" % (what, "x".join(map(str, X)), os.environ.get("QUDA_AMD_PROLONG_VAR", "0"), 1e6 * s, nbytes / s * 1e-9, nbytes / s / 8e12))
print("setup %.3f s" % mp.secs)
mg.free()
qa.end()
