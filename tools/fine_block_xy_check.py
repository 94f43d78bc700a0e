#!/usr/bin/env python3
"""The 8-right-hand-side stencil (with QUDA_AMD_BLOCK_FINE_XYTILE=1: its x / y tile variant) against the host tm_mat on a lattice the 8 x 4 tiles cover,
unpartitioned and with self-neighbour partition masks (x / y faces through the ghost zone)."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_api  # noqa: E402  (test infrastructure: the checker only)
from synth import smooth_gauge  # noqa: E402

qa = importlib.import_module("quda-qkxtm-multigrid_amd")
oracle = oracle_api.load()
qa.init(0)
kappa, mu = 0.124, 0.005
worst = 0.0
for X in ((16, 8, 8, 8), (32, 4, 4, 8)):
    V = int(np.prod(X))
    gauge = smooth_gauge(X, 0.35)
    for mask in (0, 3, 15, 12):
        qa.lib().qudaAmdSetPartitionMask(mask)
        qa.load_gauge(gauge, qa.gauge_param(X, cuda_prec=8, prec_sloppy=4, prec_precondition=4, t_boundary=qa.QUDA_PERIODIC_T))
        ip = qa.invert_param(qa.QUDA_TWISTED_MASS_DSLASH, kappa, mu, +1, "ee", 0, cuda_prec=8, prec_sloppy=4, prec_precondition=4, solution_type=qa.QUDA_MAT_SOLUTION)
        ip.solve_type, ip.inv_type, ip.verbosity = qa.QUDA_DIRECT_SOLVE, qa.QUDA_GCR_INVERTER, qa.QUDA_SILENT
        mp = qa.multigrid_param(ip, n_level=2, geo_block=(4, 4, 4, 4) if X[1] % 4 == 0 and X[2] % 4 == 0 else (4, 4, 4, 4), n_vec=8, setup_maxiter=20, setup_tol=1e-2)
        mg = qa.Multigrid(mp)
        rng = np.random.default_rng(7)
        phi = (rng.standard_normal((8, V, 4, 3)) + 1j * rng.standard_normal((8, V, 4, 3))).astype(np.complex64)
        got, _ = mg.apply_block(0, phi)
        oracle.set_threads(8)
        for k in range(8):
            v = np.ascontiguousarray(phi[k].astype(np.complex128)).view(np.float64).reshape(-1)
            want = oracle.tm_mat(gauge, v, list(X), kappa, mu, +1, 0).view(np.complex128).reshape(-1, 4, 3)
            rel = float(np.max(np.abs(got[k] - want)) / np.max(np.abs(want)))
            worst = max(worst, rel)
        oracle.set_threads(1)
        print("XYCHECK lattice %s mask %d: worst relative deviation so far %.2e, null method %d" % ("x".join(map(str, X)), mask, worst, mg.level_info(0)["null_method"]), flush=True)
        mg.free()
qa.lib().qudaAmdSetPartitionMask(0)
assert worst < 2e-5, worst
print("XYCHECK ok", flush=True)
qa.end()
