#!/usr/bin/env python3
"""Plain GCR(20) and MG-GCR on a 16^4 twisted-mass problem; prints one JSON line with iteration counts, solver seconds and the residuals
recomputed on the HOST with the oracle's tm_mat.  QUDA_AMD_GCR_BLOCK_ORTHO=0 in the environment selects the one-direction-at-a-time
orthogonalisation (the reference's chain) instead of the blocked one — tests/test_solver_gpu.py runs both and compares."""
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_api  # noqa: E402
from synth import smooth_gauge  # noqa: E402

qa = importlib.import_module("quda-qkxtm-multigrid_amd")
oracle = oracle_api.load()
qa.init(0)
X, kappa, mu = (16, 16, 16, 16), 0.124, 0.005
gauge = smooth_gauge(X, 0.35)
qa.load_gauge(gauge, qa.gauge_param(X, cuda_prec=8, prec_sloppy=4, prec_precondition=4, t_boundary=qa.QUDA_PERIODIC_T))
b = np.random.default_rng(5).random(int(np.prod(X)) * 24)
out = {}


def host_res(x):
    oracle.set_threads(8)
    r = float(np.linalg.norm(b - oracle.tm_mat(gauge, x, list(X), kappa, mu, +1, 0)) / np.linalg.norm(b))
    oracle.set_threads(1)
    return r


for name, sloppy in (("gcr_fp64", 8), ("gcr_mixed", 4)):
    ip = qa.invert_param(qa.QUDA_TWISTED_MASS_DSLASH, kappa, mu, +1, "ee", 0, cuda_prec=8, prec_sloppy=sloppy, prec_precondition=sloppy, solution_type=qa.QUDA_MAT_SOLUTION)
    ip.solve_type, ip.inv_type, ip.gcrNkrylov, ip.tol, ip.maxiter = qa.QUDA_DIRECT_SOLVE, qa.QUDA_GCR_INVERTER, 20, 1e-10, 5000
    if sloppy == 8:
        qa.load_gauge(gauge, qa.gauge_param(X, cuda_prec=8, t_boundary=qa.QUDA_PERIODIC_T))
    else:
        qa.load_gauge(gauge, qa.gauge_param(X, cuda_prec=8, prec_sloppy=4, prec_precondition=4, t_boundary=qa.QUDA_PERIODIC_T))
    qa.invert(b, ip)
    secs = []
    for _ in range(3):      # best of three: single calls on a shared test box occasionally stall on the host side
        x = qa.invert(b, ip)
        secs.append(ip.secs)
    out[name] = dict(iters=ip.iter, secs=min(secs), res=host_res(x))
mp = qa.multigrid_param(ip, n_level=3, geo_block=[(4, 4, 4, 4), (2, 2, 2, 2), (2, 2, 2, 2)], n_vec=24, setup_maxiter=500, setup_tol=5e-6, smoother_pc=True)
mg = qa.Multigrid(mp)
ip.inv_type_precondition, ip.preconditioner = qa.QUDA_MG_INVERTER, mg.h
ip.tol_precondition, ip.maxiter_precondition, ip.precondition_cycle, ip.omega = 1e-1, 1, 1, 1.0
qa.invert(b, ip)
secs = []
for _ in range(3):
    x = qa.invert(b, ip)
    secs.append(ip.secs)
out["mg_gcr"] = dict(iters=ip.iter, secs=min(secs), res=host_res(x))
mg.free()
qa.end()
print("RESULT " + json.dumps(out), flush=True)
