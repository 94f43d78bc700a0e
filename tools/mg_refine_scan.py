#!/usr/bin/env python3
"""MG-GCR at the critical kappa of the synthetic 32^4 field (0.147, mu 0.001): outer iterations and solver seconds after the plain set-up
and after set-up refinement passes (multigrid_solver::refine: inverse iteration of the null vectors through the hierarchy)."""
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from synth import smooth_gauge  # noqa: E402

qa = importlib.import_module("quda-qkxtm-multigrid_amd")
L = int(sys.argv[1]) if len(sys.argv) > 1 else 32
kappa = float(sys.argv[2]) if len(sys.argv) > 2 else 0.147
mu = float(sys.argv[3]) if len(sys.argv) > 3 else 0.001
X = (L, L, L, L)
qa.init(0)
gauge = smooth_gauge(X, 0.35)
qa.load_gauge(gauge, qa.gauge_param(X, cuda_prec=8, prec_sloppy=4, prec_precondition=4, t_boundary=qa.QUDA_PERIODIC_T))
ip = qa.invert_param(qa.QUDA_TWISTED_MASS_DSLASH, kappa, mu, +1, "ee", 0, cuda_prec=8, prec_sloppy=4, prec_precondition=4, solution_type=qa.QUDA_MAT_SOLUTION)
ip.solve_type, ip.inv_type, ip.gcrNkrylov, ip.tol, ip.maxiter = qa.QUDA_DIRECT_SOLVE, qa.QUDA_GCR_INVERTER, 20, 5e-11, 2000
b = np.random.default_rng(5).random(int(np.prod(X)) * 24)
mp = qa.multigrid_param(ip, n_level=3, geo_block=[(4, 4, 4, 4), (2, 2, 2, 2), (2, 2, 2, 2)], n_vec=24, setup_maxiter=500, setup_tol=5e-6, smoother_pc=True)
mg = qa.Multigrid(mp)
ip.inv_type_precondition, ip.preconditioner = qa.QUDA_MG_INVERTER, mg.h
ip.tol_precondition, ip.maxiter_precondition, ip.precondition_cycle, ip.omega = 1e-1, 1, 1, 1.0


def solve(tag, extra):
    qa.invert(b, ip)
    x = qa.invert(b, ip)
    res = float(np.linalg.norm(b - qa.mat(x, ip)) / np.linalg.norm(b))
    print("REFINE " + json.dumps(dict(stage=tag, iters=ip.iter, solver_secs=round(ip.secs, 4), true_res=res, **extra)), flush=True)


solve("plain set-up", dict(setup_secs=round(mp.secs, 3)))
for p in range(1, 4):
    secs = mg.refine(1, 1)
    solve("after refinement pass %d" % p, dict(refine_secs=round(secs, 3)))
mg.free()
qa.end()
