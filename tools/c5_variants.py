#!/usr/bin/env python3
"""48^3 x 96 MG-GCR on one GPU in several cycle configurations (K-cycle / V-cycle on level 1, fp32 / fp16 storage of V and the coarse
links + 16-bit level-0 smoother, smoother sweeps): iterations, solver seconds (best of 3), residual through MatQuda.  One JSON line each."""
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from synth import smooth_gauge, smooth_gauge_cayley  # noqa: E402

qa = importlib.import_module("quda-qkxtm-multigrid_amd")
Ls = int(sys.argv[1]) if len(sys.argv) > 1 else 48
Lt = int(sys.argv[2]) if len(sys.argv) > 2 else 96
X = (Ls, Ls, Ls, Lt)
kappa, mu = 0.124, 0.005
qa.init(0)
gauge = smooth_gauge_cayley(X, 0.35, workers=min(16, os.cpu_count() or 8)) if Ls >= 32 else smooth_gauge(X, 0.35)
qa.load_gauge(gauge, qa.gauge_param(X, cuda_prec=8, prec_sloppy=4, prec_precondition=4, t_boundary=qa.QUDA_PERIODIC_T))
b = np.random.default_rng(5).random(int(np.prod(X)) * 24)
blocks = [(4, 4, 4, 4), (2, 2, 2, 4) if X == (48, 48, 48, 96) else (2, 2, 2, 2), (2, 2, 2, 2)]
variants = sys.argv[3:] or ["K", "V", "K:nu1", "K:half", "V:half"]
for var in variants:
    opts = var.split(":")
    ip = qa.invert_param(qa.QUDA_TWISTED_MASS_DSLASH, kappa, mu, +1, "ee", 0, cuda_prec=8, prec_sloppy=4, prec_precondition=4, solution_type=qa.QUDA_MAT_SOLUTION)
    ip.solve_type, ip.inv_type, ip.gcrNkrylov, ip.tol, ip.maxiter = qa.QUDA_DIRECT_SOLVE, qa.QUDA_GCR_INVERTER, 20, 1e-10, 5000
    nu = 1 if "nu1" in opts else (3 if "nu3" in opts else 2)
    mp = qa.multigrid_param(ip, n_level=3, geo_block=blocks, n_vec=24, setup_maxiter=500, setup_tol=5e-6, smoother_pc=True, nu_pre=nu, nu_post=nu,
                            cycle=qa.QUDA_MG_CYCLE_VCYCLE if opts[0] == "V" else qa.QUDA_MG_CYCLE_RECURSIVE)
    mg = qa.Multigrid(mp)
    if "half" in opts:
        mg.set_half_storage(True)
    ip.inv_type_precondition, ip.preconditioner = qa.QUDA_MG_INVERTER, mg.h
    ip.tol_precondition, ip.maxiter_precondition, ip.precondition_cycle, ip.omega = 1e-1, 1, 1, 1.0
    qa.invert(b, ip)
    best = None
    for _ in range(3):
        x = qa.invert(b, ip)
        if best is None or ip.secs < best[0]:
            best = (ip.secs, ip.iter)
    res = float(np.linalg.norm(b - qa.mat(x, ip)) / np.linalg.norm(b))
    dev = mg.verify()
    print("VARIANT " + json.dumps(dict(variant=var, setup_secs=round(mp.secs, 3), iters=best[1], solver_secs=round(best[0], 4), true_res=res, verify=[float("%.2e" % v) for v in dev])), flush=True)
    mg.set_half_storage(False)
    mg.free()
qa.end()
