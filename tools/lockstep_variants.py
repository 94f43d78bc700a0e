#!/usr/bin/env python3
"""One set-up with the lockstep BiCGstab in the variant the environment selects (QUDA_AMD_BLOCK_BICG_FUSED, QUDA_AMD_BLOCK_FINE_DOTS), twisted mass and
twisted clover: lockstep iteration count, a fingerprint of the null vectors, |M v| / |v| through the library's own fine operator, and the MG-GCR
iteration count of one solve.  One JSON line; tests/test_mg_gpu.py compares the variants."""
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from synth import smooth_gauge  # noqa: E402

qa = importlib.import_module("quda-qkxtm-multigrid_amd")
qa.init(0)
X, kappa, mu = (8, 8, 8, 16), 0.124, 0.005
gauge = smooth_gauge(X, 0.35)
out = {}
for action in ("tm", "tmc"):
    qa.lib().freeCloverQuda()
    qa.load_gauge(gauge, qa.gauge_param(X, cuda_prec=8, prec_sloppy=4, prec_precondition=4, t_boundary=qa.QUDA_PERIODIC_T))
    ip = qa.invert_param(qa.QUDA_TWISTED_CLOVER_DSLASH if action == "tmc" else qa.QUDA_TWISTED_MASS_DSLASH, kappa, mu, +1, "ee", 0, cuda_prec=8, prec_sloppy=4,
                         prec_precondition=4, solution_type=qa.QUDA_MAT_SOLUTION)
    ip.solve_type, ip.inv_type, ip.gcrNkrylov, ip.tol, ip.maxiter, ip.reliable_delta, ip.verbosity = qa.QUDA_DIRECT_SOLVE, qa.QUDA_GCR_INVERTER, 20, 1e-10, 2000, 1e-4, qa.QUDA_SILENT
    if action == "tmc":
        ip.clover_coeff = kappa * 1.57551
        qa.load_clover(None, None, ip)
    mp = qa.multigrid_param(ip, n_level=2, geo_block=(4, 4, 4, 4), n_vec=8, setup_maxiter=300, setup_tol=1e-5)
    mg = qa.Multigrid(mp)
    info = mg.level_info(0)
    vecs = [mg.null_vector(0, k).astype(np.complex128) for k in range(8)]
    quality = []
    for v in vecs:
        mv = mg.apply(0, "M", v)
        quality.append(float(np.linalg.norm(mv) / np.linalg.norm(v)))
    ip.inv_type_precondition, ip.preconditioner = qa.QUDA_MG_INVERTER, mg.h
    ip.tol_precondition, ip.maxiter_precondition, ip.precondition_cycle, ip.omega = 1e-1, 1, 1, 1.0
    b = np.random.default_rng(3).random(int(np.prod(X)) * 24)
    x = qa.invert(b, ip)
    res = float(np.linalg.norm(b - qa.mat(x, ip)) / np.linalg.norm(b))
    out[action] = dict(null_method=info["null_method"], null_iters=info["null_iters"], quality=quality, iters=int(ip.iter), true_res=res,
                       fingerprint=[[float(v.reshape(-1)[j].real), float(v.reshape(-1)[j].imag)] for v in vecs[:3] for j in (0, 1234, 5000)])
    mg.free()
print("LOCKSTEP " + json.dumps(out), flush=True)
qa.end()
