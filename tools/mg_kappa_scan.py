#!/usr/bin/env python3
"""Where multigrid matters: plain GCR against MG-GCR on the synthetic warm-start field while kappa approaches its critical value
(the twisted mass keeps the operator regular: smallest singular value ~ 2 kappa mu).  One JSON line per kappa; the last line is the
list, which tools/r02 scripts copy to profiles/.

usage: mg_kappa_scan.py LATTICE EPS KAPPA,KAPPA,... MU [maxiter] [out.json]"""
import importlib
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from synth import smooth_gauge

X = tuple(int(v) for v in sys.argv[1].split(","))
eps = float(sys.argv[2])
ks = [float(v) for v in sys.argv[3].split(",")]
mu = float(sys.argv[4])
maxiter = int(sys.argv[5]) if len(sys.argv) > 5 else 20000
outfile = sys.argv[6] if len(sys.argv) > 6 else None
qa = importlib.import_module("quda-qkxtm-multigrid_amd")
qa.init(0)
gauge = smooth_gauge(X, eps)
b = np.random.default_rng(5).random(int(np.prod(X)) * 24)
blocks = [(4, 4, 4, 4), (2, 2, 2, 2), (2, 2, 2, 2)]
rows = []
for kappa in ks:
    gp = qa.gauge_param(X, cuda_prec=8, prec_sloppy=4, prec_precondition=4, t_boundary=qa.QUDA_PERIODIC_T)
    qa.load_gauge(gauge, gp)
    ip = qa.invert_param(qa.QUDA_TWISTED_MASS_DSLASH, kappa, mu, +1, "ee", 0, cuda_prec=8, prec_sloppy=4, prec_precondition=4, solution_type=qa.QUDA_MAT_SOLUTION)
    ip.solve_type, ip.inv_type, ip.gcrNkrylov, ip.tol, ip.maxiter = qa.QUDA_DIRECT_SOLVE, qa.QUDA_GCR_INVERTER, 20, 1e-10, maxiter
    ip.inv_type_precondition = qa.QUDA_INVALID_ENUM
    t0 = time.perf_counter(); x = qa.invert(b, ip); tp = time.perf_counter() - t0
    itp, sp = ip.iter, ip.secs
    resp = float(np.linalg.norm(b - qa.mat(x, ip)) / np.linalg.norm(b))
    mp = qa.multigrid_param(ip, n_level=3, geo_block=blocks, n_vec=24, setup_maxiter=500, setup_tol=5e-6, smoother_pc=True)
    mg = qa.Multigrid(mp)
    ip.inv_type_precondition = qa.QUDA_MG_INVERTER; ip.preconditioner = mg.h
    ip.tol_precondition, ip.maxiter_precondition, ip.precondition_cycle, ip.omega = 1e-1, 1, 1, 1.0
    ip.maxiter = 2000
    qa.invert(b, ip)
    t0 = time.perf_counter(); x = qa.invert(b, ip); tm = time.perf_counter() - t0
    res = float(np.linalg.norm(b - qa.mat(x, ip)) / np.linalg.norm(b))
    row = dict(lattice="x".join(map(str, X)), eps=eps, kappa=kappa, mu=mu, plain_gcr=dict(iters=itp, secs=round(tp, 4), solver_secs=round(sp, 4), true_res=resp),
               mg_gcr=dict(iters=ip.iter, secs=round(tm, 4), solver_secs=round(ip.secs, 4), setup_secs=round(mp.secs, 3), true_res=res),
               speedup_solve=round(tp / tm, 2))
    rows.append(row)
    print(json.dumps(row), flush=True)
    mg.free()
qa.end()
if outfile:
    json.dump(rows, open(outfile, "w"), indent=1)
