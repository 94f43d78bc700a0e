#!/usr/bin/env python3
"""invertMultiSrcQuda against the same sources through invertQuda one after the other: MG-GCR to 1e-10 on a smooth synthetic field.
  python3 tools/multisrc_timing.py [L_s L_t] [nsrc] [outer: full|pc]        default 32 32 12 pc (the QKXTM drivers' outer even-odd solve)
QA_PROFILE_MARKERS=1 brackets the lockstep solve with marker dispatches and dumps the launch accounting (tools/summarize_solve_trace.py)."""
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
Ls = int(sys.argv[1]) if len(sys.argv) > 1 else 32
Lt = int(sys.argv[2]) if len(sys.argv) > 2 else 32
nsrc = int(sys.argv[3]) if len(sys.argv) > 3 else 12
outer = sys.argv[4] if len(sys.argv) > 4 else "pc"
acct = sys.argv[5] if len(sys.argv) > 5 else "/tmp/multisrc_acct.json"
cycle = os.environ.get("QA_MULTISRC_CYCLE", "V")   # K: the reference harness' default K-cycle (coarse solves source by source, block smoother on the fine level)
X = (Ls, Ls, Ls, Lt)
kappa, mu = 0.124, 0.005
qa.init(0)
gauge = smooth_gauge_cayley(X, 0.35, workers=min(16, os.cpu_count() or 8)) if min(X) >= 32 else smooth_gauge(X, 0.35)
qa.load_gauge(gauge, qa.gauge_param(X, cuda_prec=8, prec_sloppy=4, prec_precondition=4, t_boundary=qa.QUDA_PERIODIC_T))
ip = qa.invert_param(qa.QUDA_TWISTED_MASS_DSLASH, kappa, mu, +1, "ee", 0, cuda_prec=8, prec_sloppy=4, prec_precondition=4, solution_type=qa.QUDA_MAT_SOLUTION)
ip.solve_type, ip.inv_type, ip.gcrNkrylov, ip.tol, ip.maxiter = qa.QUDA_DIRECT_SOLVE, qa.QUDA_GCR_INVERTER, 20, 1e-10, 5000
blocks = [(4, 4, 4, 4), (2, 2, 2, 4) if X == (48, 48, 48, 96) else (2, 2, 2, 2), (2, 2, 2, 2)]
mp = qa.multigrid_param(ip, n_level=3, geo_block=blocks, n_vec=24, setup_maxiter=500, setup_tol=5e-6, smoother_pc=True, cycle=qa.QUDA_MG_CYCLE_VCYCLE if cycle == "V" else qa.QUDA_MG_CYCLE_RECURSIVE)
mg = qa.Multigrid(mp)
ip.inv_type_precondition, ip.preconditioner = qa.QUDA_MG_INVERTER, mg.h
ip.tol_precondition, ip.maxiter_precondition, ip.precondition_cycle, ip.omega = 1e-1, 1, 1, 1.0
if outer == "pc":
    ip.solve_type = qa.QUDA_DIRECT_PC_SOLVE
rng = np.random.default_rng(5)
bs = [rng.random(int(np.prod(X)) * 24) for _ in range(nsrc)]
xbuf = np.zeros_like(bs[0])
qa.invert(bs[0], ip, out=xbuf)
xs = [np.zeros_like(b) for b in bs]
qa.invert_multi_src(bs, ip, out=xs)          # warm both paths (the hierarchy keeps its multi-source work space per source count)
markers = bool(os.environ.get("QA_PROFILE_MARKERS"))
seq = []
for _ in range(1 if markers else 3):          # both legs: the best of three passes over the same sources
    seq_wall, seq_solver, iters = 0.0, 0.0, []
    for b in bs:
        t0 = time.perf_counter()
        qa.invert(b, ip, out=xbuf)
        seq_wall += time.perf_counter() - t0
        seq_solver += ip.secs
        iters.append(ip.iter)
    seq.append((seq_solver, seq_wall))
seq_solver, seq_wall = min(seq)
if markers:
    qa.lib().qudaAmdDeviceSynchronize(); qa.lib().qudaAmdProfileMarker(1); qa.lib().qudaAmdAccountStart()
reps = []
for _ in range(1 if markers else 3):
    t0 = time.perf_counter()
    qa.invert_multi_src(bs, ip, out=xs)
    reps.append((ip.secs, time.perf_counter() - t0))
blk_solver, blk_wall = min(reps)
blk_iter = ip.iter
if markers:
    qa.lib().qudaAmdAccountDump(acct.encode()); qa.lib().qudaAmdProfileMarker(2); qa.lib().qudaAmdDeviceSynchronize()
ip.solve_type = qa.QUDA_DIRECT_SOLVE
worst = max(float(np.linalg.norm(b - qa.mat(x, ip)) / np.linalg.norm(b)) for x, b in zip(xs, bs))
print("SOLVE " + json.dumps(dict(lattice="x".join(map(str, X)), sources=nsrc, outer=outer, cycle=cycle, sequential=dict(solver_secs=round(seq_solver, 4), wall_secs=round(seq_wall, 4), iters=iters, repeats=[round(a, 4) for a, _ in seq]),
                               lockstep=dict(solver_secs=round(blk_solver, 4), wall_secs=round(blk_wall, 4), iters=blk_iter, worst_true_res=worst, repeats=[round(a, 4) for a, _ in reps]), speedup_solver=round(seq_solver / blk_solver, 3),
                               solver_secs=blk_solver, acct=os.path.basename(acct))))
mg.free()
qa.end()
