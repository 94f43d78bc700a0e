#!/usr/bin/env python3
"""One warmed MG-GCR solve, bracketed for the profiler: set-up and warm-up solves run first, then
    marker dispatch -> launch accounting on -> invertQuda -> accounting dumped -> marker dispatch
so that tools/summarize_solve_trace.py can cut exactly one solve out of a `rocprofv3 --kernel-trace` of this program and attach the
ALGORITHMIC bytes of every launch (qa_core.h acct) to its measured duration.

  rocprofv3 --kernel-trace --output-format csv -d <dir> -o t -- python3 tools/mg_solve_profile.py [L_s L_t] [tm|tmc] [acct.json]
default 48 96 (BASELINE configs[4] resident on one GPU)."""
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
dslash = sys.argv[3] if len(sys.argv) > 3 else "tm"
acct = sys.argv[4] if len(sys.argv) > 4 else os.path.join(ROOT, "gpurun_out", "mg_solve_acct.json")
cycle = sys.argv[5] if len(sys.argv) > 5 else "V"   # V: the plain V-cycle of BASELINE.json configs[4] (bench.py), K: the harness' K-cycle
X = (Ls, Ls, Ls, Lt)
if os.environ.get("QA_PROFILE_LATTICE"):   # any lattice, e.g. the 32x16x16x16 sub-lattice of one rank of an 8-GPU split of 32^4 ...
    X = tuple(int(v) for v in os.environ["QA_PROFILE_LATTICE"].split(","))
kappa, mu = 0.124, 0.005
qa.init(0)
if os.environ.get("QA_PROFILE_MASK"):      # ... with its partitioned dimensions emulated by self-neighbour exchanges (bit d: dimension d)
    qa.lib().qudaAmdSetPartitionMask(int(os.environ["QA_PROFILE_MASK"]))
gauge = smooth_gauge_cayley(X, 0.35, workers=min(16, os.cpu_count() or 8)) if min(X) >= 32 else smooth_gauge(X, 0.35)
qa.load_gauge(gauge, qa.gauge_param(X, cuda_prec=8, prec_sloppy=4, prec_precondition=4, t_boundary=qa.QUDA_PERIODIC_T))
ip = qa.invert_param(qa.QUDA_TWISTED_CLOVER_DSLASH if dslash == "tmc" else qa.QUDA_TWISTED_MASS_DSLASH, kappa, mu, +1, "ee", 0, cuda_prec=8, prec_sloppy=4, prec_precondition=4,
                     solution_type=qa.QUDA_MAT_SOLUTION)
ip.solve_type, ip.inv_type, ip.gcrNkrylov, ip.tol, ip.maxiter = qa.QUDA_DIRECT_SOLVE, qa.QUDA_GCR_INVERTER, 20, 1e-10, 5000
if dslash == "tmc":
    ip.clover_coeff = kappa * 1.57551
    qa.load_clover(None, None, ip)
b1 = tuple(2 if ((x // 4) % 2 == 0 and (x // 8) % 2 == 0) else (2 if (x // 4) % 2 == 0 else 1) for x in X)
blocks = [(4, 4, 4, 4), (2, 2, 2, 4) if X == (48, 48, 48, 96) else (2, 2, 2, 2), (2, 2, 2, 2)]
mp = qa.multigrid_param(ip, n_level=3, geo_block=blocks, n_vec=24, setup_maxiter=500, setup_tol=5e-6, smoother_pc=True,
                        cycle=qa.QUDA_MG_CYCLE_VCYCLE if cycle == "V" else qa.QUDA_MG_CYCLE_RECURSIVE)
t0 = time.perf_counter()
mg = qa.Multigrid(mp)
setup = time.perf_counter() - t0
if os.environ.get("QA_PROFILE_HALF"):      # fp16 V + 16-bit level-0 smoother (qudaAmdMultigridSetHalfStorage)
    mg.set_half_storage(True)
if os.environ.get("QA_PROFILE_FUSED") is not None:   # 0: the kernel-per-operation coarse cycle
    qa.lib().qudaAmdMultigridSetFused(int(os.environ["QA_PROFILE_FUSED"]))
ip.inv_type_precondition, ip.preconditioner = qa.QUDA_MG_INVERTER, mg.h
ip.tol_precondition, ip.maxiter_precondition, ip.precondition_cycle, ip.omega = 1e-1, 1, 1, 1.0
b = np.random.default_rng(5).random(int(np.prod(X)) * 24)
for _ in range(2):
    x = qa.invert(b, ip)
qa.lib().qudaAmdDeviceSynchronize()
qa.lib().qudaAmdProfileMarker(1)
qa.lib().qudaAmdAccountStart()
t0 = time.perf_counter()
x = qa.invert(b, ip)
wall = time.perf_counter() - t0
qa.lib().qudaAmdAccountDump(acct.encode())
qa.lib().qudaAmdProfileMarker(2)
qa.lib().qudaAmdDeviceSynchronize()
res = float(np.linalg.norm(b - qa.mat(x, ip)) / np.linalg.norm(b))
print("SOLVE " + json.dumps(dict(lattice="x".join(map(str, X)), action=dslash, cycle=cycle, iters=ip.iter, solver_secs=ip.secs, solve_secs=wall, setup_secs=setup, true_res=res, acct=os.path.basename(acct))), flush=True)
mg.free()
qa.end()
