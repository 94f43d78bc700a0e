#!/usr/bin/env python3
"""Microseconds per application of the MFMA multi-right-hand-side coarse operator (block.hip coarse_block_kernel<48, nrhs>) on the first coarse
lattice of a 3-level hierarchy: 12^3 x 24 from 48^3 x 96 by default.  usage: coarse_block_timing.py [L_s L_t] [nrhs ...]"""
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from synth import tiled_gauge  # noqa: E402

qa = importlib.import_module("quda-qkxtm-multigrid_amd")
Ls = int(sys.argv[1]) if len(sys.argv) > 1 else 48
Lt = int(sys.argv[2]) if len(sys.argv) > 2 else 96
nrhs_list = [int(v) for v in sys.argv[3:]] or [24]
X = (Ls, Ls, Ls, Lt)
qa.init(0)
qa.load_gauge(tiled_gauge(list(X)), qa.gauge_param(X, cuda_prec=8, prec_sloppy=4, prec_precondition=4, t_boundary=qa.QUDA_PERIODIC_T))
ip = qa.invert_param(qa.QUDA_TWISTED_MASS_DSLASH, 0.124, 0.005, +1, "ee", 0, cuda_prec=8, prec_sloppy=4, prec_precondition=4, solution_type=qa.QUDA_MAT_SOLUTION)
ip.solve_type, ip.inv_type = qa.QUDA_DIRECT_SOLVE, qa.QUDA_GCR_INVERTER
mp = qa.multigrid_param(ip, n_level=2, geo_block=(4, 4, 4, 4), n_vec=24, setup_maxiter=2, setup_tol=1e-1)
mg = qa.Multigrid(mp)
Xc = mg.level_info(0)["Xc"]
nc = int(np.prod(Xc))
rng = np.random.default_rng(2)
out = dict(coarse_lattice="x".join(map(str, Xc)))
for nrhs in nrhs_list:
    eta = (rng.standard_normal((nrhs, nc, 2, 24)) + 1j * rng.standard_normal((nrhs, nc, 2, 24))).astype(np.complex64)
    mg.apply_block(1, eta, niter=5)
    secs = min(mg.apply_block(1, eta, niter=30)[1] for _ in range(3))
    flops = 8.0 * 9 * 48 * 48 * nc * nrhs
    out["nrhs_%d" % nrhs] = dict(us=round(1e6 * secs, 1), mfma_tflops=round(flops / secs * 1e-12, 1), mfma_frac_of_157=round(flops / secs / 157e12, 3),
                                 hbm_frac=round((9 * 48 * 48 * 8 + 2 * 48 * 8 * nrhs) * nc / secs / 8e12, 3))
print("COARSEBLOCK " + json.dumps(out), flush=True)
mg.free()
qa.end()
