#!/usr/bin/env python3
"""Seconds per application of the 8-right-hand-side fine operator (csrc/dslash.hip fine_block_kernel<8>: two parity launches = M on
8 vectors), its HBM fraction on the algorithmic bytes (576/8 + 96 + 96 + 96 = 360 B per site and right-hand side), twisted mass and
twisted clover.  QUDA_AMD_BLOCK_FINE_TILE=0 selects the linear site mapping for comparison."""
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
X = tuple(int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "48,48,48,48").split(","))
qa.init(0)
gauge = tiled_gauge(list(X))
V = int(np.prod(X))
rng = np.random.default_rng(1)
out = {}
for dslash in ("tm", "tmc"):
    qa.load_gauge(gauge, qa.gauge_param(X, cuda_prec=8, prec_sloppy=4, prec_precondition=4, t_boundary=qa.QUDA_PERIODIC_T))
    ip = qa.invert_param(qa.QUDA_TWISTED_CLOVER_DSLASH if dslash == "tmc" else qa.QUDA_TWISTED_MASS_DSLASH, 0.124, 0.005, +1, "ee", 0, cuda_prec=8, prec_sloppy=4, prec_precondition=4,
                         solution_type=qa.QUDA_MAT_SOLUTION)
    ip.solve_type, ip.inv_type = qa.QUDA_DIRECT_SOLVE, qa.QUDA_GCR_INVERTER
    if dslash == "tmc":
        ip.clover_coeff = 0.124 * 1.57551
        qa.load_clover(None, None, ip)
    mp = qa.multigrid_param(ip, n_level=2, geo_block=(4, 4, 4, 4), n_vec=8, setup_maxiter=4, setup_tol=1e-1)
    mg = qa.Multigrid(mp)
    phi = (rng.standard_normal((8, V, 4, 3)) + 1j * rng.standard_normal((8, V, 4, 3))).astype(np.complex64)
    _, secs = mg.apply_block(0, phi, niter=20)
    algo = 360.0 + (72.0 if dslash == "tmc" else 0.0)
    out[dslash] = dict(us_per_M_8rhs=round(1e6 * secs, 1), us_per_parity_launch=round(0.5e6 * secs, 1), hbm_frac=round(V * 8 * algo / secs / 8e12, 4))
    mg.free()
    del phi
print("FINEBLOCK " + json.dumps(dict(lattice="x".join(map(str, X)), tile=os.environ.get("QUDA_AMD_BLOCK_FINE_TILE", "1"), **out)), flush=True)
qa.end()
