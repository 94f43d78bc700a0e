#!/usr/bin/env python3
"""First-use verification of the peer-store halo (csrc/dslash.hip): with QUDA_AMD_P2P_VERIFY_FAIL=1 the check is made to fail, the
library must drop back to the staged transport and keep producing the right answer; without it the transport stays 1."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_api  # noqa: E402

qa = importlib.import_module("quda-qkxtm-multigrid_amd")
oracle = oracle_api.load()
X = [4, 4, 6, 8]
gauge, spinor, _ = oracle.make_fields(X)
nh = spinor.size // 2
qa.init(0)
qa.lib().qudaAmdSetPartitionMask(0b1100)
worst = 0.0
for prec, tol in ((8, 1e-12), (4, 2e-5), (2, 1e-2)):
    qa.load_gauge(gauge, qa.gauge_param(X, cuda_prec=prec))
    for parity in (0, 1):
        ip = qa.invert_param(qa.QUDA_TWISTED_MASS_DSLASH, 0.1, 0.3, +1, "ee", 0, cuda_prec=prec)
        for _ in range(3):   # first call verifies, later ones use whatever was decided
            got = qa.dslash(spinor[(1 - parity) * nh:(2 - parity) * nh].copy(), ip, parity)
        want = oracle.tm_dslash(gauge, spinor[(1 - parity) * nh:(2 - parity) * nh].copy(), X, 0.1, 0.3, +1, parity, "ee", 0)
        err = float(np.max(np.abs(got - want)) / np.max(np.abs(want)))
        assert err < tol, (prec, parity, err)
        worst = max(worst, err / tol)
print("transport %d, worst error / tolerance %.2f" % (qa.lib().qudaAmdHaloTransport(), worst))
qa.lib().qudaAmdSetPartitionMask(0)
qa.end()
