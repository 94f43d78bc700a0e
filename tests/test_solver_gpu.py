"""The blocked orthogonalisation of GCR (csrc/blas.hip multi_dot_kernel / multi_caxpy_kernel; reference lib/inv_gcr_quda.cpp:53-84,
:103-121: N dots in one pass, N caxpys in one pass) against the one-direction-at-a-time chain (QUDA_AMD_GCR_BLOCK_ORTHO=0): same
iteration counts within one, every solution's residual recomputed on the HOST with the oracle's tm_mat <= 1e-10."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def _run(block):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, QUDA_AMD_GCR_BLOCK_ORTHO="1" if block else "0")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "gcr_ortho_check.py")], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1]
    return json.loads(line[7:])


def test_blocked_gcr_orthogonalisation_matches_the_sequential_chain():
    blocked, seq = _run(True), _run(False)
    print("blocked:", blocked)
    print("sequential:", seq)
    for name in ("gcr_fp64", "gcr_mixed", "mg_gcr"):
        assert blocked[name]["res"] < 1e-10 and seq[name]["res"] < 1e-10, (name, blocked[name], seq[name])
        assert abs(blocked[name]["iters"] - seq[name]["iters"]) <= 1, (name, blocked[name]["iters"], seq[name]["iters"])
    # the point of the exercise: fewer field passes per iteration in the plain solver (the MG-preconditioned one spends its time in the cycle)
    assert blocked["gcr_mixed"]["secs"] < seq["gcr_mixed"]["secs"], (blocked["gcr_mixed"], seq["gcr_mixed"])
