"""The drop-in boundary on the GPU: programs that are NOT the Python harness drive libquda.so.

  * tests/consumer/c_driver.c (plain C, the call sequence of INTEGRATION.md section 1) — its MG-GCR solution is re-checked here
    with the oracle's tm_mat;
  * tests/consumer/cxx_consumer.cpp (the C++ surface under the reference's header names);
  * oracle/_ref/mg_invert_test — the reference's own tests/multigrid_invert_test.cpp, unmodified, built in the build container
    against include/ + libquda.so (oracle/Makefile `dropin`); it verifies its solution with the reference's own host tm_mat.
"""
import os
import re
import subprocess

import numpy as np
import pytest

from test_dropin_build import ROOT, build_consumers

pytestmark = pytest.mark.gpu


def test_plain_c_driver_runs_the_integration_sequence(oracle, tmp_path):
    cdrv, _ = build_consumers(str(tmp_path))
    out = tmp_path / "c_driver.bin"
    r = subprocess.run([cdrv, "8", "8", str(out)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    m = re.search(r"recomputed ([0-9.eE+-]+)", r.stdout)
    assert m and float(m.group(1)) < 1e-9, r.stdout
    X, V = [8, 8, 8, 8], 8 ** 4
    raw = np.fromfile(str(out))
    gauge = raw[:4 * V * 18].reshape(4, V * 18)
    b, x = raw[4 * V * 18:4 * V * 18 + V * 24], raw[4 * V * 18 + V * 24:]
    # the links of the driver are unitary to ~1e-3 only (second-order exponential, Gram-Schmidt): the oracle takes them as they are
    want = oracle.tm_mat(gauge, x.copy(), X, 0.12, 0.02, +1, 0)
    res = np.linalg.norm(b - want) / np.linalg.norm(b)
    assert res < 1e-9, res


def test_cxx_consumer_uses_the_cxx_surface(tmp_path):
    _, cxx = build_consumers(str(tmp_path))
    r = subprocess.run([cxx, "8"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    assert "through the C++ surface" in r.stdout


def test_reference_mg_test_program_runs_against_this_library():
    # --mass -0.2 (kappa 0.132) on the program's random gauge field: 12 solves of 18 iterations, ~20 s.  With --mass -0.9 the same
    # program needs 161 iterations per solve and five minutes without a line of output — beyond what a test runner waits for.
    exe = os.path.join(ROOT, "oracle", "_ref", "mg_invert_test")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/mg_invert_test is built only where the reference tree exists")
    tol = 1e-8
    r = subprocess.run([exe, "--dim", "16", "16", "16", "16", "--dslash-type", "twisted-mass", "--flavor", "plus", "--mass", "-0.2", "--mu", "0.1", "--tol", str(tol),
                        "--prec", "double", "--prec-sloppy", "single", "--prec-precondition", "single", "--recon", "18", "--recon-sloppy", "18",
                        "--recon-precondition", "18", "--mg-levels", "2", "--mg-nvec", "0", "24", "--niter", "200"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    m = re.search(r"Residuals: \(L2 relative\) tol ([0-9.eE+-]+), QUDA = ([0-9.eE+-]+), host = ([0-9.eE+-]+)", r.stdout)
    assert m, r.stdout[-3000:]
    assert float(m.group(3)) < 10 * tol, r.stdout[-1500:]
