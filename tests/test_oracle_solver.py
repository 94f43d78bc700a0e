"""The CPU solver baseline (oracle/qo_solver.c: the reference's restarted GCR, lib/inv_gcr_quda.cpp:235-516, in its plainest
configuration on the host tm_mat with lib/blas_cpu.cpp-style BLAS) is itself checked: its solution satisfies the golden-pinned
operator to the requested tolerance, restarts included."""
import numpy as np


def test_cpu_gcr_solves_the_pinned_operator(oracle):
    X, kappa, mu = [4, 4, 4, 8], 0.12, 0.1
    gauge, spinor, _ = oracle.make_fields(X, seed=21, antiperiodic_t=True, clover=False)
    b = spinor.copy()
    oracle.set_threads(4)
    try:
        for flavor, nk in ((+1, 20), (-1, 6)):
            x, iters, secs, res = oracle.gcr_tm(gauge, b, X, kappa, mu, flavor, tol=1e-10, nkrylov=nk, maxiter=2000)
            assert 0 < iters < 2000 and res < 1e-10 and secs > 0
            r = b - oracle.tm_mat(gauge, x, X, kappa, mu, flavor, 0)
            assert np.linalg.norm(r) / np.linalg.norm(b) < 1e-10
            assert abs(np.linalg.norm(r) / np.linalg.norm(b) - res) < 1e-12   # the reported residual is the true one
    finally:
        oracle.set_threads(1)
