"""synth.py — synthetic input generators shared by bench.py and the GPU tests (no reference data needed on the GPU box).

  random_su3 / make_gauge : Haar-random SU(3) links in the host QDP order (hot start, as the reference's dslash tests use:
                            tests/test_util.cpp:879-956 draws random rows and orthonormalises them)
  smooth_gauge            : exp(i eps H) links (warm start): with kappa close to 1/8 the operator is ill-conditioned, which
                            is the regime multigrid is for (the reference MG test uses the unit gauge, multigrid_invert_test.cpp:446)
  make_clover             : uniform(-0.1, 0.1) + 1 on the 12 diagonals (tests/test_util.cpp:1100-1120)
"""
import numpy as np


def random_su3(rng, n):
    g = rng.standard_normal((n, 3, 3)) + 1j * rng.standard_normal((n, 3, 3))
    q, r = np.linalg.qr(g)
    d = np.diagonal(r, axis1=-2, axis2=-1)
    q = q * (d / np.abs(d))[..., None, :]
    q = q / np.linalg.det(q)[..., None, None] ** (1.0 / 3.0)
    return q


def make_gauge(X, seed=137, antiperiodic=True):
    """(4, V*18): even sites then odd, row-major 3x3 complex; anti-periodic T folded into the last time slice"""
    rng = np.random.default_rng(seed)
    V = int(np.prod(X))
    out = np.empty((4, V * 18))
    for mu in range(4):
        q = random_su3(rng, V)
        out[mu] = np.stack([q.real, q.imag], axis=-1).reshape(-1)
    if antiperiodic:
        Vh = V // 2
        lo = (X[0] // 2) * X[1] * X[2] * (X[3] - 1)
        g3 = out[3].reshape(2, Vh, 18)
        g3[:, lo:, :] *= -1.0
    return out


def smooth_gauge(X, eps, seed=3):
    rng = np.random.default_rng(seed)
    V = int(np.prod(X))
    out = np.empty((4, V * 18))
    for mu in range(4):
        a = rng.standard_normal((V, 3, 3)) + 1j * rng.standard_normal((V, 3, 3))
        h = 0.5 * (a + a.conj().transpose(0, 2, 1))
        h -= np.trace(h, axis1=1, axis2=2)[:, None, None] * np.eye(3) / 3.0
        w, v = np.linalg.eigh(h)
        u = (v * np.exp(1j * eps * w)[:, None, :]) @ v.conj().transpose(0, 2, 1)
        out[mu] = np.stack([u.real, u.imag], axis=-1).reshape(-1)
    return out


def make_clover(X, seed=11):
    rng = np.random.default_rng(seed)
    V = int(np.prod(X))
    c = rng.uniform(-0.1, 0.1, size=(V, 72))
    c[:, 0:6] += 1.0
    c[:, 36:42] += 1.0
    return c.reshape(-1)
