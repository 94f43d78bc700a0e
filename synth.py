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


def tiled_gauge(X, seed=137, base=1 << 16):
    """links for TIMING runs on big lattices: a periodic repetition of `base` Haar-random SU(3) matrices (seconds instead of
    minutes at 48^3 x 96; kernel time does not depend on the values)"""
    rng = np.random.default_rng(seed)
    V = int(np.prod(X))
    out = np.empty((4, V * 18))
    base = int(np.gcd(V, base))
    for mu in range(4):
        q = random_su3(rng, base)
        out[mu].reshape(V // base, base * 18)[:] = np.stack([q.real, q.imag], axis=-1).reshape(-1)
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


def _cayley_chunk(o, n, eps, seed_seq):
    rng = np.random.default_rng(seed_seq)
    a = rng.standard_normal((3, 3, n)) + 1j * rng.standard_normal((3, 3, n))
    h = 0.5 * (a + a.conj().transpose(1, 0, 2))
    tr = (h[0, 0] + h[1, 1] + h[2, 2]) / 3.0
    for i in range(3):
        h[i, i] -= tr
    m = 0.5j * eps * h
    b = -m
    p = m.copy()
    for i in range(3):
        b[i, i] += 1.0
        p[i, i] += 1.0
    c = np.empty_like(b)   # adjugate of b = transposed cofactors
    for i in range(3):
        for j in range(3):
            i1, i2, j1, j2 = (i + 1) % 3, (i + 2) % 3, (j + 1) % 3, (j + 2) % 3
            c[j, i] = b[i1, j1] * b[i2, j2] - b[i1, j2] * b[i2, j1]
    det = b[0, 0] * c[0, 0] + b[0, 1] * c[1, 0] + b[0, 2] * c[2, 0]
    c /= det
    u = np.empty_like(b)
    for i in range(3):
        for j in range(3):
            u[i, j] = p[i, 0] * c[0, j] + p[i, 1] * c[1, j] + p[i, 2] * c[2, j]
    d = (u[0, 0] * (u[1, 1] * u[2, 2] - u[1, 2] * u[2, 1]) - u[0, 1] * (u[1, 0] * u[2, 2] - u[1, 2] * u[2, 0])
         + u[0, 2] * (u[1, 0] * u[2, 1] - u[1, 1] * u[2, 0]))
    u /= d ** (1.0 / 3.0)
    o[:, :, :, 0] = u.real.transpose(2, 0, 1)
    o[:, :, :, 1] = u.imag.transpose(2, 0, 1)


def smooth_gauge_cayley(X, eps, seed=3, chunk=1 << 18, workers=8):
    """warm-start links like smooth_gauge but through the Cayley map u = (1 + i eps H / 2)(1 - i eps H / 2)^-1, det-normalised,
    with closed-form 3x3 algebra on component-major arrays (no batched LAPACK) in a thread pool, one seeded generator per
    chunk so the field does not depend on the scheduling — for the 48^3 x 96 single-GPU run (tools/c5_single_gpu.py)"""
    from concurrent.futures import ThreadPoolExecutor
    V = int(np.prod(X))
    out = np.empty((4, V * 18))
    tasks = [(mu, lo) for mu in range(4) for lo in range(0, V, chunk)]
    seeds = np.random.SeedSequence(seed).spawn(len(tasks))
    with ThreadPoolExecutor(workers) as pool:
        futs = []
        for (mu, lo), ss in zip(tasks, seeds):
            n = min(chunk, V - lo)
            futs.append(pool.submit(_cayley_chunk, out[mu].reshape(V, 3, 3, 2)[lo:lo + n], n, eps, ss))
        for f in futs:
            f.result()
    return out


def make_clover(X, seed=11):
    rng = np.random.default_rng(seed)
    V = int(np.prod(X))
    c = rng.uniform(-0.1, 0.1, size=(V, 72))
    c[:, 0:6] += 1.0
    c[:, 36:42] += 1.0
    return c.reshape(-1)
