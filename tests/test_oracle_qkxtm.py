"""CPU checks of oracle/qo_qkxtm.c (Gaussian smearing restated from lib/code_pieces_Kepler/Gauss_core_Kepler.h and
lib/qudaQKXTM_Vector_Kepler.cpp:386-421).  The reference holds no vectors for this kernel (CUDA only), so the restatement
is checked against an independent array formulation and against exact properties."""
import numpy as np
import pytest

import oracle_api


@pytest.fixture(scope="module")
def oracle():
    return oracle_api.load()


def lex_fields(oracle, X, seed=5):
    gauge, spinor, _ = oracle.make_fields(X, seed=seed, antiperiodic_t=False, clover=False)
    V = int(np.prod(X))
    g_lex = np.stack([oracle.eo_to_lex(np.ascontiguousarray(gauge[d]), X, 18) for d in range(4)])
    rng = np.random.default_rng(seed)
    v_lex = rng.standard_normal(V * 24)
    return g_lex, v_lex


def numpy_smear(v_lex, g_lex, X, alpha, n):
    """psi' = (psi + alpha sum_{i<3} [U_i(x) psi(x+i) + U_i(x-i)^dag psi(x-i)]) / (1 + 6 alpha) with array shifts"""
    shape = (X[3], X[2], X[1], X[0])
    psi = v_lex.reshape(-1, 4, 3, 2)
    psi = (psi[..., 0] + 1j * psi[..., 1]).reshape(shape + (4, 3))
    U = g_lex.reshape(4, -1, 3, 3, 2)
    U = (U[..., 0] + 1j * U[..., 1]).reshape((4,) + shape + (3, 3))
    for _ in range(n):
        acc = np.zeros_like(psi)
        for mu in range(3):
            ax = 3 - mu   # axis of direction mu in the (t, z, y, x) array
            fwd = np.roll(psi, -1, axis=ax)
            acc += np.einsum("tzyxab,tzyxsb->tzyxsa", U[mu], fwd)
            back = np.einsum("tzyxba,tzyxsb->tzyxsa", U[mu].conj(), psi)
            acc += np.roll(back, 1, axis=ax)
        psi = (psi + alpha * acc) / (1 + 6 * alpha)
    out = np.stack([psi.real, psi.imag], axis=-1)
    return out.reshape(-1)


@pytest.mark.parametrize("X", [[4, 4, 4, 4], [6, 4, 2, 8]])
def test_smear_matches_array_formulation(oracle, X):
    g_lex, v_lex = lex_fields(oracle, X)
    got = oracle.gauss_smear(v_lex, g_lex, X, 0.7, 3)
    want = numpy_smear(v_lex, g_lex, X, 0.7, 3)
    assert np.max(np.abs(got - want)) < 1e-13 * np.max(np.abs(want))


def test_unit_gauge_point_source_weights(oracle):
    """with unit links one step spreads a point source to its six spatial neighbours with weight alpha / (1 + 6 alpha) and
    never leaves the time slice"""
    X = [4, 4, 4, 4]
    V = int(np.prod(X))
    g = np.zeros((4, V, 3, 3, 2))
    for c in range(3):
        g[:, :, c, c, 0] = 1
    g = g.reshape(4, V * 18)
    v = np.zeros(V * 24)
    x0 = (1, 2, 3, 1)
    iv = ((x0[3] * X[2] + x0[2]) * X[1] + x0[1]) * X[0] + x0[0]
    v[iv * 24 + 2 * 4] = 1.0  # spin 1, colour 1
    alpha = 4.0
    out = oracle.gauss_smear(v, g, X, alpha, 1).reshape(X[3], X[2], X[1], X[0], 24)
    assert abs(out[1, 3, 2, 1, 8] - 1 / 25) < 1e-15
    assert abs(out[1, 3, 2, 2, 8] - 4 / 25) < 1e-15 and abs(out[1, 0, 2, 1, 8] - 4 / 25) < 1e-15
    assert np.count_nonzero(out) == 7
    assert np.count_nonzero(out[0]) == 0 and np.count_nonzero(out[2]) == 0


def test_gauge_covariance(oracle):
    """smear[U^g](g psi) = g smear[U](psi) for a random gauge transformation g(x)"""
    X = [4, 4, 2, 4]
    V = int(np.prod(X))
    g_lex, v_lex = lex_fields(oracle, X, seed=9)
    rng = np.random.default_rng(3)
    A = rng.standard_normal((V, 3, 3)) + 1j * rng.standard_normal((V, 3, 3))
    G, _ = np.linalg.qr(A)
    U = g_lex.reshape(4, V, 3, 3, 2)
    U = U[..., 0] + 1j * U[..., 1]
    shape = (X[3], X[2], X[1], X[0])
    Ug = np.empty_like(U)
    Gs = G.reshape(shape + (3, 3))
    for mu in range(4):
        Gf = np.roll(Gs, -1, axis=3 - mu).reshape(V, 3, 3)
        Ug[mu] = np.einsum("xab,xbc,xdc->xad", G, U[mu], Gf.conj())
    psi = v_lex.reshape(V, 4, 3, 2)
    psi = psi[..., 0] + 1j * psi[..., 1]
    gpsi = np.einsum("xab,xsb->xsa", G, psi)
    pack = lambda a: np.ascontiguousarray(np.stack([a.real, a.imag], axis=-1)).reshape(a.shape[0], -1) if a.ndim == 4 else np.stack([a.real, a.imag], axis=-1).reshape(-1)
    lhs = oracle.gauss_smear(pack(gpsi), np.ascontiguousarray(pack(Ug)), X, 0.5, 4).reshape(V, 4, 3, 2)
    rhs = oracle.gauss_smear(v_lex, g_lex, X, 0.5, 4).reshape(V, 4, 3, 2)
    lhs = lhs[..., 0] + 1j * lhs[..., 1]
    rhs = np.einsum("xab,xsb->xsa", G, rhs[..., 0] + 1j * rhs[..., 1])
    assert np.max(np.abs(lhs - rhs)) < 1e-12


def test_site_order_round_trip(oracle):
    X = [6, 4, 2, 8]
    V = int(np.prod(X))
    eo = np.arange(V * 24, dtype=np.float64)
    lex = oracle.eo_to_lex(eo, X, 24)
    assert np.array_equal(oracle.lex_to_eo(lex, X, 24), eo)
    # checkerboard index i of parity p sits at lexicographic site full_index(i, p) (tests/test_util.cpp:419-443)
    for p, i in [(0, 0), (1, 0), (0, 17), (1, 101)]:
        full = oracle.full_index(X, i, p)
        assert lex[full * 24] == eo[(p * (V // 2) + i) * 24]


def test_basis_rotations_are_inverse(oracle):
    rng = np.random.default_rng(0)
    v = rng.standard_normal((10, 24))
    assert np.allclose(oracle.ukqcd_to_dr(oracle.dr_to_ukqcd(v)), v, atol=1e-15)


def _lex_links(oracle, gauge, X):
    V = int(np.prod(X))
    U = np.stack([oracle.eo_to_lex(np.ascontiguousarray(gauge[d]), X, 18) for d in range(4)]).reshape(4, V, 3, 3, 2)
    return (U[..., 0] + 1j * U[..., 1]).reshape((4, X[3], X[2], X[1], X[0], 3, 3))


def numpy_ape_step(U, alpha):
    """U: (4, t, z, y, x, 3, 3).  Staples by array shifts; the projection in closed form: the unitary polar factor W = u v^dag
    of the SVD (the fixed point of the Newton iteration X <- (X + X^-dag)/2) divided by det(W)^(1/3)"""
    dag = lambda a: a.conj().swapaxes(-1, -2)
    sh = lambda a, mu, s: np.roll(a, -s, axis=3 - mu)   # field at x + s mu
    out = U.copy()
    for nu in range(3):
        S = np.zeros_like(U[0])
        for mu in range(3):
            if mu == nu:
                continue
            S += U[mu] @ sh(U[nu], mu, 1) @ dag(sh(U[mu], nu, 1))
            low = dag(U[mu]) @ U[nu] @ sh(U[mu], nu, 1)
            S += sh(low, mu, -1)
        T = (1 - alpha) * np.eye(3) + (alpha / 4) * S @ dag(U[nu])
        u, _, vh = np.linalg.svd(T)
        W = u @ vh
        det = np.linalg.det(W)
        W = W * (np.exp(-1j * np.angle(det) / 3) / np.abs(det) ** (1 / 3))[..., None, None]
        out[nu] = W @ U[nu]
    return out


@pytest.mark.parametrize("X", [[4, 4, 4, 4], [6, 4, 2, 4]])
def test_ape_matches_array_formulation(oracle, X):
    gauge, _, _ = oracle.make_fields(X, seed=21, antiperiodic_t=True, clover=False)
    got = _lex_links(oracle, oracle.ape_smear(gauge, X, 0.5, 2), X)
    want = _lex_links(oracle, gauge, X)
    for _ in range(2):
        want = numpy_ape_step(want, 0.5)
    assert np.max(np.abs(got - want)) < 1e-12
    assert np.array_equal(got[3], _lex_links(oracle, gauge, X)[3])            # time links untouched
    eye = np.eye(3)
    assert np.max(np.abs(got @ got.conj().swapaxes(-1, -2) - eye)) < 1e-13   # still SU(3)
    assert np.max(np.abs(np.linalg.det(got[:3]) - 1)) < 1e-13                 # (the t links of the last slice carry the boundary sign)


def test_plaquette_values(oracle):
    X = [4, 4, 4, 6]
    V = int(np.prod(X))
    unit = np.zeros((4, V, 9, 2))
    unit[:, :, [0, 4, 8], 0] = 1
    assert np.allclose(oracle.plaquette(unit.reshape(4, -1), X), 1.0, atol=1e-15)
    gauge, _, _ = oracle.make_fields(X, seed=4, antiperiodic_t=True, clover=False)
    U = _lex_links(oracle, gauge, X)
    dag = lambda a: a.conj().swapaxes(-1, -2)
    sh = lambda a, mu: np.roll(a, -1, axis=3 - mu)
    sp = sum(np.trace(U[m] @ sh(U[n], m) @ dag(sh(U[m], n)) @ dag(U[n]), axis1=-2, axis2=-1).real.sum() for m in range(3) for n in range(m + 1, 3))
    tm = sum(np.trace(U[m] @ sh(U[3], m) @ dag(sh(U[m], 3)) @ dag(U[3]), axis1=-2, axis2=-1).real.sum() for m in range(3))
    pl = oracle.plaquette(gauge, X)
    assert abs(pl[1] - sp / (9 * V)) < 1e-13 and abs(pl[2] - tm / (9 * V)) < 1e-13 and abs(pl[0] - 0.5 * (pl[1] + pl[2])) < 1e-15
    # smearing raises the spatial plaquette of a rough field
    assert oracle.plaquette(oracle.ape_smear(gauge, X, 0.5, 3), X)[1] > pl[1] + 0.1
