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
