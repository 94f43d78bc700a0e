"""CPU tests of the multigrid pieces of the oracle (oracle/qo_mg.c).

The reference's own host code for these lives in .cu files that cannot be built here, so there are no golden
vectors; the restatement is pinned the way the reference pins its MG (MG::verify, lib/multigrid.cpp:372-486): block
Gram-Schmidt must give P^dag P = 1, and the coarse operator built from the (bit-pinned) fine links by the restated
calculateY must equal R D P, with D = the oracle's tm_mat / tmc_mat that IS pinned by the reference's golden vectors."""
import numpy as np
import pytest

import oracle_api

X = [4, 4, 4, 8]
BS = [2, 2, 2, 2]
NVEC = 4
KAPPA, MU = 0.12, 0.3


@pytest.fixture(scope="module")
def env():
    o = oracle_api.load()
    gauge, _, clover = o.make_fields(X, seed=11, antiperiodic_t=True, clover=True)
    rng = np.random.default_rng(3)
    V = rng.standard_normal((int(np.prod(X)), 4, 3, NVEC)) + 1j * rng.standard_normal((int(np.prod(X)), 4, 3, NVEC))
    V = o.mg_block_orthogonalize(V, X, BS, 4, 3, NVEC, 2)
    return o, gauge, clover, V, rng


def cvec(rng, shape):
    return rng.standard_normal(shape) + 1j * rng.standard_normal(shape)


def as_real(v):
    return np.ascontiguousarray(v).view(np.float64).reshape(-1)


def test_block_orthonormal(env):
    o, _, _, V, rng = env
    Vc = int(np.prod(X)) // int(np.prod(BS))
    eta = cvec(rng, (Vc, 2, NVEC))
    back = o.mg_restrict(o.mg_prolongate(eta, V, X, BS, 4, 3, NVEC, 2), V, X, BS, 4, 3, NVEC, 2)
    assert np.max(np.abs(back - eta)) < 1e-12
    # R is the adjoint of P
    phi = cvec(rng, (int(np.prod(X)), 4, 3))
    lhs = np.vdot(o.mg_restrict(phi, V, X, BS, 4, 3, NVEC, 2), eta)
    rhs = np.vdot(phi, o.mg_prolongate(eta, V, X, BS, 4, 3, NVEC, 2))
    assert abs(lhs - rhs) < 1e-10 * abs(lhs)


@pytest.mark.parametrize("op", ["wilson", "tm_plus", "tm_minus", "tmc"])
def test_galerkin_fine(env, op):
    o, gauge, clover, V, rng = env
    Xc = [X[d] // BS[d] for d in range(4)]
    flavor = -1 if op == "tm_minus" else 1
    mu = 0.0 if op == "wilson" else MU
    Y, Xm = o.mg_coarse_op_fine(V, gauge, clover if op == "tmc" else None, KAPPA, 2 * KAPPA * mu * flavor, X, BS, NVEC)
    eta = cvec(rng, (int(np.prod(Xc)), 2, NVEC))
    fine = as_real(o.mg_prolongate(eta, V, X, BS, 4, 3, NVEC, 2))
    if op == "wilson":
        Dp = o.wil_mat(gauge, fine, X, KAPPA, 0)
    elif op == "tmc":
        Dp = o.tmc_mat(gauge, clover, fine, X, KAPPA, mu, flavor, 0)
    else:
        Dp = o.tm_mat(gauge, fine, X, KAPPA, mu, flavor, 0)
    want = o.mg_restrict(Dp.view(np.complex128).reshape(-1, 4, 3), V, X, BS, 4, 3, NVEC, 2)
    got = o.mg_coarse_apply(eta, Y, Xm, KAPPA, Xc, NVEC)
    assert np.max(np.abs(got - want)) < 1e-11 * np.max(np.abs(want))


def test_galerkin_coarse(env):
    """second coarsening (from_coarse = true): R2 D_c P2 = D_cc"""
    o, gauge, _, V, rng = env
    Xc = [X[d] // BS[d] for d in range(4)]
    Y, Xm = o.mg_coarse_op_fine(V, gauge, None, KAPPA, 2 * KAPPA * MU, X, BS, NVEC)
    bs2, nv2 = [1, 1, 1, 2], 6
    V2 = o.mg_block_orthogonalize(cvec(rng, (int(np.prod(Xc)), 2, NVEC, nv2)), Xc, bs2, 2, NVEC, nv2, 1)
    Xcc = [Xc[d] // bs2[d] for d in range(4)]
    Y2, X2 = o.mg_coarse_op_coarse(V2, Y, Xm, KAPPA, Xc, bs2, NVEC, nv2)
    eta = cvec(rng, (int(np.prod(Xcc)), 2, nv2))
    mid = o.mg_coarse_apply(o.mg_prolongate(eta, V2, Xc, bs2, 2, NVEC, nv2, 1), Y, Xm, KAPPA, Xc, NVEC)
    want = o.mg_restrict(mid, V2, Xc, bs2, 2, NVEC, nv2, 1)
    got = o.mg_coarse_apply(eta, Y2, X2, KAPPA, Xcc, nv2)
    assert np.max(np.abs(got - want)) < 1e-11 * np.max(np.abs(want))


def test_clover_from_gauge_restatement(env):
    """oracle.clover_compute (restated computeFmunu + computeClover): unit links give the identity; on a random field the
    two chiral blocks equal 1 + i c sum_{mu>nu} sigma_mu_nu (x) F_mu_nu with sigma = (i/2)[gamma_mu, gamma_nu] in the
    DeGrand-Rossi matrices of include/gamma.cuh and F built independently in numpy."""
    o, gauge, _, _, _ = env
    V, Vh, c = int(np.prod(X)), int(np.prod(X)) // 2, 0.3
    unit = np.zeros((4, V, 9), dtype=complex)
    unit[:, :, [0, 4, 8]] = 1
    expect = np.zeros(36)
    expect[:6] = 1
    assert np.max(np.abs(o.clover_compute(np.ascontiguousarray(unit).view(float).reshape(4, -1), c, X).reshape(-1, 36) - expect)) == 0.0
    A = o.clover_compute(gauge, c, X).reshape(-1, 2, 36)

    def expand(blk):
        M = np.diag(blk[:6]).astype(complex)
        L = blk[6:].reshape(15, 2)
        for a in range(6):
            for b in range(a + 1, 6):
                k = 15 - (6 - a) * (5 - a) // 2 + b - a - 1
                M[b, a] = L[k, 0] + 1j * L[k, 1]
                M[a, b] = np.conj(M[b, a])
        return M

    coup = [[3, 2, 1, 0], [3, 2, 1, 0], [2, 3, 0, 1], [2, 3, 0, 1]]
    elem = [[1j, 1j, -1j, -1j], [-1, 1, 1, -1], [1j, -1j, -1j, 1j], [1, 1, 1, 1]]
    g = [np.zeros((4, 4), dtype=complex) for _ in range(4)]
    for d in range(4):
        for s in range(4):
            g[d][s, coup[d][s]] = elem[d][s]

    def link(mu, x):
        x = [x[d] % X[d] for d in range(4)]
        cb = (((x[3] * X[2] + x[2]) * X[1] + x[1]) * X[0] + x[0]) >> 1
        return (gauge[mu].reshape(-1, 9, 2)[(sum(x) & 1) * Vh + cb] @ np.array([1, 1j])).reshape(3, 3)

    def sh(x, mu, s):
        y = list(x)
        y[mu] += s
        return y

    def dag(m):
        return m.conj().T

    for site in ([0, 0, 0, 0], [1, 2, 3, 4], [3, 3, 0, 7]):
        S = np.zeros((12, 12), dtype=complex)
        for mu in range(4):
            for nu in range(mu):
                U = link
                Q = U(mu, site) @ U(nu, sh(site, mu, 1)) @ dag(U(mu, sh(site, nu, 1))) @ dag(U(nu, site))
                Q += U(nu, site) @ dag(U(mu, sh(sh(site, nu, 1), mu, -1))) @ dag(U(nu, sh(site, mu, -1))) @ U(mu, sh(site, mu, -1))
                Q += dag(U(nu, sh(site, nu, -1))) @ U(mu, sh(site, nu, -1)) @ U(nu, sh(sh(site, mu, 1), nu, -1)) @ dag(U(mu, site))
                Q += dag(U(mu, sh(site, mu, -1))) @ dag(U(nu, sh(sh(site, mu, -1), nu, -1))) @ U(mu, sh(sh(site, mu, -1), nu, -1)) @ U(nu, sh(site, nu, -1))
                S += np.kron(0.5j * (g[mu] @ g[nu] - g[nu] @ g[mu]), (Q - dag(Q)) / 8)
        i = (sum(site) & 1) * Vh + ((((site[3] * X[2] + site[2]) * X[1] + site[1]) * X[0] + site[0]) >> 1)
        full = np.zeros((12, 12), dtype=complex)
        full[:6, :6], full[6:, 6:] = expand(A[i, 0]), expand(A[i, 1])
        assert np.max(np.abs(full - (np.eye(12) + 1j * c * S))) < 1e-13
