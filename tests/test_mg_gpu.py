"""Multigrid on the GPU through the C ABI (newMultigridQuda / invertQuda), pinned the way the reference pins it
(SURVEY 8c): (1) the three MG::verify() identities, which are known-answer (= 0) tests of R, P, block-orthonormality and
the Galerkin coarse operator against the already-oracled fine operator; (2) the MG-GCR solution, whose true residual
|b - M x| / |b| is recomputed on the host with the oracle's tm_mat (as tests/multigrid_invert_test.cpp:529-577)."""
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def qa():
    mod = importlib.import_module("quda-qkxtm-multigrid_amd")
    mod.init(0)
    yield mod
    mod.end()


from synth import smooth_gauge  # noqa: E402


def _setup(qa, X, kappa, mu, eps=0.35):
    gauge = smooth_gauge(X, eps)
    gp = qa.gauge_param(X, cuda_prec=8, prec_sloppy=4, prec_precondition=4, t_boundary=qa.QUDA_PERIODIC_T)
    qa.load_gauge(gauge, gp)
    ip = qa.invert_param(qa.QUDA_TWISTED_MASS_DSLASH, kappa, mu, +1, "ee", 0, cuda_prec=8, prec_sloppy=4, prec_precondition=4,
                         solution_type=qa.QUDA_MAT_SOLUTION)
    ip.solve_type = qa.QUDA_DIRECT_SOLVE
    ip.inv_type = qa.QUDA_GCR_INVERTER
    ip.gcrNkrylov = 20
    ip.tol = 1e-10
    ip.maxiter = 2000
    ip.reliable_delta = 1e-4
    ip.verbosity = qa.QUDA_SILENT
    return gauge, ip


def _true_residual(oracle, gauge, X, kappa, mu, x, b):
    oracle.set_threads(8)
    try:
        mx = oracle.tm_mat(gauge, x, list(X), kappa, mu, +1, 0)
    finally:
        oracle.set_threads(1)
    return float(np.linalg.norm(b - mx) / np.linalg.norm(b))


@pytest.mark.parametrize("smoother_pc", [False, True], ids=["full-smoother", "pc-smoother"])
@pytest.mark.parametrize("X,levels,blocks,nvec", [((8, 8, 8, 8), 2, (4, 4, 4, 4), 8), ((16, 8, 8, 16), 3, [(4, 4, 4, 4), (2, 2, 2, 2), (2, 2, 2, 2)], 8)])
def test_verify_identities_and_mg_gcr_solve(qa, oracle, X, levels, blocks, nvec, smoother_pc):
    kappa, mu = 0.124, 0.005
    gauge, ip = _setup(qa, X, kappa, mu)
    V = int(np.prod(X))
    rng = np.random.default_rng(5)
    b = rng.random(V * 24)

    # plain GCR (no preconditioner) as the baseline
    ip.inv_type_precondition = qa.QUDA_INVALID_ENUM
    x0 = qa.invert(b, ip)
    iters_plain = ip.iter
    assert _true_residual(oracle, gauge, X, kappa, mu, x0, b) < 1e-10

    mp = qa.multigrid_param(ip, n_level=levels, geo_block=blocks, n_vec=nvec, setup_maxiter=300, setup_tol=1e-5, smoother_pc=smoother_pc)
    mg = qa.Multigrid(mp)
    try:
        dev = mg.verify()
        # reference threshold 10^(4 - 2 prec) = 1e-4 for fp32 (lib/multigrid.cpp:381)
        assert dev[0] < 1e-4 and dev[1] < 1e-4 and dev[2] < 1e-4, dev
        ip.inv_type_precondition = qa.QUDA_MG_INVERTER
        ip.preconditioner = mg.h
        ip.tol_precondition = 1e-1
        ip.maxiter_precondition = 1
        ip.precondition_cycle = 1
        ip.omega = 1.0
        x = qa.invert(b, ip)
        iters_mg = ip.iter
        res = _true_residual(oracle, gauge, X, kappa, mu, x, b)
        assert res < 1e-10, res
        assert abs(ip.true_res - res) < 1e-9
        assert iters_mg < iters_plain, (iters_mg, iters_plain)
        print("MG-GCR %s pc=%s: %d iterations (plain GCR %d), true residual %.2e, setup %.2f s, solve %.3f s" % (X, smoother_pc, iters_mg, iters_plain, res, mp.secs, ip.secs))
    finally:
        mg.free()


def test_restrictor_prolongator_are_adjoint(qa):
    """<P c, f> = <c, R f> on the device hierarchy, via one V-cycle object: exercised through qudaAmdMultigridVerify identity (2)
    plus an explicit cycle call that must reduce the residual."""
    X = (8, 8, 8, 8)
    kappa, mu = 0.124, 0.005
    gauge, ip = _setup(qa, X, kappa, mu)
    mp = qa.multigrid_param(ip, n_level=2, geo_block=(4, 4, 4, 4), n_vec=8, cycle=qa.QUDA_MG_CYCLE_VCYCLE, setup_maxiter=300, setup_tol=1e-5)
    mg = qa.Multigrid(mp)
    try:
        rng = np.random.default_rng(9)
        b = rng.random(int(np.prod(X)) * 24)
        x = mg.cycle(b, ip)
        r = b - qa.mat(x, ip)
        assert np.linalg.norm(r) < 0.5 * np.linalg.norm(b)  # one V-cycle is a contraction on a random right-hand side
    finally:
        mg.free()


@pytest.mark.parametrize("mask", [0, 15, 9], ids=["unpartitioned", "self-neighbour-xyzt", "self-neighbour-xt"])
def test_hierarchy_against_oracle_restatement(qa, oracle, mask):
    """mask != 0: every dimension in the mask is treated as grid-decomposed with the process as its own neighbour, so
    the Galerkin probing (ghost-aware single-direction hops), the coarse-operator halo exchange and the solver's
    global reductions run through the multi-rank code path and must reproduce the same hierarchy.

    Every piece of a 3-level hierarchy against the CPU restatement of the reference's algorithms (oracle/qo_mg.c):
    block Gram-Schmidt of the device's own null vectors, the Galerkin links (calculateY, both the fine and the
    from-coarse variants), R, P and the coarse operator apply.  fp32 device arithmetic vs fp64 oracle; tolerances are
    relative to the largest element: 2e-5 for single kernels, 2e-4 after the 8-deep Gram-Schmidt recursion."""
    X, kappa, mu = (8, 8, 8, 16), 0.124, 0.005
    qa.lib().qudaAmdSetPartitionMask(mask)
    gauge, ip = _setup(qa, X, kappa, mu)
    nvec = [8, 8]
    mp = qa.multigrid_param(ip, n_level=3, geo_block=[(4, 4, 4, 4), (1, 1, 1, 2), (2, 2, 2, 2)], n_vec=8, setup_maxiter=100, setup_tol=1e-4)
    mg = qa.Multigrid(mp)
    rng = np.random.default_rng(17)

    def rel(a, b):
        return float(np.max(np.abs(a - b)) / np.max(np.abs(b)))

    try:
        assert mg.levels() == 3
        # lockstep set-up on every level, partitioned or not
        assert mg.level_info(0)["null_method"] == 1 and mg.level_info(1)["null_method"] == 2
        Yprev = Xprev = None
        for level in range(2):
            i = mg.level_info(level)
            Xf, Xc, bs, Ns, Nc, Nv, sbs = i["Xf"], i["Xc"], i["geo_bs"], i["fineSpin"], i["fineColor"], i["Nvec"], i["spin_bs"]
            assert Nv == nvec[level] and [Xf[d] // bs[d] for d in range(4)] == Xc
            # block orthonormalisation of the device's null vectors (lib/transfer_util.cu:168-363)
            B = np.stack([mg.null_vector(level, k) for k in range(Nv)], axis=-1)
            Vd = mg.V(level).astype(np.complex128)
            Vo = oracle.mg_block_orthogonalize(B, Xf, bs, Ns, Nc, Nv, sbs)
            assert rel(Vd, Vo) < 2e-4, (level, rel(Vd, Vo))
            # R and P with the device's V (lib/restrictor.cu:51-125, lib/prolongator.cu:42-116)
            phi = (rng.standard_normal((int(np.prod(Xf)), Ns, Nc)) + 1j * rng.standard_normal((int(np.prod(Xf)), Ns, Nc)))
            eta = (rng.standard_normal((int(np.prod(Xc)), 2, Nv)) + 1j * rng.standard_normal((int(np.prod(Xc)), 2, Nv)))
            assert rel(mg.apply(level, "R", phi), oracle.mg_restrict(phi, Vd, Xf, bs, Ns, Nc, Nv, sbs)) < 2e-5
            assert rel(mg.apply(level, "P", eta), oracle.mg_prolongate(eta, Vd, Xf, bs, Ns, Nc, Nv, sbs)) < 2e-5
            # Galerkin links (lib/coarse_op.cuh:1310-1498); the device stores -kappa Y
            Yd, Xd = mg.coarse_links(level)
            if level == 0:
                Yo, Xo = oracle.mg_coarse_op_fine(Vd, gauge, None, kappa, 2 * kappa * mu, Xf, bs, Nv)
            else:
                Yo, Xo = oracle.mg_coarse_op_coarse(Vd, Yprev, Xprev, kappa, Xf, bs, Nc, Nv)
            assert rel(Xd, Xo) < 2e-5, (level, rel(Xd, Xo))
            assert rel(Yd, -kappa * Yo) < 2e-5, (level, rel(Yd, -kappa * Yo))
            # coarse operator apply (lib/dslash_coarse.cu:50-290) with the device's own links
            Yref = Yd.astype(np.complex128) / (-kappa)
            want = oracle.mg_coarse_apply(eta, Yref, Xd.astype(np.complex128), kappa, Xc, Nv)
            assert rel(mg.apply(level + 1, "M", eta), want) < 2e-5
            Yprev, Xprev = Yref, Xd.astype(np.complex128)
        # the level-0 operator of the hierarchy is the oracle's tm_mat
        phi = rng.standard_normal((int(np.prod(X)), 4, 3)) + 1j * rng.standard_normal((int(np.prod(X)), 4, 3))
        want = oracle.tm_mat(gauge, np.ascontiguousarray(phi).view(np.float64).reshape(-1), list(X), kappa, mu, +1, 0).view(np.complex128).reshape(-1, 4, 3)
        assert rel(mg.apply(0, "M", phi), want) < 2e-5
        # and the preconditioned solve through the same (possibly partitioned) path
        b = rng.random(int(np.prod(X)) * 24)
        ip.inv_type_precondition = qa.QUDA_MG_INVERTER
        ip.preconditioner = mg.h
        ip.tol_precondition, ip.maxiter_precondition, ip.precondition_cycle, ip.omega = 1e-1, 1, 1, 1.0
        x = qa.invert(b, ip)
        assert _true_residual(oracle, gauge, X, kappa, mu, x, b) < 1e-10
        assert ip.iter < 40, ip.iter
    finally:
        mg.free()
        qa.lib().qudaAmdSetPartitionMask(0)


@pytest.mark.parametrize("X", [(16, 8, 8, 16), (32, 8, 8, 8), (16, 16, 8, 8)])
def test_restrictor_and_prolongator_in_the_x_neighbour_order(qa, oracle, X):
    """Lattices with four or more aggregates along x, where the transfer kernels walk the x-neighbour aggregates on one XCD
    (transfer.hip aggregate_of_block) — a re-numbering of the work-groups that must not change R or P: element-wise against the
    oracle with the device's own V, with the order switched on (default) — the hierarchy test above runs on 2 aggregates along x,
    where it is off."""
    kappa, mu = 0.124, 0.005
    gauge, ip = _setup(qa, X, kappa, mu)
    mp = qa.multigrid_param(ip, n_level=2, geo_block=(4, 4, 4, 4), n_vec=8, setup_maxiter=50, setup_tol=1e-3)
    mg = qa.Multigrid(mp)
    rng = np.random.default_rng(23)

    def rel(a, b):
        return float(np.max(np.abs(a - b)) / np.max(np.abs(b)))

    try:
        i = mg.level_info(0)
        Xf, Xc, bs, Ns, Nc, Nv, sbs = i["Xf"], i["Xc"], i["geo_bs"], i["fineSpin"], i["fineColor"], i["Nvec"], i["spin_bs"]
        assert Xc[0] % 4 == 0 and int(np.prod(Xc)) % 32 == 0   # the condition under which the order is active
        Vd = mg.V(0).astype(np.complex128)
        phi = (rng.standard_normal((int(np.prod(Xf)), Ns, Nc)) + 1j * rng.standard_normal((int(np.prod(Xf)), Ns, Nc)))
        eta = (rng.standard_normal((int(np.prod(Xc)), 2, Nv)) + 1j * rng.standard_normal((int(np.prod(Xc)), 2, Nv)))
        assert rel(mg.apply(0, "R", phi), oracle.mg_restrict(phi, Vd, Xf, bs, Ns, Nc, Nv, sbs)) < 2e-5
        assert rel(mg.apply(0, "P", eta), oracle.mg_prolongate(eta, Vd, Xf, bs, Ns, Nc, Nv, sbs)) < 2e-5
        # Galerkin links from the forward hops + hermitian completion (restrict4 in the same order)
        Yd, Xd = mg.coarse_links(0)
        Yo, Xo = oracle.mg_coarse_op_fine(Vd, gauge, None, kappa, 2 * kappa * mu, Xf, bs, Nv)
        assert rel(Xd, Xo) < 2e-5 and rel(Yd, -kappa * Yo) < 2e-5
    finally:
        mg.free()


@pytest.mark.parametrize("tb", ["periodic", "antiperiodic"])
def test_twelve_real_links_in_the_fast_setup_path(qa, oracle, tb):
    """Sloppy and preconditioner links stored as 12 reals (reconstruct_sloppy = reconstruct_precondition = QUDA_RECONSTRUCT_12, what production
    runs of the reference use): the multi-right-hand-side stencil rebuilds the third row while it stages the links, the direct Galerkin product
    when it loads them — so the set-up keeps its lockstep solves (null_method 1) and its batched products.  Antiperiodic: the rebuilt row of the
    boundary t links carries the folded sign.  Stencil per right-hand side against the host tm_mat, Y and X against the oracle's calculateY
    (2e-5), verify, and an MG-GCR solve with the residual recomputed on the host."""
    X, kappa, mu = (8, 8, 8, 16), 0.124, 0.005
    gauge = smooth_gauge(X, 0.35)
    V, Vh = int(np.prod(X)), int(np.prod(X)) // 2
    if tb == "antiperiodic":   # the host field carries the boundary, as the reference's applyGaugeFieldScaling leaves it (tests/test_util.cpp:1118-1141)
        t_of = np.arange(Vh) // (X[0] // 2 * X[1] * X[2])
        gauge[3].reshape(2, Vh, 18)[:, t_of == X[3] - 1, :] *= -1.0
    gp = qa.gauge_param(X, cuda_prec=8, prec_sloppy=4, prec_precondition=4, recon_sloppy=qa.QUDA_RECONSTRUCT_12,
                        t_boundary=qa.QUDA_ANTI_PERIODIC_T if tb == "antiperiodic" else qa.QUDA_PERIODIC_T)
    assert gp.reconstruct_precondition == qa.QUDA_RECONSTRUCT_12
    qa.load_gauge(gauge, gp)
    ip = qa.invert_param(qa.QUDA_TWISTED_MASS_DSLASH, kappa, mu, +1, "ee", 0, cuda_prec=8, prec_sloppy=4, prec_precondition=4, solution_type=qa.QUDA_MAT_SOLUTION)
    ip.solve_type, ip.inv_type, ip.gcrNkrylov, ip.tol, ip.maxiter, ip.reliable_delta, ip.verbosity = qa.QUDA_DIRECT_SOLVE, qa.QUDA_GCR_INVERTER, 20, 1e-10, 2000, 1e-4, qa.QUDA_SILENT
    mp = qa.multigrid_param(ip, n_level=2, geo_block=(4, 4, 4, 4), n_vec=8, setup_maxiter=100, setup_tol=1e-4, smoother_pc=True)
    mg = qa.Multigrid(mp)
    rng = np.random.default_rng(41)

    def rel(a, b):
        return float(np.max(np.abs(a - b)) / np.max(np.abs(b)))

    try:
        i = mg.level_info(0)
        assert i["null_method"] == 1
        phi = (rng.standard_normal((8, V, 4, 3)) + 1j * rng.standard_normal((8, V, 4, 3))).astype(np.complex64)
        got, _ = mg.apply_block(0, phi)
        oracle.set_threads(8)
        try:
            for k in range(8):
                v = np.ascontiguousarray(phi[k].astype(np.complex128)).view(np.float64).reshape(-1)
                want = oracle.tm_mat(gauge, v, list(X), kappa, mu, +1, 0)
                assert rel(got[k], want.view(np.complex128).reshape(-1, 4, 3)) < 2e-5, k
        finally:
            oracle.set_threads(1)
        Xf, bs, Nv = i["Xf"], i["geo_bs"], i["Nvec"]
        Vd = mg.V(0).astype(np.complex128)
        Yd, Xd = mg.coarse_links(0)
        Yo, Xo = oracle.mg_coarse_op_fine(Vd, gauge, None, kappa, 2 * kappa * mu, Xf, bs, Nv)
        assert rel(Xd, Xo) < 2e-5 and rel(Yd, -kappa * Yo) < 2e-5
        assert max(mg.verify()) < 1e-4
        ip.inv_type_precondition, ip.preconditioner = qa.QUDA_MG_INVERTER, mg.h
        ip.tol_precondition, ip.maxiter_precondition, ip.precondition_cycle, ip.omega = 1e-1, 1, 1, 1.0
        b = rng.random(V * 24)
        x = qa.invert(b, ip)
        oracle.set_threads(8)
        try:
            res = float(np.linalg.norm(b - oracle.tm_mat(gauge, x, list(X), kappa, mu, +1, 0)) / np.linalg.norm(b))
        finally:
            oracle.set_threads(1)
        assert res < 1e-10 and ip.iter < 40, (res, ip.iter)
    finally:
        mg.free()


@pytest.mark.parametrize("X,bs,nvec", [((8, 8, 8, 8), (4, 4, 4, 2), 32), ((8, 8, 8, 16), (2, 4, 4, 4), 24), ((8, 8, 8, 8), (4, 4, 2, 2), 8), ((8, 8, 8, 8), (4, 4, 4, 4), 32)],
                         ids=["4x4x4x2-nvec32", "2x4x4x4-nvec24", "4x4x2x2-nvec8", "4x4x4x4-nvec32"])
def test_transfer_kernels_on_other_aggregate_shapes(qa, oracle, X, bs, nvec):
    """The barrier-free restrictor and the pipelined prolongator (transfer.hip restrict_stream_kernel / prolong_kernel) with aggregates of 128 and
    64 sites (two waves / one wave per work-group, one parity per wave in the parity-major order) and 8 / 24 / 32 vectors: R and P element-wise
    against the oracle with the device's own V, then an MG-GCR solve with the even-odd smoother — whose cycle restricts and prolongates ONE parity —
    to 1e-10 with the residual recomputed on the host."""
    kappa, mu = 0.124, 0.005
    gauge, ip = _setup(qa, X, kappa, mu)
    mp = qa.multigrid_param(ip, n_level=2, geo_block=bs, n_vec=nvec, setup_maxiter=100, setup_tol=1e-4, smoother_pc=True)
    mg = qa.Multigrid(mp)
    rng = np.random.default_rng(29)

    def rel(a, b):
        return float(np.max(np.abs(a - b)) / np.max(np.abs(b)))

    try:
        i = mg.level_info(0)
        Xf, Xc, gbs, Ns, Nc, Nv, sbs = i["Xf"], i["Xc"], i["geo_bs"], i["fineSpin"], i["fineColor"], i["Nvec"], i["spin_bs"]
        assert list(gbs) == list(bs) and Nv == nvec
        Vd = mg.V(0).astype(np.complex128)
        phi = (rng.standard_normal((int(np.prod(Xf)), Ns, Nc)) + 1j * rng.standard_normal((int(np.prod(Xf)), Ns, Nc)))
        eta = (rng.standard_normal((int(np.prod(Xc)), 2, Nv)) + 1j * rng.standard_normal((int(np.prod(Xc)), 2, Nv)))
        assert rel(mg.apply(0, "R", phi), oracle.mg_restrict(phi, Vd, Xf, gbs, Ns, Nc, Nv, sbs)) < 2e-5
        assert rel(mg.apply(0, "P", eta), oracle.mg_prolongate(eta, Vd, Xf, gbs, Ns, Nc, Nv, sbs)) < 2e-5
        # Galerkin operator: the batched product on the matrix cores where the aggregates are 4^4 (Nvec 8 / 24 / 32), probing elsewhere
        Yd, Xd = mg.coarse_links(0)
        Yo, Xo = oracle.mg_coarse_op_fine(Vd, gauge, None, kappa, 2 * kappa * mu, Xf, gbs, Nv)
        assert rel(Xd, Xo) < 2e-5 and rel(Yd, -kappa * Yo) < 2e-5
        assert max(mg.verify()) < 1e-4
        ip.inv_type_precondition, ip.preconditioner = qa.QUDA_MG_INVERTER, mg.h
        ip.tol_precondition, ip.maxiter_precondition, ip.precondition_cycle, ip.omega = 1e-1, 1, 1, 1.0
        b = rng.random(int(np.prod(X)) * 24)
        x = qa.invert(b, ip)
        oracle.set_threads(8)
        try:
            res = float(np.linalg.norm(b - oracle.tm_mat(gauge, x, list(X), kappa, mu, +1, 0)) / np.linalg.norm(b))
        finally:
            oracle.set_threads(1)
        assert res < 1e-10 and ip.iter < 60, (res, ip.iter)
    finally:
        mg.free()


@pytest.mark.parametrize("mask", [0, 15, 9], ids=["unpartitioned", "self-neighbour-xyzt", "self-neighbour-xt"])
@pytest.mark.parametrize("nvec,nrhs_list", [(8, (8, 16, 24, 32)), (24, (24, 8))], ids=["n16", "n48"])
def test_block_coarse_operator_on_mfma(qa, oracle, nvec, nrhs_list, mask):
    """The multi-right-hand-side coarse operator on the matrix cores (csrc/block.hip, v_mfma_f32_16x16x4_f32; reference: the
    multi-source 5th dimension of coarseDslashKernel, lib/dslash_coarse.cu:294-333) against the oracle's restatement of the
    reference's CPU coarse operator (lib/dslash_coarse.cu:50-290) applied to every right-hand side separately, with the device's
    own links, and against the single-vector device kernel.  fp32 MFMA is exact fp32 arithmetic: same 2e-5 bar as the
    single-vector kernel.  n = 2 Nvec = 16 and the production 48 x 24 shape.
    mask != 0: the dimensions in the mask are grid-decomposed with the process as its own neighbour — the panels
    of the face sites travel through the pack kernel and the exchange into the ghost zone behind the field and the MFMA kernel reaches
    them through the neighbour table (reference: ghost of the multi-source coarse field, lib/dslash_coarse.cu:68-137); the result must
    not change."""
    X, kappa, mu = (8, 8, 8, 16), 0.124, 0.005
    qa.lib().qudaAmdSetPartitionMask(mask)
    gauge, ip = _setup(qa, X, kappa, mu)
    mp = qa.multigrid_param(ip, n_level=2, geo_block=[(4, 4, 4, 4), (2, 2, 2, 2)], n_vec=nvec, setup_maxiter=50, setup_tol=1e-3)
    mg = qa.Multigrid(mp)
    rng = np.random.default_rng(23)

    def rel(a, b):
        return float(np.max(np.abs(a - b)) / np.max(np.abs(b)))

    try:
        i = mg.level_info(0)
        Xc, Nv = i["Xc"], i["Nvec"]
        Vc = int(np.prod(Xc))
        Yd, Xd = mg.coarse_links(0)
        Yref = Yd.astype(np.complex128) / (-kappa)
        for nrhs in nrhs_list:
            eta = (rng.standard_normal((nrhs, Vc, 2, Nv)) + 1j * rng.standard_normal((nrhs, Vc, 2, Nv))).astype(np.complex64)
            # right-hand sides of very different size: a slip in the column map would mix them visibly
            eta *= (10.0 ** rng.integers(-2, 3, size=nrhs)).astype(np.float32)[:, None, None, None]
            got, _ = mg.apply_block(1, eta)
            for k in range(nrhs):
                want = oracle.mg_coarse_apply(eta[k].astype(np.complex128), Yref, Xd.astype(np.complex128), kappa, Xc, Nv)
                assert rel(got[k], want) < 2e-5, (nrhs, k, rel(got[k], want))
                if k in (0, nrhs - 1):
                    assert rel(got[k], mg.apply(1, "M", eta[k])) < 2e-5
        if mask:
            assert qa.comm_stats()["block_exchanges"] > 0
            # and the set-up itself went through the lockstep solves on the partitioned lattice
            assert mg.level_info(0)["null_method"] == 1
    finally:
        mg.free()
        qa.lib().qudaAmdSetPartitionMask(0)


def test_block_bicgstab_null_vectors_give_the_same_hierarchy_quality(qa, oracle):
    """The coarse-level null vectors now come from ONE lockstep block BiCGstab on the MFMA operator (multigrid.cpp,
    MG::generateNullVectors) instead of Nvec sequential solves (QUDA_AMD_BLOCK_COARSE=0 keeps the latter): the reference's
    MG::verify() identities (lib/multigrid.cpp:372-486) hold on every level and the 3-level MG-GCR still converges to 1e-10
    in the same few iterations, the solution checked with the oracle's tm_mat."""
    X, kappa, mu = (8, 8, 8, 16), 0.124, 0.005
    gauge, ip = _setup(qa, X, kappa, mu)
    mp = qa.multigrid_param(ip, n_level=3, geo_block=[(4, 4, 4, 4), (1, 1, 1, 2), (2, 2, 2, 2)], n_vec=8, setup_maxiter=100, setup_tol=1e-4)
    mg = qa.Multigrid(mp)
    try:
        dev = mg.verify()
        assert max(dev) < 1e-4, dev
        b = np.random.default_rng(3).random(int(np.prod(X)) * 24)
        ip.inv_type_precondition = qa.QUDA_MG_INVERTER
        ip.preconditioner = mg.h
        ip.tol_precondition, ip.maxiter_precondition, ip.precondition_cycle, ip.omega = 1e-1, 1, 1, 1.0
        x = qa.invert(b, ip)
        assert _true_residual(oracle, gauge, X, kappa, mu, x, b) < 1e-10
        assert ip.iter < 40, ip.iter
    finally:
        mg.free()


def test_mg_gcr_where_plain_gcr_stalls(qa, oracle):
    """The regime multigrid is for (reference tests/multigrid_invert_test.cpp:529-577 for the check): the synthetic
    warm-start field at its critical kappa (0.147 at 32^4, mu = 0.001: smallest singular value ~ 2 kappa mu; scan in
    profiles/r02_mg_kappa_scan_32x4_c.json — plain GCR(20) needs 11 258 iterations there and stagnates beyond it).  Plain GCR is
    nowhere near converged after 1000 iterations; the 3-level MG-GCR reaches 1e-10, the residual recomputed on the HOST with the
    oracle's tm_mat."""
    X, kappa, mu = (32, 32, 32, 32), 0.147, 0.001
    gauge, ip = _setup(qa, X, kappa, mu)
    b = np.random.default_rng(5).random(int(np.prod(X)) * 24)
    ip.inv_type_precondition = qa.QUDA_INVALID_ENUM
    ip.maxiter = 1000
    x_plain = qa.invert(b, ip)
    res_plain = _true_residual(oracle, gauge, X, kappa, mu, x_plain, b)
    assert ip.iter >= 1000 and res_plain > 1e-6, (ip.iter, res_plain)
    mp = qa.multigrid_param(ip, n_level=3, geo_block=[(4, 4, 4, 4), (2, 2, 2, 2), (2, 2, 2, 2)], n_vec=24, setup_maxiter=500, setup_tol=5e-6, smoother_pc=True)
    mg = qa.Multigrid(mp)
    try:
        ip.inv_type_precondition = qa.QUDA_MG_INVERTER
        ip.preconditioner = mg.h
        ip.tol_precondition, ip.maxiter_precondition, ip.precondition_cycle, ip.omega = 1e-1, 1, 1, 1.0
        ip.maxiter, ip.tol = 2000, 5e-11
        x = qa.invert(b, ip)
        res = _true_residual(oracle, gauge, X, kappa, mu, x, b)
        print("critical kappa: plain GCR %.1e after 1000 iterations; MG-GCR %d iterations, %.3f s in the solver, host residual %.2e" % (res_plain, ip.iter, ip.secs, res))
        assert res < 1e-10, res
        assert ip.iter < 400, ip.iter
        # set-up refinement: the null-vector solves of the plain set-up stop at their 500-iteration cap here; three
        # inverse-iteration passes through the hierarchy (multigrid_solver::refine) bring the outer solve from ~100 to well under 40 iterations
        plain_iters = ip.iter
        secs = mg.refine(3, 1)
        x = qa.invert(b, ip)
        res = _true_residual(oracle, gauge, X, kappa, mu, x, b)
        print("after 3 refinement passes (%.2f s): MG-GCR %d iterations (plain set-up %d), %.3f s in the solver, host residual %.2e" % (secs, ip.iter, plain_iters, ip.secs, res))
        assert res < 1e-10, res
        assert ip.iter <= 40 and ip.iter < plain_iters, (ip.iter, plain_iters)
    finally:
        mg.free()


def test_outer_even_odd_solve_with_up_and_down_hierarchies(qa, oracle):
    """The QKXTM production shape (reference lib/interface_quda.cpp:6041, :6389-6520; qkxtm/CalcMG_2pt3pt_EvenOdd.cpp:649-747):
    outer GCR on the even-odd preconditioned system (solve_type = QUDA_DIRECT_PC_SOLVE, full-field MAT solution through
    prepare / reconstruct) preconditioned by a multigrid hierarchy, one hierarchy per twist flavour (preconditionerUP for
    +mu, preconditionerDN for -mu), the flavour flipped between solves.  Residuals are recomputed with the oracle's tm_mat."""
    X, kappa, mu = (8, 8, 8, 16), 0.124, 0.005
    gauge, ip = _setup(qa, X, kappa, mu)
    rng = np.random.default_rng(23)
    b = rng.random(int(np.prod(X)) * 24)
    ip.solve_type = qa.QUDA_DIRECT_PC_SOLVE
    hier = {}
    try:
        for flavor in (+1, -1):
            ipm = qa.invert_param(qa.QUDA_TWISTED_MASS_DSLASH, kappa, mu, flavor, "ee", 0, cuda_prec=8, prec_sloppy=4, prec_precondition=4,
                                  solution_type=qa.QUDA_MAT_SOLUTION)
            ipm.solve_type = qa.QUDA_DIRECT_SOLVE   # the MG-internal parameter set (reference lib/interface_quda.cpp:2183)
            ipm.inv_type, ipm.gcrNkrylov, ipm.tol, ipm.maxiter, ipm.reliable_delta, ipm.verbosity = qa.QUDA_GCR_INVERTER, 20, 1e-10, 2000, 1e-4, qa.QUDA_SILENT
            mp = qa.multigrid_param(ipm, n_level=3, geo_block=[(4, 4, 4, 4), (1, 1, 1, 2), (2, 2, 2, 2)], n_vec=8, setup_maxiter=300, setup_tol=1e-5, smoother_pc=True,
                                    coarse_matpc=True)   # single-parity injection, as the harness configures an outer even-odd solve
            hier[flavor] = (qa.Multigrid(mp), ipm, mp)
        ip.preconditionerUP, ip.preconditionerDN = hier[+1][0].h, hier[-1][0].h
        for flavor in (+1, -1, +1):
            ip.twist_flavor = qa.QUDA_TWIST_PLUS if flavor > 0 else qa.QUDA_TWIST_MINUS
            ip.inv_type_precondition = qa.QUDA_INVALID_ENUM
            ip.preconditioner = None
            qa.invert(b, ip)
            plain = ip.iter
            ip.inv_type_precondition = qa.QUDA_MG_INVERTER
            ip.preconditioner = ip.preconditionerUP if flavor > 0 else ip.preconditionerDN
            ip.tol_precondition, ip.maxiter_precondition, ip.precondition_cycle, ip.omega = 1e-1, 1, 1, 1.0
            x = qa.invert(b, ip)
            oracle.set_threads(8)
            res = float(np.linalg.norm(b - oracle.tm_mat(gauge, x, list(X), kappa, mu, flavor, 0)) / np.linalg.norm(b))
            oracle.set_threads(1)
            print("outer even-odd MG-GCR flavour %+d: %d iterations (plain even-odd GCR %d), true residual %.2e" % (flavor, ip.iter, plain, res))
            assert res < 1e-10, (flavor, res)
            assert ip.iter * 3 < plain, (flavor, ip.iter, plain)
        # the same hierarchies under a full-system outer solve (outer QUDA_MAT_SOLUTION / inner QUDA_MATPC_SOLUTION, lib/multigrid.cpp:513-560)
        ip.solve_type = qa.QUDA_DIRECT_SOLVE
        ip.twist_flavor = qa.QUDA_TWIST_PLUS
        ip.preconditioner = ip.preconditionerUP
        x = qa.invert(b, ip)
        oracle.set_threads(8)
        res = float(np.linalg.norm(b - oracle.tm_mat(gauge, x, list(X), kappa, mu, +1, 0)) / np.linalg.norm(b))
        oracle.set_threads(1)
        assert res < 1e-10 and ip.iter < 30, (res, ip.iter)
    finally:
        for h, _, _ in hier.values():
            h.free()


def test_null_vector_persistence(qa, tmp_path):
    """vec_outfile / vec_infile (reference MG::saveVectors / loadVectors, lib/multigrid.cpp:607-691): a hierarchy rebuilt from the
    saved null vectors with compute_null_vector = NO has bit-identical transfer operators and coarse links on every level."""
    X, kappa, mu = (8, 8, 8, 16), 0.124, 0.005
    _, ip = _setup(qa, X, kappa, mu)
    base = str(tmp_path / "nullvec").encode()
    mp = qa.multigrid_param(ip, n_level=3, geo_block=[(4, 4, 4, 4), (1, 1, 1, 2), (2, 2, 2, 2)], n_vec=8, setup_maxiter=100, setup_tol=1e-4)
    mp.vec_outfile = base
    mg = qa.Multigrid(mp)
    V0, V1 = mg.V(0), mg.V(1)
    Y0, X0 = mg.coarse_links(0)
    Y1, X1 = mg.coarse_links(1)
    mg.free()
    assert (tmp_path / "nullvec_level_0").exists() and (tmp_path / "nullvec_level_1").exists()
    mp2 = qa.multigrid_param(ip, n_level=3, geo_block=[(4, 4, 4, 4), (1, 1, 1, 2), (2, 2, 2, 2)], n_vec=8)
    mp2.compute_null_vector = qa.QUDA_COMPUTE_NULL_VECTOR_NO
    mp2.vec_infile = base
    mg2 = qa.Multigrid(mp2)
    try:
        assert np.array_equal(mg2.V(0), V0) and np.array_equal(mg2.V(1), V1)
        Yb, Xb = mg2.coarse_links(0)
        assert np.array_equal(Yb, Y0) and np.array_equal(Xb, X0)
        Yb, Xb = mg2.coarse_links(1)
        assert np.array_equal(Yb, Y1) and np.array_equal(Xb, X1)
        assert mp2.secs < mp.secs
    finally:
        mg2.free()


def test_half_precision_storage_of_the_hierarchy(qa, oracle):
    """Opt-in fp16 mirrors of V and of the coarse links (qudaAmdMultigridSetHalfStorage): the cycle streams half the bytes, the
    outer fp64 GCR still reaches 1e-10 with the oracle-verified residual, and switching back restores the fp32 behaviour."""
    X, kappa, mu = (16, 8, 8, 16), 0.124, 0.005
    gauge, ip = _setup(qa, X, kappa, mu)
    b = np.random.default_rng(29).random(int(np.prod(X)) * 24)
    mp = qa.multigrid_param(ip, n_level=3, geo_block=[(4, 4, 4, 4), (2, 2, 2, 2), (2, 2, 2, 2)], n_vec=8, setup_maxiter=300, setup_tol=1e-5, smoother_pc=True)
    mg = qa.Multigrid(mp)
    try:
        ip.inv_type_precondition, ip.preconditioner = qa.QUDA_MG_INVERTER, mg.h
        ip.tol_precondition, ip.maxiter_precondition, ip.precondition_cycle, ip.omega = 1e-1, 1, 1, 1.0
        x32 = qa.invert(b, ip)
        it32 = ip.iter
        mg.set_half_storage(True)
        x16 = qa.invert(b, ip)
        it16 = ip.iter
        assert _true_residual(oracle, gauge, X, kappa, mu, x16, b) < 1e-10
        assert it16 <= it32 + 2, (it16, it32)
        mg.set_half_storage(False)
        x32b = qa.invert(b, ip)
        assert ip.iter == it32 and np.array_equal(x32b, x32)
        dev = mg.verify()
        assert max(dev) < 1e-4
    finally:
        mg.set_half_storage(False)
        mg.free()


def test_c5_full_size_on_one_gpu(qa, oracle):
    """BASELINE.json configs[4] at full size — 48^3 x 96, three levels by the reference's blocking rule (4^4, then 2^3 x 4 because
    12 / 4 is odd; lib/transfer.cpp:31-44), 24 null vectors — resident on ONE MI355X (the reference quotes it on 8 GPUs).  The
    size-independent property checked is the one the reference's harness checks (tests/multigrid_invert_test.cpp:529-577): the
    HOST operator (oracle tm_mat) applied to the returned solution reproduces the source to the requested 1e-10."""
    import time
    from synth import smooth_gauge_cayley
    X, kappa, mu = (48, 48, 48, 96), 0.124, 0.005
    gauge = smooth_gauge_cayley(X, 0.35, workers=16)
    gp = qa.gauge_param(X, cuda_prec=8, prec_sloppy=4, prec_precondition=4, t_boundary=qa.QUDA_PERIODIC_T)
    qa.load_gauge(gauge, gp)
    ip = qa.invert_param(qa.QUDA_TWISTED_MASS_DSLASH, kappa, mu, +1, "ee", 0, cuda_prec=8, prec_sloppy=4, prec_precondition=4,
                         solution_type=qa.QUDA_MAT_SOLUTION)
    ip.solve_type, ip.inv_type, ip.gcrNkrylov, ip.tol, ip.maxiter, ip.verbosity = qa.QUDA_DIRECT_SOLVE, qa.QUDA_GCR_INVERTER, 20, 1e-10, 5000, qa.QUDA_SILENT
    mp = qa.multigrid_param(ip, n_level=3, geo_block=[(4, 4, 4, 4), (2, 2, 2, 4), (2, 2, 2, 2)], n_vec=24, setup_maxiter=500, setup_tol=5e-6, smoother_pc=True)
    mg = qa.Multigrid(mp)
    try:
        assert mg.level_info(1)["Xc"] == [6, 6, 6, 6]
        ip.inv_type_precondition, ip.preconditioner = qa.QUDA_MG_INVERTER, mg.h
        ip.tol_precondition, ip.maxiter_precondition, ip.precondition_cycle, ip.omega = 1e-1, 1, 1, 1.0
        b = np.random.default_rng(5).random(int(np.prod(X)) * 24)
        qa.invert(b, ip)
        t0 = time.perf_counter()
        x = qa.invert(b, ip)
        wall = time.perf_counter() - t0
        oracle.set_threads(16)
        try:
            res = float(np.linalg.norm(b - oracle.tm_mat(gauge, x, list(X), kappa, mu, +1, 0)) / np.linalg.norm(b))
        finally:
            oracle.set_threads(1)
        print("48^3 x 96 on one GPU: setup %.2f s, MG-GCR %d iterations, %.3f s wall (%.3f s in the solver), host-verified |r|/|b| = %.2e"
              % (mp.secs, ip.iter, wall, ip.secs, res))
        assert res < 1e-10 and ip.iter < 40, (res, ip.iter)
    finally:
        mg.free()


def _write_null_vectors(path, X, B):
    """a null-vector file of this library (csrc/multigrid.cpp NullVecHeader: 64-byte header + Nvec site-major fp32 vectors), so a test can
    hand the hierarchy the vectors it wants through vec_infile / compute_null_vector = NO (reference MG::loadVectors, lib/multigrid.cpp:639-691)"""
    import struct
    with open(path, "wb") as f:
        f.write(struct.pack("<8s4i4i4i2i", b"QAMDNV02", *X, 4, 3, len(B), 4, 1, 1, 1, 1, 0, 0x01020304))
        for v in B:
            f.write(np.ascontiguousarray(v, dtype=np.complex64).tobytes())


@pytest.mark.parametrize("eps,expect_fallback", [(3e-2, False), (1e-3, True), ("mixed", True)], ids=["cond-1e2", "cond-1e3", "ill-conditioned-in-some-blocks"])
def test_block_orthonormalisation_of_nearly_dependent_vectors(qa, oracle, tmp_path, eps, expect_fallback):
    """ADVICE r2 (transfer.hip pivot test): locally coherent near-null vectors make the block Gram matrices ill-conditioned.  The fp32
    CholeskyQR2 kernel has to notice what is beyond it — pivots at the round-off level of an fp32 Gram matrix, a first round that is
    not close to orthonormal — and hand exactly those blocks to Gram-Schmidt.  Vectors v_k = v_0 + eps r_k go in through vec_infile;
    checked: P^dag P = 1 per block to fp32 level (|V^dag V - 1| element-wise), the span (identity (1) of MG::verify), agreement with
    the oracle's fp64 Gram-Schmidt (lib/transfer_util.cu:328-363) at the accuracy the conditioning allows, and whether the fall-back ran."""
    X, kappa, mu = (8, 8, 8, 8), 0.124, 0.005
    _, ip = _setup(qa, X, kappa, mu)
    V4, nvec = int(np.prod(X)), 8
    rng = np.random.default_rng(77)
    v0 = (rng.standard_normal((V4, 4, 3)) + 1j * rng.standard_normal((V4, 4, 3))).astype(np.complex64)
    c = np.indices(X[::-1]).reshape(4, -1)[::-1]                       # x, y, z, t of the lexicographic sites
    lex = ((c[3] * X[2] + c[2]) * X[1] + c[1]) * X[0] + c[0]
    eo = ((c[0] + c[1] + c[2] + c[3]) & 1) * (V4 // 2) + lex // 2      # their place in the even-odd host order
    amp = np.full(V4, 0.3 if eps == "mixed" else eps)
    if eps == "mixed":
        amp[eo[c[3] < 4]] = 1e-3       # nearly dependent on the time slices 0..3 only: half of the aggregates
    B = [v0] + [(v0 + amp[:, None, None] * (rng.standard_normal((V4, 4, 3)) + 1j * rng.standard_normal((V4, 4, 3)))).astype(np.complex64) for _ in range(nvec - 1)]
    base = str(tmp_path / "nv")
    _write_null_vectors(base + "_level_0", X, B)
    mp = qa.multigrid_param(ip, n_level=2, geo_block=(4, 4, 4, 4), n_vec=nvec)
    mp.compute_null_vector = qa.QUDA_COMPUTE_NULL_VECTOR_NO
    mp.vec_infile = base.encode()
    mg = qa.Multigrid(mp)
    try:
        i = mg.level_info(0)
        Xf, bs = i["Xf"], i["geo_bs"]
        nfb = mg.ortho_fallback_blocks(0)
        assert (nfb > 0) == expect_fallback, nfb
        if eps == "mixed":
            assert nfb < 2 * int(np.prod(i["Xc"])), nfb     # only the ill-conditioned blocks
        Vd = mg.V(0).astype(np.complex128)
        # per (aggregate, chirality) Gram matrix of the device's V
        agg = (((c[3] // bs[3]) * (X[2] // bs[2]) + c[2] // bs[2]) * (X[1] // bs[1]) + c[1] // bs[1]) * (X[0] // bs[0]) + c[0] // bs[0]
        worst = 0.0
        for a in range(int(agg.max()) + 1):
            sites = eo[agg == a]
            for chi in range(2):
                M = Vd[sites][:, 2 * chi:2 * chi + 2].reshape(-1, nvec)
                worst = max(worst, float(np.max(np.abs(M.conj().T @ M - np.eye(nvec)))))
        bound = 5e-4 if expect_fallback else 1e-5     # modified Gram-Schmidt in fp32 loses eps x cond; CholeskyQR2 ends at round-off
        print("block orthonormalisation eps=%s: %d blocks to Gram-Schmidt, max |V^dag V - 1| = %.2e" % (eps, nfb, worst))
        assert worst < bound, worst
        dev = mg.verify()
        assert dev[0] < (2e-3 if expect_fallback else 1e-4) and dev[1] < bound * 10, dev
        Bd = np.stack([mg.null_vector(0, k) for k in range(nvec)], axis=-1)
        Vo = oracle.mg_block_orthogonalize(Bd, Xf, bs, 4, 3, nvec, 2)
        cond = 1.0 / (1e-3 if expect_fallback else 3e-2)   # fp32 input rounding is amplified by the conditioning in either algorithm
        assert float(np.max(np.abs(Vd - Vo)) / np.max(np.abs(Vo))) < 2e-6 * cond * 10
    finally:
        mg.free()


def test_lockstep_bicgstab_variants_agree():
    """The lockstep BiCGstab of the set-up in its three forms — separate sweeps with rho' from the updated residual (round 2), fused sweeps with rho' by
    linearity (blockblas::bicgstabDots / bicgstabFused), and the inner products taken in the stencil's epilogue (dslash.h FineBlockDots) — is the same
    recurrence in exact arithmetic: same lockstep iteration count (+- 1), null vectors equal up to what fp32 round-off does to a dozen BiCGstab steps
    (sampled components within 10 % of the largest; measured 2 %), same |M v| / |v|, same MG-GCR iteration count (+- 1).
    The switches are read once per process, hence three child processes (tools/lockstep_variants.py), twisted mass and twisted clover."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    runs = {}
    for name, env in (("separate", dict(QUDA_AMD_BLOCK_BICG_FUSED="0")), ("fused", dict(QUDA_AMD_BLOCK_FINE_DOTS="0")), ("epilogue", {})):
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "lockstep_variants.py")], capture_output=True, text=True, timeout=600, env=dict(os.environ, **env))
        assert r.returncode == 0, (name, r.stdout[-2000:], r.stderr[-2000:])
        line = [l for l in r.stdout.splitlines() if l.startswith("LOCKSTEP ")][-1]
        runs[name] = json.loads(line[9:])
    for action in ("tm", "tmc"):
        ref = runs["separate"][action]
        assert ref["null_method"] == 1 and ref["true_res"] < 1e-10
        for name in ("fused", "epilogue"):
            got = runs[name][action]
            assert got["null_method"] == 1
            assert abs(got["null_iters"] - ref["null_iters"]) <= 1, (action, name, got["null_iters"], ref["null_iters"])
            assert abs(got["iters"] - ref["iters"]) <= 1 and got["true_res"] < 1e-10, (action, name, got["iters"], ref["iters"], got["true_res"])
            assert max(got["quality"]) < 1.5 * max(ref["quality"]), (action, name, got["quality"], ref["quality"])
            fa, fb = np.array(got["fingerprint"]), np.array(ref["fingerprint"])
            assert np.max(np.abs(fa - fb)) < 0.1 * np.max(np.abs(fb)), (action, name, np.max(np.abs(fa - fb)), np.max(np.abs(fb)))
        print(action, {n: (runs[n][action]["null_iters"], runs[n][action]["iters"], "%.2e" % max(runs[n][action]["quality"])) for n in runs})


def test_opt_in_xy_tile_of_the_multi_rhs_stencil():
    """QUDA_AMD_BLOCK_FINE_XYTILE=1 (dslash.hip fine_block_kernel<8, 0, 1>: the x / y neighbour panels of an 8 x 4 tile staged in LDS; measured slower, off by
    default) stays correct: every right-hand side against the host tm_mat on lattices the tiles cover, unpartitioned and under partition masks whose x / y
    faces come out of the ghost zone.  The switch is read once per process, hence a child process (tools/fine_block_xy_check.py)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fine_block_xy_check.py")], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, QUDA_AMD_BLOCK_FINE_XYTILE="1"))
    assert r.returncode == 0 and "XYCHECK ok" in r.stdout, (r.stdout[-1500:], r.stderr[-1500:])


def test_hierarchy_from_a_scidac_vector_file_built_by_hand(qa, oracle, tmp_path):
    """f4 (reference MG::loadVectors -> read_spinor_field, lib/multigrid.cpp:639-691, lib/qio_field.cpp:198-260): the null vectors arrive in a
    SciDAC / QIO single-file container that THIS TEST assembles byte by byte from the format description — LIME records (144-byte
    big-endian headers, data padded to 8 bytes), one field record with datacount = Nvec, global lexicographic sites, per site vector 0 ..
    Nvec-1, big-endian fp32, QIO's rotated-CRC checksum pair — not with the library's writer.  The hierarchy built from it (compute_null_vector
    = NO) must restrict and prolongate like the oracle's R / P (lib/restrictor.cu:51-125, lib/prolongator.cu:42-116) built from the SAME
    vectors through the oracle's block Gram-Schmidt (lib/transfer_util.cu:328-363)."""
    import struct
    import zlib
    X, kappa, mu, nvec, bs = (8, 8, 8, 8), 0.124, 0.005, 8, (4, 4, 4, 4)
    _, ip = _setup(qa, X, kappa, mu)
    V4 = int(np.prod(X))
    rng = np.random.default_rng(91)
    B = [(rng.standard_normal((V4, 4, 3)) + 1j * rng.standard_normal((V4, 4, 3))).astype(np.complex64) for _ in range(nvec)]   # even-odd host order
    c = np.indices(X[::-1]).reshape(4, -1)[::-1]
    lex = ((c[3] * X[2] + c[2]) * X[1] + c[1]) * X[0] + c[0]
    eo = ((c[0] + c[1] + c[2] + c[3]) & 1) * (V4 // 2) + lex // 2
    payload = np.zeros((V4, nvec, 24), dtype=">f4")
    for v in range(nvec):
        payload[lex, v] = B[v].reshape(V4, 12).view(np.float32).reshape(V4, 24)[eo]
    suma = sumb = 0
    for s in range(V4):
        crc = zlib.crc32(payload[s].tobytes()) & 0xFFFFFFFF
        r29, r31 = s % 29, s % 31
        suma ^= ((crc << r29) | (crc >> (32 - r29))) & 0xFFFFFFFF
        sumb ^= ((crc << r31) | (crc >> (32 - r31))) & 0xFFFFFFFF

    def rec(rtype, data, mb, me):
        head = struct.pack(">IHHQ", 0x456789AB, 1, (0x8000 if mb else 0) | (0x4000 if me else 0), len(data)) + rtype.encode().ljust(128, b"\0")
        return head + data + b"\0" * (-len(data) % 8)

    blob = (rec("scidac-private-file-xml", ("<?xml version=\"1.0\" encoding=\"UTF-8\"?><scidacFile><version>1.1</version><spacetime>4</spacetime><dims>%d %d %d %d </dims><volfmt>0</volfmt></scidacFile>" % X).encode() + b"\0", True, False)
            + rec("scidac-file-xml", b"assembled by tests/test_mg_gpu.py\0", False, True)
            + rec("scidac-private-record-xml", ("<?xml version=\"1.0\" encoding=\"UTF-8\"?><scidacRecord><version>1.1</version><date>today</date><recordtype>0</recordtype><datatype>QUDA_FNs4Nc3_ColorSpinorField</datatype>"
                                               "<precision>F</precision><colors>3</colors><spins>4</spins><typesize>96</typesize><datacount>%d</datacount></scidacRecord>" % nvec).encode() + b"\0", True, False)
            + rec("scidac-record-xml", b"null vectors\0", False, False)
            + rec("scidac-binary-data", payload.tobytes(), False, False)
            + rec("scidac-checksum", ("<?xml version=\"1.0\" encoding=\"UTF-8\"?><scidacChecksum><version>1.0</version><suma>%x</suma><sumb>%x</sumb></scidacChecksum>" % (suma, sumb)).encode() + b"\0", False, True))
    base = str(tmp_path / "handmade")
    with open(base + "_level_0", "wb") as f:
        f.write(blob)
    mp = qa.multigrid_param(ip, n_level=2, geo_block=bs, n_vec=nvec)
    mp.compute_null_vector = qa.QUDA_COMPUTE_NULL_VECTOR_NO
    mp.vec_infile = base.encode()
    mg = qa.Multigrid(mp)
    try:
        for k in range(nvec):   # the device holds exactly the vectors of the file
            assert np.array_equal(mg.null_vector(0, k).reshape(V4, 4, 3), B[k]), k
        Bd = np.stack([b.astype(np.complex128) for b in B], axis=-1)
        Vo = oracle.mg_block_orthogonalize(Bd, list(X), list(bs), 4, 3, nvec, 2)
        Vc = V4 // int(np.prod(bs))
        fine = (rng.standard_normal((V4, 4, 3)) + 1j * rng.standard_normal((V4, 4, 3))).astype(np.complex64)
        coarse = (rng.standard_normal((Vc, 2, nvec)) + 1j * rng.standard_normal((Vc, 2, nvec))).astype(np.complex64)
        r_dev, p_dev = mg.apply(0, "R", fine), mg.apply(0, "P", coarse)
        r_ref = oracle.mg_restrict(fine.astype(np.complex128), Vo, list(X), list(bs), 4, 3, nvec, 2)
        p_ref = oracle.mg_prolongate(coarse.astype(np.complex128), Vo, list(X), list(bs), 4, 3, nvec, 2)
        assert float(np.max(np.abs(r_dev.reshape(r_ref.shape) - r_ref)) / np.max(np.abs(r_ref))) < 2e-5
        assert float(np.max(np.abs(p_dev.reshape(p_ref.shape) - p_ref)) / np.max(np.abs(p_ref))) < 2e-5
    finally:
        mg.free()
