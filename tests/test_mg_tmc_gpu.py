"""Multigrid on the twisted-CLOVER operator — the production ETMC action (reference drivers: qkxtm/CalcMG_2pt3pt_EvenOdd.cpp:222-240,
loadCloverQuda(NULL, NULL)) — through the C ABI.  SURVEY 8a row a13: DiracTwistedClover[PC]::createCoarseOp
(lib/dirac_twisted_clover.cpp:161-164, :423-426), the coarse clover V^dag A V + i mu' gamma5 (lib/coarse_op.cuh:813-884), and the
harness' own check of a solve (tests/multigrid_invert_test.cpp:177-198, :484, :529-577) with the HOST tmc_mat.

Pinned as the twisted-mass hierarchy is (tests/test_mg_gpu.py): Galerkin links Y and the local matrix X of BOTH coarsenings against
the oracle's restated calculateY fed with the device's own V (2e-5, fp32 device arithmetic), the three MG::verify() identities
< 1e-4, and every MG-GCR solution's residual recomputed on the host with the golden-pinned tmc_mat <= 1e-10."""
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from synth import smooth_gauge  # noqa: E402


@pytest.fixture(scope="module")
def qa():
    mod = importlib.import_module("quda-qkxtm-multigrid_amd")
    mod.init(0)
    yield mod
    mod.end()


CSW_COEFF = 0.124 * 1.57551      # clover_coeff = kappa * csw, csw as the reference harness' default (tests/test_util.cpp:1562-1618)


def _tmc_param(qa, kappa, mu, flavor, coeff):
    ip = qa.invert_param(qa.QUDA_TWISTED_CLOVER_DSLASH, kappa, mu, flavor, "ee", 0, cuda_prec=8, prec_sloppy=4, prec_precondition=4,
                         solution_type=qa.QUDA_MAT_SOLUTION)
    ip.solve_type = qa.QUDA_DIRECT_SOLVE
    ip.inv_type, ip.gcrNkrylov, ip.tol, ip.maxiter, ip.reliable_delta, ip.verbosity = qa.QUDA_GCR_INVERTER, 20, 1e-10, 2000, 1e-4, qa.QUDA_SILENT
    ip.clover_coeff = coeff
    return ip


def _setup(qa, oracle, X, kappa, mu, coeff=CSW_COEFF, device_clover=True, eps=0.35):
    """gauge + clover resident in fp64 / fp32 / fp32 (precise / sloppy / precondition).  device_clover: the clover term is built by the
    library from the resident links (what the QKXTM drivers do); otherwise the host field (oracle construction) is uploaded."""
    gauge = smooth_gauge(X, eps)
    gp = qa.gauge_param(X, cuda_prec=8, prec_sloppy=4, prec_precondition=4, t_boundary=qa.QUDA_PERIODIC_T)
    qa.load_gauge(gauge, gp)
    ip = _tmc_param(qa, kappa, mu, +1, coeff)
    clover = oracle.clover_compute(gauge, coeff, list(X))
    if device_clover:
        qa.load_clover(None, None, ip)
    else:
        qa.load_clover(clover, None, ip)
    return gauge, clover, ip


def _true_residual(oracle, gauge, clover, X, kappa, mu, flavor, x, b):
    oracle.set_threads(8)
    try:
        mx = oracle.tmc_mat(gauge, clover, x, list(X), kappa, mu, flavor, 0)
    finally:
        oracle.set_threads(1)
    return float(np.linalg.norm(b - mx) / np.linalg.norm(b))


def _rel(a, b):
    return float(np.max(np.abs(a - b)) / np.max(np.abs(b)))


def _use_mg(qa, ip, mg):
    ip.inv_type_precondition = qa.QUDA_MG_INVERTER
    ip.preconditioner = mg.h
    ip.tol_precondition, ip.maxiter_precondition, ip.precondition_cycle, ip.omega = 1e-1, 1, 1, 1.0


@pytest.mark.parametrize("mask,device_clover", [(0, True), (0, False), (15, True), (9, True)],
                         ids=["unpartitioned-device-clover", "unpartitioned-supplied-clover", "self-neighbour-xyzt", "self-neighbour-xt"])
def test_twisted_clover_hierarchy_against_oracle_restatement(qa, oracle, mask, device_clover):
    """3 levels on a twisted-clover fine operator.  Level 0 -> 1: Y, X against oracle.mg_coarse_op_fine(V, gauge, clover, ...)
    (coarse clover = V^dag A V, twisted-mass term +- i mu' on the chirality diagonals; lib/coarse_op.cuh:732-868); level 1 -> 2 against
    the from-coarse variant; R, P, the coarse apply; the fine operator of the hierarchy against tmc_mat; MG::verify(); then MG-GCR on
    the full system, residual by the host tmc_mat."""
    X, kappa, mu = (8, 8, 8, 16), 0.124, 0.005
    qa.lib().qudaAmdSetPartitionMask(mask)
    gauge, clover, ip = _setup(qa, oracle, X, kappa, mu, device_clover=device_clover)
    mp = qa.multigrid_param(ip, n_level=3, geo_block=[(4, 4, 4, 4), (1, 1, 1, 2), (2, 2, 2, 2)], n_vec=8, setup_maxiter=100, setup_tol=1e-4)
    mg = qa.Multigrid(mp)
    rng = np.random.default_rng(17)
    try:
        assert mg.levels() == 3
        Yprev = Xprev = None
        for level in range(2):
            i = mg.level_info(level)
            Xf, Xc, bs, Ns, Nc, Nv, sbs = i["Xf"], i["Xc"], i["geo_bs"], i["fineSpin"], i["fineColor"], i["Nvec"], i["spin_bs"]
            B = np.stack([mg.null_vector(level, k) for k in range(Nv)], axis=-1)
            Vd = mg.V(level).astype(np.complex128)
            Vo = oracle.mg_block_orthogonalize(B, Xf, bs, Ns, Nc, Nv, sbs)
            assert _rel(Vd, Vo) < 2e-4, (level, _rel(Vd, Vo))
            phi = (rng.standard_normal((int(np.prod(Xf)), Ns, Nc)) + 1j * rng.standard_normal((int(np.prod(Xf)), Ns, Nc)))
            eta = (rng.standard_normal((int(np.prod(Xc)), 2, Nv)) + 1j * rng.standard_normal((int(np.prod(Xc)), 2, Nv)))
            assert _rel(mg.apply(level, "R", phi), oracle.mg_restrict(phi, Vd, Xf, bs, Ns, Nc, Nv, sbs)) < 2e-5
            assert _rel(mg.apply(level, "P", eta), oracle.mg_prolongate(eta, Vd, Xf, bs, Ns, Nc, Nv, sbs)) < 2e-5
            Yd, Xd = mg.coarse_links(level)
            if level == 0:
                Yo, Xo = oracle.mg_coarse_op_fine(Vd, gauge, clover, kappa, 2 * kappa * mu, Xf, bs, Nv)
                # the clover term really is in there: the same construction without it differs at the percent level
                _, Xnc = oracle.mg_coarse_op_fine(Vd, gauge, None, kappa, 2 * kappa * mu, Xf, bs, Nv)
                assert _rel(Xnc, Xo) > 1e-3
            else:
                Yo, Xo = oracle.mg_coarse_op_coarse(Vd, Yprev, Xprev, kappa, Xf, bs, Nc, Nv)
            assert _rel(Xd, Xo) < 2e-5, (level, _rel(Xd, Xo))
            assert _rel(Yd, -kappa * Yo) < 2e-5, (level, _rel(Yd, -kappa * Yo))
            Yref = Yd.astype(np.complex128) / (-kappa)
            want = oracle.mg_coarse_apply(eta, Yref, Xd.astype(np.complex128), kappa, Xc, Nv)
            assert _rel(mg.apply(level + 1, "M", eta), want) < 2e-5
            Yprev, Xprev = Yref, Xd.astype(np.complex128)
        # the level-0 operator of the hierarchy is the oracle's tmc_mat
        phi = rng.standard_normal((int(np.prod(X)), 4, 3)) + 1j * rng.standard_normal((int(np.prod(X)), 4, 3))
        want = oracle.tmc_mat(gauge, clover, np.ascontiguousarray(phi).view(np.float64).reshape(-1), list(X), kappa, mu, +1, 0).view(np.complex128).reshape(-1, 4, 3)
        assert _rel(mg.apply(0, "M", phi), want) < 2e-5
        dev = mg.verify()
        assert max(dev) < 1e-4, dev
        b = rng.random(int(np.prod(X)) * 24)
        ip.inv_type_precondition = qa.QUDA_INVALID_ENUM
        qa.invert(b, ip)
        plain = ip.iter
        _use_mg(qa, ip, mg)
        x = qa.invert(b, ip)
        res = _true_residual(oracle, gauge, clover, X, kappa, mu, +1, x, b)
        print("twisted-clover MG-GCR mask %d device clover %s: %d iterations (plain GCR %d), host residual %.2e" % (mask, device_clover, ip.iter, plain, res))
        assert res < 1e-10, res
        assert ip.iter < 40 and ip.iter < plain, (ip.iter, plain)
    finally:
        mg.free()
        qa.lib().qudaAmdSetPartitionMask(0)


@pytest.mark.parametrize("smoother_pc", [False, True], ids=["full-smoother", "pc-smoother"])
@pytest.mark.parametrize("X,levels,blocks,nvec", [((8, 8, 8, 8), 2, (4, 4, 4, 4), 8), ((16, 8, 8, 16), 3, [(4, 4, 4, 4), (2, 2, 2, 2), (2, 2, 2, 2)], 8)])
def test_twisted_clover_verify_identities_and_mg_gcr(qa, oracle, X, levels, blocks, nvec, smoother_pc):
    """the reference harness' flow for --dslash-type twisted-clover (tests/multigrid_invert_test.cpp:177-198): setup, run_verify, solve,
    host check — with the full-operator and with the even-odd preconditioned (DiracTwistedCloverPC) smoother"""
    kappa, mu = 0.124, 0.005
    gauge, clover, ip = _setup(qa, oracle, X, kappa, mu)
    b = np.random.default_rng(5).random(int(np.prod(X)) * 24)
    ip.inv_type_precondition = qa.QUDA_INVALID_ENUM
    x0 = qa.invert(b, ip)
    plain = ip.iter
    assert _true_residual(oracle, gauge, clover, X, kappa, mu, +1, x0, b) < 1e-10
    mp = qa.multigrid_param(ip, n_level=levels, geo_block=blocks, n_vec=nvec, setup_maxiter=300, setup_tol=1e-5, smoother_pc=smoother_pc)
    mg = qa.Multigrid(mp)
    try:
        dev = mg.verify()
        assert max(dev) < 1e-4, dev
        _use_mg(qa, ip, mg)
        x = qa.invert(b, ip)
        res = _true_residual(oracle, gauge, clover, X, kappa, mu, +1, x, b)
        assert res < 1e-10, res
        assert abs(ip.true_res - res) < 1e-9
        assert ip.iter < plain, (ip.iter, plain)
        print("twisted-clover MG-GCR %s pc=%s: %d iterations (plain GCR %d), host residual %.2e, setup %.2f s, solve %.3f s" % (X, smoother_pc, ip.iter, plain, res, mp.secs, ip.secs))
    finally:
        mg.free()


def test_twisted_clover_outer_even_odd_solve_with_up_and_down_hierarchies(qa, oracle):
    """the QKXTM production shape on the production action: outer even-odd GCR (solve_type = QUDA_DIRECT_PC_SOLVE), one hierarchy per
    twist flavour in preconditionerUP / preconditionerDN (reference lib/interface_quda.cpp:6041, :6389-6520), single-parity injection;
    then the same hierarchies under a full-system outer solve"""
    X, kappa, mu = (8, 8, 8, 16), 0.124, 0.005
    gauge, clover, ip = _setup(qa, oracle, X, kappa, mu)
    b = np.random.default_rng(23).random(int(np.prod(X)) * 24)
    ip.solve_type = qa.QUDA_DIRECT_PC_SOLVE
    hier = {}
    try:
        for flavor in (+1, -1):
            ipm = _tmc_param(qa, kappa, mu, flavor, CSW_COEFF)
            mp = qa.multigrid_param(ipm, n_level=3, geo_block=[(4, 4, 4, 4), (1, 1, 1, 2), (2, 2, 2, 2)], n_vec=8, setup_maxiter=300, setup_tol=1e-5, smoother_pc=True,
                                    coarse_matpc=True)
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
            res = _true_residual(oracle, gauge, clover, X, kappa, mu, flavor, x, b)
            print("twisted-clover outer even-odd MG-GCR flavour %+d: %d iterations (plain even-odd GCR %d), host residual %.2e" % (flavor, ip.iter, plain, res))
            assert res < 1e-10, (flavor, res)
            assert ip.iter * 3 < plain, (flavor, ip.iter, plain)
        ip.solve_type = qa.QUDA_DIRECT_SOLVE
        ip.twist_flavor = qa.QUDA_TWIST_PLUS
        ip.preconditioner = ip.preconditionerUP
        x = qa.invert(b, ip)
        res = _true_residual(oracle, gauge, clover, X, kappa, mu, +1, x, b)
        assert res < 1e-10 and ip.iter < 30, (res, ip.iter)
    finally:
        for h, _, _ in hier.values():
            h.free()


@pytest.mark.parametrize("dslash", ["tm", "tmc"])
def test_matdagmat_through_the_c_abi(qa, oracle, dslash):
    """MatDagMatQuda (include/quda.h:747) against Mdag(M x) of the oracle: full operator (QUDA_MAT_SOLUTION) and the even-odd
    preconditioned one (QUDA_MATPC_SOLUTION), twisted mass and twisted clover, fp64 1e-12"""
    X, kappa, mu = (8, 4, 6, 8), 0.12, 0.3
    gauge, spinor, clover = oracle.make_fields(list(X), seed=41)
    nh = spinor.size // 2
    qa.load_gauge(gauge, qa.gauge_param(X, cuda_prec=8))
    cinv = oracle.clover_twisted_inverse(clover, 4 * kappa * kappa * mu * mu)
    for flavor in (+1, -1):
        ty = qa.QUDA_TWISTED_MASS_DSLASH if dslash == "tm" else qa.QUDA_TWISTED_CLOVER_DSLASH
        ip = qa.invert_param(ty, kappa, mu, flavor, "ee", 0, cuda_prec=8, solution_type=qa.QUDA_MAT_SOLUTION)
        if dslash == "tmc":
            qa.load_clover(clover, None, ip)
        got = qa.matdagmat(spinor.copy(), ip)
        if dslash == "tm":
            want = oracle.tm_mat(gauge, oracle.tm_mat(gauge, spinor.copy(), list(X), kappa, mu, flavor, 0), list(X), kappa, mu, flavor, 1)
        else:
            want = oracle.tmc_mat(gauge, clover, oracle.tmc_mat(gauge, clover, spinor.copy(), list(X), kappa, mu, flavor, 0), list(X), kappa, mu, flavor, 1)
        assert _rel(got, want) < 1e-12, (dslash, flavor, "full")
        for matpc in ("ee", "oo", "eeasym", "ooasym"):
            ip = qa.invert_param(ty, kappa, mu, flavor, matpc, 0, cuda_prec=8, solution_type=qa.QUDA_MATPC_SOLUTION)
            got = qa.matdagmat(spinor[:nh].copy(), ip)
            if dslash == "tm":
                want = oracle.tm_matpc(gauge, oracle.tm_matpc(gauge, spinor[:nh].copy(), list(X), kappa, mu, flavor, matpc, 0), list(X), kappa, mu, flavor, matpc, 1)
            else:
                want = oracle.tmc_matpc(gauge, oracle.tmc_matpc(gauge, spinor[:nh].copy(), clover, cinv, list(X), kappa, mu, flavor, matpc, 0), clover, cinv, list(X), kappa, mu, flavor, matpc, 1)
            assert _rel(got, want) < 1e-12, (dslash, flavor, matpc)


@pytest.mark.parametrize("mask", [0, 15, 6], ids=["unpartitioned", "self-neighbour-xyzt", "self-neighbour-yz"])
@pytest.mark.parametrize("dslash", ["tm", "tmc"])
def test_multi_rhs_fine_stencil_against_the_oracle(qa, oracle, dslash, mask):
    """The 8/16/24/32-right-hand-side stencil of the lockstep null-vector solves (csrc/dslash.hip fine_block_kernel; twisted clover:
    the dense A + i a g5 site matrices of cloverTwistDense in its epilogue) applied to a batch, every right-hand side against the
    host tm_mat / tmc_mat (fp32 device arithmetic: 2e-5 of the largest element); and the hierarchy's own record that the level-0 null
    vectors came from the lockstep solve on that stencil (null_method 1), the coarse ones from the MFMA operator (2).
    mask != 0: grid-decomposed lattice with the process as its own neighbour — the hops across a partitioned face read the neighbour's
    panel from the ghost zone behind the block field (block.h BlockGhost), filled by one pack launch + one grouped exchange."""
    X, kappa, mu = (8, 8, 8, 16), 0.124, 0.005
    qa.lib().qudaAmdSetPartitionMask(mask)
    if dslash == "tmc":
        gauge, clover, ip = _setup(qa, oracle, X, kappa, mu)
    else:
        gauge, clover, ip = _setup(qa, oracle, X, kappa, mu)
        ip = qa.invert_param(qa.QUDA_TWISTED_MASS_DSLASH, kappa, mu, +1, "ee", 0, cuda_prec=8, prec_sloppy=4, prec_precondition=4, solution_type=qa.QUDA_MAT_SOLUTION)
        ip.solve_type, ip.inv_type, ip.verbosity = qa.QUDA_DIRECT_SOLVE, qa.QUDA_GCR_INVERTER, qa.QUDA_SILENT
    mp = qa.multigrid_param(ip, n_level=3, geo_block=[(4, 4, 4, 4), (1, 1, 1, 2), (2, 2, 2, 2)], n_vec=8, setup_maxiter=100, setup_tol=1e-4)
    mg = qa.Multigrid(mp)
    rng = np.random.default_rng(31)
    V = int(np.prod(X))
    try:
        assert mg.level_info(0)["null_method"] == 1 and mg.level_info(0)["null_iters"] > 0
        assert mg.level_info(1)["null_method"] == 2
        for nrhs in (8, 16, 24, 32):
            phi = (rng.standard_normal((nrhs, V, 4, 3)) + 1j * rng.standard_normal((nrhs, V, 4, 3))).astype(np.complex64)
            phi *= (10.0 ** rng.integers(-2, 3, size=nrhs)).astype(np.float32)[:, None, None, None]
            got, _ = mg.apply_block(0, phi)
            oracle.set_threads(8)
            try:
                for k in range(nrhs):
                    v = np.ascontiguousarray(phi[k].astype(np.complex128)).view(np.float64).reshape(-1)
                    if dslash == "tmc":
                        want = oracle.tmc_mat(gauge, clover, v, list(X), kappa, mu, +1, 0)
                    else:
                        want = oracle.tm_mat(gauge, v, list(X), kappa, mu, +1, 0)
                    assert _rel(got[k], want.view(np.complex128).reshape(-1, 4, 3)) < 2e-5, (nrhs, k)
            finally:
                oracle.set_threads(1)
        if mask:
            assert qa.comm_stats()["block_exchanges"] > 0
    finally:
        mg.free()
        qa.lib().qudaAmdSetPartitionMask(0)


def test_dense_clover_twist_inverse_in_the_lockstep_solve(qa, oracle):
    """The even-odd preconditioned lockstep solve uses (A + i a g5)^-1 as a dense site matrix (Gauss-Jordan on the device) and
    reconstructs x_o = kappa (A + i a g5)^-1 D_oe x_e (reference DiracTwistedCloverPC::reconstruct with b = 0,
    lib/dirac_twisted_clover.cpp:400-421).  Sharp consequence, checked with the HOST operator: the odd half of tmc_mat v vanishes to
    fp32 round-off for every null vector v — (M v)_o = (A + i a g5) x_o - kappa D_oe x_e — which it only does if the dense inverse
    really is the inverse of the host's clover-twist term; and the vectors are rich in low modes (|M v| / |v| well below a random
    vector's)."""
    X, kappa, mu = (8, 8, 8, 16), 0.124, 0.005
    gauge, clover, ip = _setup(qa, oracle, X, kappa, mu)
    mp = qa.multigrid_param(ip, n_level=2, geo_block=(4, 4, 4, 4), n_vec=8, setup_maxiter=300, setup_tol=1e-5)
    mg = qa.Multigrid(mp)
    try:
        assert mg.level_info(0)["null_method"] == 1
        rng = np.random.default_rng(1)
        r = rng.standard_normal(int(np.prod(X)) * 24)
        scale = np.linalg.norm(oracle.tmc_mat(gauge, clover, r, list(X), kappa, mu, +1, 0)) / np.linalg.norm(r)
        worst, worst_odd = 0.0, 0.0
        nh = r.size // 2
        for k in range(8):
            v = np.ascontiguousarray(mg.null_vector(0, k).astype(np.complex128)).view(np.float64).reshape(-1)
            mv = oracle.tmc_mat(gauge, clover, v, list(X), kappa, mu, +1, 0)
            worst = max(worst, np.linalg.norm(mv) / np.linalg.norm(v))
            worst_odd = max(worst_odd, np.linalg.norm(mv[nh:]) / np.linalg.norm(v))
        print("twisted-clover null vectors: worst |M v| / |v| = %.3e (random vector %.3e), odd half %.3e" % (worst, scale, worst_odd))
        assert worst_odd < 2e-6, worst_odd
        assert worst < 0.5 * scale, (worst, scale)
    finally:
        mg.free()


def test_twisted_clover_half_precision_cycle(qa, oracle):
    """fp16 mirrors of V and the coarse links AND a 16-bit level-0 smoother on the twisted-clover operator (16-bit copies of the links and of
    the clover term, its twisted inverse recomputed on the device; round 3): the outer fp64 GCR still reaches 1e-10 with the residual
    recomputed on the host by tmc_mat, in at most two iterations more, and switching back restores the fp32 solve bit for bit."""
    X, kappa, mu = (16, 8, 8, 16), 0.124, 0.005
    gauge, clover, ip = _setup(qa, oracle, X, kappa, mu)
    b = np.random.default_rng(29).random(int(np.prod(X)) * 24)
    mp = qa.multigrid_param(ip, n_level=3, geo_block=[(4, 4, 4, 4), (2, 2, 2, 2), (2, 2, 2, 2)], n_vec=8, setup_maxiter=300, setup_tol=1e-5, smoother_pc=True)
    mg = qa.Multigrid(mp)
    try:
        _use_mg(qa, ip, mg)
        x32 = qa.invert(b, ip)
        it32 = ip.iter
        mg.set_half_storage(True)
        x16 = qa.invert(b, ip)
        it16 = ip.iter
        assert _true_residual(oracle, gauge, clover, X, kappa, mu, +1, x16, b) < 1e-10
        assert it16 <= it32 + 2, (it16, it32)
        mg.set_half_storage(False)
        x32b = qa.invert(b, ip)
        assert ip.iter == it32 and np.array_equal(x32b, x32)
    finally:
        mg.set_half_storage(False)
        mg.free()
