"""GPU parity tests proper: the HIP path, called through the C ABI (dslashQuda / MatQuda / cloverQuda /
loadGaugeQuda / loadCloverQuda and the resident-field extension), against (1) the golden vectors produced by
the reference's own host operators and (2) the oracle on larger seeded lattices, in every storage precision.

Tolerances (BASELINE.json north_star: 1e-5 relative in double, 1e-2 in 16-bit, per site):
    fp64  1e-12   (observed ~1e-15; the bar 1e-5 is met with 7 orders to spare)
    fp32  2e-5
    16-bit 1e-2
"""
import ctypes as C
import importlib
import json
import os

import numpy as np
import pytest

import qa_cases as qc

pytestmark = pytest.mark.gpu

TOL = {8: 1e-12, 4: 2e-5, 2: 1e-2}


@pytest.fixture(scope="module")
def qa():
    mod = importlib.import_module("quda-qkxtm-multigrid_amd")
    mod.init(0)
    yield mod
    mod.end()


def _load_fields(qa, gauge, clover, X, kappa, mu, prec, recon, host_dtype=np.float64):
    cpu_prec = qa.QUDA_DOUBLE_PRECISION if host_dtype == np.float64 else qa.QUDA_SINGLE_PRECISION
    gp = qa.gauge_param(X, cpu_prec=cpu_prec, cuda_prec=prec, recon=recon)
    qa.load_gauge(gauge.astype(host_dtype), gp)
    assert gp.gaugeGiB > 0
    if clover is not None:
        ip = qa.invert_param(qa.QUDA_TWISTED_CLOVER_DSLASH, kappa, mu, cpu_prec=cpu_prec, cuda_prec=prec)
        qa.load_clover(clover.astype(host_dtype), None, ip)  # inverse (A^2 + mu2)^-1 computed on the device


@pytest.mark.parametrize("path", qc.FILES, ids=[os.path.basename(f) for f in qc.FILES])
@pytest.mark.parametrize("prec,recon", [(8, 18), (8, 12), (8, 8), (4, 18), (4, 12), (4, 8), (2, 18), (2, 12), (2, 8)])
def test_all_golden_cases_through_c_abi(qa, path, prec, recon):
    z, X, kappa, mu, gauge = qc.load(path)
    _load_fields(qa, gauge, z["clover"], X, kappa, mu, prec, recon)
    worst = {}
    for name in qc.case_names(z):
        got = qc.run_abi(qa, name, z["spinor"], X, kappa, mu, prec)
        err = qc.rel_err(got, z[name])
        worst[name] = err
        assert err < TOL[prec], "%s prec=%d recon=%d: %g" % (name, prec, recon, err)
    assert len(worst) == 66


def test_single_precision_host_fields(qa):
    z, X, kappa, mu, gauge = qc.load(qc.FILES[0])
    _load_fields(qa, gauge, z["clover"], X, kappa, mu, 4, 18, host_dtype=np.float32)
    for name in ("tm_dslash_fp_ee_d0_p0", "tmc_matpc_fp_ee_d0", "tm_mat_fm_d1"):
        got = qc.run_abi(qa, name, z["spinor"], X, kappa, mu, 4, host_dtype=np.float32)
        assert qc.rel_err(got, z[name]) < 3e-5


def test_supplied_clover_inverse_and_return(qa):
    """loadCloverQuda with the caller's inverse field, and return_clover_inverse handing back the device-computed one."""
    z, X, kappa, mu, gauge = qc.load(qc.FILES[1])
    gp = qa.gauge_param(X)
    qa.load_gauge(gauge, gp)
    ip = qa.invert_param(qa.QUDA_TWISTED_CLOVER_DSLASH, kappa, mu)
    qa.load_clover(z["clover"], z["clover_inv"].copy(), ip)
    got = qc.run_abi(qa, "tmc_dslash_fp_ee_d0_p0", z["spinor"], X, kappa, mu, 8)
    assert qc.rel_err(got, z["tmc_dslash_fp_ee_d0_p0"]) < 1e-12
    back = np.zeros_like(z["clover_inv"])
    ip.return_clover_inverse = 1
    ip.compute_clover_inverse = 1
    qa.load_clover(z["clover"], back, ip)
    assert qc.rel_err(back, z["clover_inv"]) < 1e-12


def test_ukqcd_host_basis_and_qdp_dirac_order(qa, oracle):
    """gamma_basis = UKQCD and dirac_order = QDP (spin inside colour) host fields go through the same rotation
    the reference applies (lib/copy_color_spinor.cuh:49-91)."""
    z, X, kappa, mu, gauge = qc.load(qc.FILES[0])
    gp = qa.gauge_param(X)
    qa.load_gauge(gauge, gp)
    nh = z["spinor"].size // 2
    src_dr = z["spinor"][:nh].reshape(-1, 4, 3, 2)
    k = 1 / np.sqrt(2.0)
    S = k * np.array([[0, 1, 0, 1], [-1, 0, -1, 0], [0, 1, 0, -1], [-1, 0, 1, 0]], dtype=np.float64)  # DR -> UKQCD
    src_uk = np.einsum("st,xtcz->xscz", S, src_dr)
    ip = qa.invert_param(qa.QUDA_TWISTED_MASS_DSLASH, kappa, mu, +1, "ee", 0, gamma_basis=qa.QUDA_UKQCD_GAMMA_BASIS)
    got_uk = qa.dslash(np.ascontiguousarray(src_uk).ravel(), ip, 0).reshape(-1, 4, 3, 2)
    want_uk = np.einsum("st,xtcz->xscz", S, z["tm_dslash_fp_ee_d0_p0"].reshape(-1, 4, 3, 2))
    assert qc.rel_err(got_uk, want_uk) < 1e-12
    ip2 = qa.invert_param(qa.QUDA_TWISTED_MASS_DSLASH, kappa, mu, +1, "ee", 0, dirac_order=qa.QUDA_QDP_DIRAC_ORDER)
    src_cs = np.ascontiguousarray(src_dr.transpose(0, 2, 1, 3)).ravel()
    got_cs = qa.dslash(src_cs, ip2, 0).reshape(-1, 3, 4, 2).transpose(0, 2, 1, 3)
    assert qc.rel_err(got_cs, z["tm_dslash_fp_ee_d0_p0"].reshape(-1, 4, 3, 2)) < 1e-12


@pytest.mark.parametrize("X", [(8, 8, 8, 8), (16, 16, 16, 16), (12, 6, 10, 4), (2, 2, 2, 2), (4, 2, 2, 8)])
def test_oracle_parity_on_seeded_lattices(qa, oracle, X):
    """Same seeded inputs (glibc rand(), as the reference harness) on the HIP path and on the oracle; includes the
    smallest legal lattice 2^4 (every neighbour wraps) and non-cubic ones.  For 8^4/16^4 the oracle result is
    additionally tied to the reference's own ||out||^2 (tests/golden/ref_checksums.json)."""
    gauge, spinor, clover = oracle.make_fields(list(X))
    nh = spinor.size // 2
    sums = {tuple(s["X"]): s for s in json.load(open(os.path.join(qc.GOLD, "ref_checksums.json")))}
    kappa, mu = 0.1, 0.01
    oracle.set_threads(8)
    try:
        want_tm = oracle.tm_dslash(gauge, spinor[:nh].copy(), list(X), kappa, mu, +1, 0, "ee", 0)
        cinv = oracle.clover_twisted_inverse(clover, 4 * kappa * kappa * mu * mu)
        want_tmc = oracle.tmc_dslash(gauge, spinor[:nh].copy(), clover, cinv, list(X), kappa, mu, +1, 0, "ee", 0)
        want_mpc = oracle.tm_matpc(gauge, spinor[nh:].copy(), list(X), kappa, mu, -1, "oo", 1)
    finally:
        oracle.set_threads(1)
    if tuple(X) in sums:
        assert oracle.norm2(want_tm) == sums[tuple(X)]["tm_dslash_fp_ee_d0_p0"]
        assert oracle.norm2(want_tmc) == sums[tuple(X)]["tmc_dslash_fp_ee_d0_p0"]
    for prec in (8, 4, 2):
        _load_fields(qa, gauge, clover, X, kappa, mu, prec, 18)
        ip = qa.invert_param(qa.QUDA_TWISTED_MASS_DSLASH, kappa, mu, +1, "ee", 0, cuda_prec=prec)
        assert qc.rel_err(qa.dslash(spinor[:nh].copy(), ip, 0), want_tm) < TOL[prec]
        ipc = qa.invert_param(qa.QUDA_TWISTED_CLOVER_DSLASH, kappa, mu, +1, "ee", 0, cuda_prec=prec)
        assert qc.rel_err(qa.dslash(spinor[:nh].copy(), ipc, 0), want_tmc) < TOL[prec]
        ipm = qa.invert_param(qa.QUDA_TWISTED_MASS_DSLASH, kappa, mu, -1, "oo", 1, cuda_prec=prec)
        assert qc.rel_err(qa.mat(spinor[nh:].copy(), ipm), want_mpc) < 2 * TOL[prec]


def test_resident_operator_api_and_properties_at_full_size(qa, oracle):
    """BASELINE size (32^4) through the resident-field API used by bench.py, checked with size-independent properties:
    linearity, gamma5-hermiticity  <y, M x> = <M^dag y, x>,  MdagM = Mdag(M), and a 2-site spot check against the oracle."""
    X = (32, 32, 32, 32)
    V = int(np.prod(X))
    rng = np.random.default_rng(7)
    # random SU(3) links by QR (the oracle's rand() generator would take a minute at this size)
    g = rng.standard_normal((4, V, 3, 3)) + 1j * rng.standard_normal((4, V, 3, 3))
    q, r = np.linalg.qr(g)
    q = q * (np.diagonal(r, axis1=-2, axis2=-1) / np.abs(np.diagonal(r, axis1=-2, axis2=-1)))[..., None, :]
    q = q / np.linalg.det(q)[..., None, None] ** (1.0 / 3.0)
    gauge = np.ascontiguousarray(np.stack([q.real, q.imag], axis=-1)).reshape(4, V * 18)
    kappa, mu = 0.1, 0.01
    gp = qa.gauge_param(X, cuda_prec=8)
    qa.load_gauge(gauge, gp)
    ip = qa.invert_param(qa.QUDA_TWISTED_MASS_DSLASH, kappa, mu, +1, "ee", 0)
    nh = V // 2 * 24
    x_h, y_h = rng.random(nh), rng.random(nh)
    x, y, mx, my, t = [qa.Spinor(8) for _ in range(5)]
    x.load(x_h, ip)
    y.load(y_h, ip)
    d = qa.Dirac(ip, pc=True)
    d.M(mx, x)
    d.Mdag(my, y)
    import ctypes as C
    r1, r2 = (C.c_double * 2)(), (C.c_double * 2)()
    qa.lib().qudaAmdBlasCDot(y.h, mx.h, r1)
    qa.lib().qudaAmdBlasCDot(my.h, x.h, r2)
    scale = np.sqrt(y.norm2() * mx.norm2())
    assert abs(r1[0] - r2[0]) / scale < 1e-12 and abs(r1[1] - r2[1]) / scale < 1e-12
    # linearity: M(x + 2y) = M x + 2 M y
    qa.lib().qudaAmdBlasAxpy(2.0, y.h, x.h)
    d.M(t, x)
    d.M(my, y)
    qa.lib().qudaAmdBlasAxpy(2.0, my.h, mx.h)
    qa.lib().qudaAmdBlasAxpy(-1.0, t.h, mx.h)
    assert mx.norm2() / t.norm2() < 1e-24
    # MdagM = Mdag M
    d.MdagM(mx, y)
    d.M(t, y)
    d.Mdag(my, t)
    qa.lib().qudaAmdBlasAxpy(-1.0, my.h, mx.h)
    assert mx.norm2() / my.norm2() < 1e-24
    # spot check one dslash against the oracle on a few output sites (oracle evaluated only there via a full call on parity 0
    # would take ~0.5 s at 32^4 with 8 threads)
    oracle.set_threads(8)
    try:
        want = oracle.tm_dslash(gauge, x_h.copy(), list(X), kappa, mu, +1, 0, "ee", 0)
    finally:
        oracle.set_threads(1)
    got = qa.dslash(x_h, ip, 0)
    assert qc.rel_err(got, want) < 1e-12
    for f in (x, y, mx, my, t):
        f.free()
    d.free()


@pytest.mark.parametrize("matpc", ["ee", "oo", "eeasym", "ooasym"])
def test_prepare_and_reconstruct_element_wise(qa, oracle, matpc):
    """Dirac::prepare / reconstruct of the even-odd preconditioned twisted-mass operator (reference
    lib/dirac_twisted_mass.cpp:526-586) against the oracle's building blocks on every site: with M = A - kappa D,
    src = [A^-1] (b_p + kappa D A^-1 b_q) and x_q = A^-1 (b_q + kappa D x_p)."""
    X = (8, 8, 8, 8)
    gauge, spinor, _ = oracle.make_fields(list(X), clover=False)
    nh = spinor.size // 2
    kappa, mu = 0.1, 0.01
    rng = np.random.default_rng(3)
    b_h, x_h = spinor.copy(), rng.random(spinor.size)
    Ainv = lambda v: oracle.twist_gamma5(v, kappa, mu, +1, 0, 1)
    A = lambda v: oracle.twist_gamma5(v, kappa, mu, +1, 0, 0)
    D = lambda v, parity: oracle.wil_dslash(gauge, v, list(X), parity, 0)
    # conventions of the building blocks, pinned on the full operator: M x = A x - kappa D x
    full = np.concatenate([A(x_h[:nh]) - kappa * D(x_h[nh:], 0), A(x_h[nh:]) - kappa * D(x_h[:nh], 1)])
    assert qc.rel_err(full, oracle.tm_mat(gauge, x_h.copy(), list(X), kappa, mu, +1, 0)) < 1e-13
    _load_fields(qa, gauge, None, X, kappa, mu, 8, 18)
    ip = qa.invert_param(qa.QUDA_TWISTED_MASS_DSLASH, kappa, mu, +1, matpc, 0, cuda_prec=8, solution_type=qa.QUDA_MAT_SOLUTION)
    ip.solve_type = qa.QUDA_DIRECT_PC_SOLVE
    d = qa.Dirac(ip, pc=True)
    x, b, src = qa.Spinor(8, qa.QUDA_FULL_SITE_SUBSET), qa.Spinor(8, qa.QUDA_FULL_SITE_SUBSET), qa.Spinor(8)
    x.load(x_h, ip)
    b.load(b_h, ip)
    p = 0 if matpc.startswith("ee") else 1            # parity the system is solved on
    half = lambda v, par: v[:nh] if par == 0 else v[nh:]
    want = half(b_h, p) + kappa * D(Ainv(half(b_h, 1 - p)), p)
    if not matpc.endswith("asym"):
        want = Ainv(want)
    d.prepare(src, x, b, qa.QUDA_MAT_SOLUTION)
    assert qc.rel_err(src.save(ip, b_h[:nh]), want) < 1e-13
    # reconstruct: the solved half of x stays, the other one is A^-1 (b_q + kappa D x_p)
    x.load(x_h, ip)
    b.load(b_h, ip)
    d.reconstruct(x, b, qa.QUDA_MAT_SOLUTION)
    got = x.save(ip, x_h)
    assert np.array_equal(half(got, p), half(x_h, p))
    assert qc.rel_err(half(got, 1 - p), Ainv(half(b_h, 1 - p) + kappa * D(half(x_h, p), 1 - p))) < 1e-13
    for f in (x, b, src):
        f.free()
    d.free()


@pytest.mark.parametrize("X", [(16, 16, 8, 24), (24, 16, 12, 8), (32, 8, 16, 16), (48, 12, 16, 8)])
def test_block_orders_on_odd_shapes(qa, oracle, X):
    """Every block order of the stencil is a re-numbering of the work-groups and must not change a single site: plane-tiled XCD
    order, y groups (explicit counts and the automatic choice), the boundary-first order of partitioned launches and the legacy slab
    order, on lattices whose planes / slabs divide differently (non-cubic, 3 x 2^k extents), against the oracle."""
    gauge, spinor, _ = oracle.make_fields(list(X), clover=False)
    nh = spinor.size // 2
    kappa, mu = 0.1, 0.01
    oracle.set_threads(8)
    try:
        want = oracle.tm_dslash(gauge, spinor[:nh].copy(), list(X), kappa, mu, +1, 0, "ee", 0)
    finally:
        oracle.set_threads(1)
    L = qa.lib()
    try:
        for prec in (8, 2):
            _load_fields(qa, gauge, None, X, kappa, mu, prec, 18)
            ip = qa.invert_param(qa.QUDA_TWISTED_MASS_DSLASH, kappa, mu, +1, "ee", 0, cuda_prec=prec)
            for setting in ({}, {"ygroups": 2}, {"ygroups": 3}, {"ygroups": 0}, {"tiled": 0}, {"tiled": 1, "tt": 2}, {"nxz": 4}, {"nxz": 2}, {"block": 64}, {"block": 128, "ygroups": 2}):
                for k, v in dict(block=0, tiled=-1, nxz=0, tz=0, tt=0, ygroups=-1).items():
                    L.qudaAmdSetDslashTune(k.encode(), v)
                for k, v in setting.items():
                    L.qudaAmdSetDslashTune(k.encode(), v)
                assert qc.rel_err(qa.dslash(spinor[:nh].copy(), ip, 0), want) < TOL[prec], (prec, setting)
                for mask in (14, 9):   # boundary-first order of the fused peer-store launch, self-neighbour emulation
                    L.qudaAmdSetPartitionMask(mask)
                    _load_fields(qa, gauge, None, X, kappa, mu, prec, 18)
                    assert qc.rel_err(qa.dslash(spinor[:nh].copy(), ip, 0), want) < TOL[prec], (prec, setting, mask)
                    L.qudaAmdSetPartitionMask(0)
                _load_fields(qa, gauge, None, X, kappa, mu, prec, 18)
    finally:
        L.qudaAmdSetPartitionMask(0)
        for k, v in dict(block=0, tiled=-1, nxz=0, tz=0, tt=0, ygroups=-1).items():
            L.qudaAmdSetDslashTune(k.encode(), v)


def test_twisted_clover_at_full_size_against_the_oracle(qa, oracle):
    """BASELINE configs[2]: 32^4 twisted-clover Dslash in fp32 and 16-bit (the kernels bench.py times as extra.tmc_*), every site
    against the oracle on the same inputs — plain and xpay form (the two epilogues of DiracTwistedCloverPC::M), fp64 as well."""
    X = (32, 32, 32, 32)
    V = int(np.prod(X))
    rng = np.random.default_rng(17)
    g = rng.standard_normal((4, V, 3, 3)) + 1j * rng.standard_normal((4, V, 3, 3))
    q, r = np.linalg.qr(g)
    q = q * (np.diagonal(r, axis1=-2, axis2=-1) / np.abs(np.diagonal(r, axis1=-2, axis2=-1)))[..., None, :]
    q = q / np.linalg.det(q)[..., None, None] ** (1.0 / 3.0)
    gauge = np.ascontiguousarray(np.stack([q.real, q.imag], axis=-1)).reshape(4, V * 18)
    del g, q, r
    from synth import make_clover
    clover = make_clover(list(X), seed=5)
    kappa, mu = 0.1, 0.01
    nh = V // 2 * 24
    x_h = rng.random(nh)
    oracle.set_threads(8)
    try:
        cinv = oracle.clover_twisted_inverse(clover, 4 * kappa * kappa * mu * mu)
        want = oracle.tmc_dslash(gauge, x_h.copy(), clover, cinv, list(X), kappa, mu, +1, 0, "ee", 0)
        want_m = oracle.tmc_matpc(gauge, x_h.copy(), clover, cinv, list(X), kappa, mu, +1, "ee", 0) if hasattr(oracle, "tmc_matpc") else None
    finally:
        oracle.set_threads(1)
    for prec in (8, 4, 2):
        _load_fields(qa, gauge, clover, X, kappa, mu, prec, 18)
        ip = qa.invert_param(qa.QUDA_TWISTED_CLOVER_DSLASH, kappa, mu, +1, "ee", 0, cuda_prec=prec)
        assert qc.rel_err(qa.dslash(x_h.copy(), ip, 0), want) < TOL[prec], prec
        if want_m is not None:
            assert qc.rel_err(qa.mat(x_h.copy(), ip), want_m) < 2 * TOL[prec], prec


@pytest.mark.parametrize("fmt", [0, 1], ids=["flag-in-data", "atoms-16B"])
@pytest.mark.parametrize("mask", [1, 2, 4, 8, 6, 9, 15])
def test_partitioned_dslash_self_neighbour(qa, mask, fmt):
    """The reference's own way of testing the halo path without a cluster (tests/test_util.cpp:2047-2065 --partition):
    a single process treats dimension d as partitioned and talks to itself — pack kernel, ghost-zone exchange,
    interior + exterior kernels — and must reproduce the golden vectors exactly like the unpartitioned kernel.
    fmt: wire format of the peer-store ghost zones — flag-in-data {word, flag, word, flag} vectors, or self-validating 16-byte atoms {3 words, flag} (one 128-byte
    line per fp64 face site; ADVICE r3: every 16 bytes validate themselves), selected at run time through the tune key the environment variable sets."""
    qa.lib().qudaAmdSetDslashTune(b"halo_format", fmt)
    z, X, kappa, mu, gauge = qc.load(qc.FILES[1])  # 6x4x2x8: includes an extent-2 dimension
    names = ["wil_dslash_p0_d0", "wil_dslash_p1_d1", "tm_dslash_fp_ee_d0_p0", "tm_dslash_fm_oo_d1_p0", "tm_dslash_fp_ee_d1_p0",
             "tmc_dslash_fp_ee_d0_p0", "tmc_dslash_fm_ooasym_d1_p0", "tm_matpc_fp_ee_d0", "tm_matpc_fp_oo_d1", "tm_mat_fp_d0", "tmc_matpc_fm_eeasym_d1", "tmc_matpc_fm_ee_d1"]
    os.environ["QUDA_AMD_FORCE_GAUGE_HALO"] = "1"  # the link ghost exchange of loadGaugeQuda goes through the same self path
    try:
        for prec, recon in ((8, 18), (4, 12), (2, 18)):
            qa.lib().qudaAmdSetPartitionMask(mask)
            _load_fields(qa, gauge, z["clover"], X, kappa, mu, prec, recon)
            for name in names:
                got = qc.run_abi(qa, name, z["spinor"], X, kappa, mu, prec)
                assert qc.rel_err(got, z[name]) < TOL[prec], (name, prec, mask)
            qa.lib().qudaAmdSetPartitionMask(0)
        assert int(qa.lib().qudaAmdHaloTransport()) == 1 and qa.comm_stats()["fine_peer_store_exchanges"] > 0
    finally:
        qa.lib().qudaAmdSetPartitionMask(0)
        qa.lib().qudaAmdSetDslashTune(b"halo_format", -1)


@pytest.mark.parametrize("X", [(32, 16, 16, 16), (32, 16, 16, 32)], ids=["32^4-over-8", "configs3-32^3x64-over-8"])
@pytest.mark.parametrize("fmt", [0, 1], ids=["flag-in-data", "atoms-16B"])
def test_partitioned_dslash_at_the_8gpu_sublattice(qa, oracle, fmt, X):
    """the local lattice of an 8-GPU split (grid 1 x 2 x 2 x 2, y z t partitioned) of 32^4 (32 x 16 x 16 x 16) and of BASELINE configs[3],
    32^3 x 64 (32 x 16 x 16 x 32): many pack blocks, faces of different sizes, pack blocks that straddle two (dimension, direction)
    ranges; all three precisions against the oracle, both wire formats"""
    kappa, mu = 0.1, 0.01
    gauge, spinor, _ = oracle.make_fields(list(X), clover=False)
    nh = spinor.size // 2
    oracle.set_threads(8)
    try:
        want = oracle.tm_dslash(gauge, spinor[:nh].copy(), list(X), kappa, mu, +1, 0, "ee", 0)
        want_m = oracle.tm_matpc(gauge, spinor[:nh].copy(), list(X), kappa, mu, +1, "ee", 0)
    finally:
        oracle.set_threads(1)
    qa.lib().qudaAmdSetDslashTune(b"halo_format", fmt)
    qa.lib().qudaAmdSetPartitionMask(0b1110)
    try:
        for prec in (8, 4, 2):
            qa.load_gauge(gauge, qa.gauge_param(X, cuda_prec=prec))
            ip = qa.invert_param(qa.QUDA_TWISTED_MASS_DSLASH, kappa, mu, +1, "ee", 0, cuda_prec=prec)
            for _ in range(3):   # first use (verification against the staged transport), then both buffers of the steady state
                got = qa.dslash(spinor[:nh].copy(), ip, 0)
                assert qc.rel_err(got, want) < TOL[prec], (prec, fmt)
            assert qc.rel_err(qa.mat(spinor[:nh].copy(), ip), want_m) < 2 * TOL[prec], (prec, fmt)
        assert int(qa.lib().qudaAmdHaloTransport()) == 1
    finally:
        qa.lib().qudaAmdSetPartitionMask(0)
        qa.lib().qudaAmdSetDslashTune(b"halo_format", -1)


def test_rccl_call_sequence_self_loop():
    """The real RCCL calls of the multi-GPU path (ncclCommInitRank, grouped ncclSend/ncclRecv, ncclAllReduce) on one GPU:
    a one-rank communicator in self-test mode sends every halo message to itself through RCCL (tools/rccl_selftest.py)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "rccl_selftest.py")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "RCCL self-loop OK" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])


def test_two_process_decomposition_on_one_gpu():
    """Two real processes (ranks) sharing this GPU, lattice split 1x1x1x2 / 1x1x2x1 / 2x1x1x1: halo through the IPC-mapped
    peer-store transport (each rank's pack blocks store into the other PROCESS' ghost window), collectives through the file
    transport (RCCL refuses two ranks on one device).  Dslash / Mat / MatPC in three precisions, distributed GCR and MG-GCR
    against the single-lattice oracle (tools/mgpu_check.py)."""
    import subprocess
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory() as tmp:
        log = os.path.join(tmp, "rehearsal.log")
        env = dict(os.environ, QUDA_AMD_P2P_TIMEOUT_S="5")
        subprocess.run([os.path.join(root, "tools", "mgpu_rehearsal.sh"), "2", log], cwd=root, env=env, timeout=400, check=True)
        text = open(log).read()
    assert "rehearsal rc=0" in text, text[-3000:]
    assert "rank 0: all checks passed" in text and "rank 1: all checks passed" in text, text[-3000:]
    assert text.count("MG-GCR") >= 6


@pytest.mark.parametrize("prec,recon,tol", [(8, 18, 1e-12), (8, 12, 1e-12), (4, 18, 2e-6)])
def test_device_clover_construction(qa, oracle, prec, recon, tol):
    """loadCloverQuda(NULL, NULL): the clover term built on the device from the resident links (reference createCloverQuda,
    lib/interface_quda.cpp:3950-4010) against the oracle's restatement of computeFmunu + computeClover, then the
    twisted-clover operator on top of it against the oracle's tmc_mat / tmc_matpc fed with the oracle-built clover.
    Anti-periodic links: the boundary sign must cancel in every plaquette (folded for recon-18, reconstructed for 12)."""
    X, kappa, mu, coeff = [8, 4, 6, 8], 0.12, 0.3, 0.17
    gauge, spinor, _ = oracle.make_fields(X, seed=77, antiperiodic_t=True, clover=False)
    want_clover = oracle.clover_compute(gauge, coeff, X)
    qa.load_gauge(gauge, qa.gauge_param(X, cuda_prec=prec, recon=recon))
    ip = qa.invert_param(qa.QUDA_TWISTED_CLOVER_DSLASH, kappa, mu, +1, "ee", 0, cuda_prec=prec, solution_type=qa.QUDA_MAT_SOLUTION)
    ip.clover_coeff = coeff
    ip.compute_clover, ip.return_clover, ip.return_clover_inverse = 1, 1, 1
    got_clover, got_inv = np.zeros_like(want_clover), np.zeros_like(want_clover)
    qa.load_clover(got_clover, got_inv, ip)
    assert np.max(np.abs(got_clover - want_clover)) < tol * 10
    want_inv = oracle.clover_twisted_inverse(want_clover, 4 * kappa * kappa * mu * mu)
    assert np.max(np.abs(got_inv - want_inv)) < tol * 100
    # and with nothing handed over at all, as the QKXTM drivers call it
    ip.compute_clover, ip.return_clover, ip.return_clover_inverse = 0, 0, 0
    qa.load_clover(None, None, ip)
    got = qa.mat(spinor.copy(), ip)
    want = oracle.tmc_mat(gauge, want_clover, spinor, X, kappa, mu, +1, 0)
    assert qc.rel_err(got, want) < tol * 10
    ipc = qa.invert_param(qa.QUDA_TWISTED_CLOVER_DSLASH, kappa, mu, -1, "oo", 1, cuda_prec=prec)
    nh = spinor.size // 2
    got = qa.mat(spinor[nh:].copy(), ipc)
    want = oracle.tmc_matpc(gauge, spinor[nh:].copy(), want_clover, oracle.clover_twisted_inverse(want_clover, 4 * kappa * kappa * mu * mu), X, kappa, mu, -1, "oo", 1)
    assert qc.rel_err(got, want) < tol * 100
    qa.lib().freeCloverQuda()


@pytest.mark.parametrize("mask", [0, 15, 6], ids=["forced", "self-neighbour-xyzt", "self-neighbour-yz"])
@pytest.mark.parametrize("prec,recon,tol", [(8, 18, 1e-11), (8, 12, 1e-11), (4, 18, 2e-5)])
def test_device_clover_construction_by_transport(qa, oracle, mask, prec, recon, tol):
    """The grid-decomposed formulation of the clover construction (plaquette field built from forward-shifted links, the
    other three leaves obtained by transporting it with ghost-aware shifts) must reproduce the oracle — unpartitioned
    (forced with QUDA_AMD_CLOVER_TRANSPORT) and with self-neighbour partitioning, where every shift crosses a 'rank' boundary."""
    X, kappa, mu, coeff = [8, 4, 6, 8], 0.12, 0.3, 0.17
    gauge, _, _ = oracle.make_fields(X, seed=78, antiperiodic_t=True, clover=False)
    want = oracle.clover_compute(gauge, coeff, X)
    os.environ["QUDA_AMD_CLOVER_TRANSPORT"] = "1"
    try:
        qa.lib().qudaAmdSetPartitionMask(mask)
        qa.load_gauge(gauge, qa.gauge_param(X, cuda_prec=prec, recon=recon))
        ip = qa.invert_param(qa.QUDA_TWISTED_CLOVER_DSLASH, kappa, mu, +1, "ee", 0, cuda_prec=prec, solution_type=qa.QUDA_MAT_SOLUTION)
        ip.clover_coeff = coeff
        ip.compute_clover, ip.return_clover = 1, 1
        got = np.zeros_like(want)
        qa.load_clover(got, None, ip)
        assert np.max(np.abs(got - want)) < tol * 10
    finally:
        del os.environ["QUDA_AMD_CLOVER_TRANSPORT"]
        qa.lib().qudaAmdSetPartitionMask(0)
        qa.lib().freeCloverQuda()


@pytest.mark.parametrize("case,message", [
    ("not_initialized", "QUDA not initialized"),
    ("sentinel_gauge_param", "Parameter"),
    ("dslash_without_gauge", "Gauge field not allocated"),
    ("unsupported_dslash_type", "Unsupported dslash_type"),
    ("clover_without_coefficient", "clover coefficient not set"),
    ("mg_outer_pc_with_full_smoother", "a preconditioned smoother is required"),
])
def test_error_convention(case, message):
    """No return codes and no exceptions, as the reference (include/util_quda.h:51-61): `ERROR: <text> (rank, file:line in
    func())` on stdout and exit status 1; parameter structs are validated against their "invalid" sentinels
    (lib/check_params.h).  Each case runs in a child process (tools/error_cases.py)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "error_cases.py"), case], capture_output=True, text=True, timeout=300)
    out = r.stdout + r.stderr
    assert r.returncode == 1, (r.returncode, out[-1500:])
    assert "ERROR:" in out and message in out and "NOT REACHED" not in out, out[-1500:]
    assert "csrc/" in out and " in " in out   # file:line in func()


def test_missing_neighbour_face_is_an_error_not_a_hang():
    """Peer-store transport: the in-kernel wait for a neighbour's face has a wall-clock timeout.  Two processes on this GPU;
    rank 1 leaves after the first exchange, rank 0 applies the operator again and must abort with the time-out message."""
    import subprocess
    import sys
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory(dir="/dev/shm") as shm:
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(20000 + os.getpid() % 20000), WORLD_SIZE="2", QUDA_AMD_FORCE_DEVICE="0", QUDA_AMD_TRANSPORT="shm",
                   QUDA_AMD_SHM_DIR=shm, QUDA_AMD_P2P_TIMEOUT_S="1")
        procs = [subprocess.Popen([sys.executable, os.path.join(root, "tools", "p2p_timeout_check.py")], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                                  stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
        outs = [p.communicate(timeout=120)[0] for p in procs]
    assert "transport 1" in outs[0], outs[0][-1500:]
    assert procs[0].returncode == 1 and "halo wait ran out" in outs[0] and "NOT REACHED" not in outs[0], outs[0][-1500:]
    # the record says where and why: t is the partitioned dimension, the face comes from rank 1, and the words still
    # carry the zone's previous use — the face of this exchange was never written
    assert "in dimension 3" in outs[0] and "face from rank 1" in outs[0] and "expected flag" in outs[0], outs[0][-1500:]
    assert "never written" in outs[0], outs[0][-1500:]


@pytest.mark.parametrize("forced_failure", [False, True])
def test_peer_store_halo_is_verified_on_first_use(forced_failure):
    """The first partitioned application of every precision runs through both transports and compares (csrc/dslash.hip); a
    failed comparison (forced here through the test hook) drops every rank back to the staged transport, results stay right."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env.pop("QUDA_AMD_HALO", None)
    if forced_failure:
        env["QUDA_AMD_P2P_VERIFY_FAIL"] = "1"
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "p2p_verify_fallback.py")], capture_output=True, text=True, timeout=300, env=env)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-1500:]
    assert ("transport 0" if forced_failure else "transport 1") in out, out[-1500:]


def test_tune_cache_is_filled_by_a_sweep_and_read_back(qa, oracle, tmp_path):
    """f4, the launch-parameter half (reference lib/tune.cpp:213-355): with QudaInvertParam.tune = QUDA_TUNE_YES the first application of a
    (lattice, kernel, precision, reconstruct, epilogue, partition mask) key times the candidate knob settings interleaved and keeps the
    fastest; the table lands in $QUDA_RESOURCE_PATH/tunecache.tsv in the reference's text format; a second library start reads it and
    sweeps nothing.  The operator's result is the oracle's while tuning and with the cached parameters."""
    X, kappa, mu = (16, 16, 16, 16), 0.1, 0.01
    gauge, spinor, _ = oracle.make_fields(list(X), clover=False)
    nh = spinor.size // 2
    oracle.set_threads(8)
    try:
        want = oracle.tm_dslash(gauge, spinor[:nh].copy(), list(X), kappa, mu, +1, 0, "ee", 0)
    finally:
        oracle.set_threads(1)
    L = qa.lib()
    L.qudaAmdTuneSweeps.restype = C.c_long
    qa.end()
    os.environ["QUDA_RESOURCE_PATH"] = str(tmp_path)
    try:
        qa.init(0)
        s0 = L.qudaAmdTuneSweeps()
        for prec in (8, 2):
            qa.load_gauge(gauge, qa.gauge_param(X, cuda_prec=prec))
            ip = qa.invert_param(qa.QUDA_TWISTED_MASS_DSLASH, kappa, mu, +1, "ee", 0, cuda_prec=prec)
            ip.tune = 1   # QUDA_TUNE_YES
            for _ in range(2):
                assert qc.rel_err(qa.dslash(spinor[:nh].copy(), ip, 0), want) < TOL[prec]
        assert L.qudaAmdTuneSweeps() - s0 == 2          # one sweep per key, none for the repeated application
        L.qudaAmdSetPartitionMask(0b1010)
        qa.load_gauge(gauge, qa.gauge_param(X, cuda_prec=4))
        ip = qa.invert_param(qa.QUDA_TWISTED_MASS_DSLASH, kappa, mu, +1, "ee", 0, cuda_prec=4)
        ip.tune = 1
        for _ in range(3):   # first use of the peer-store transport is verified, the next application is tuned, the third runs from the table
            assert qc.rel_err(qa.dslash(spinor[:nh].copy(), ip, 0), want) < TOL[4]
        L.qudaAmdSetPartitionMask(0)
        assert L.qudaAmdTuneSweeps() - s0 == 3
        text = (tmp_path / "tunecache.tsv").read_text()
        rows = [ln.split("\t") for ln in text.split("\n")[3:] if ln]
        assert text.startswith("tunecache\t") and len(rows) == 3
        assert sorted(r[2].split(",")[0] for r in rows) == ["prec=2", "prec=4", "prec=8"] and all(r[0].strip() == "16x16x16x16" and r[1] == "dslash_kernel" for r in rows)
        assert [r for r in rows if r[2].startswith("prec=4")][0][2].endswith("comm=0101")
        # second start: the table comes from the file, nothing is swept
        qa.end()
        qa.init(0)
        s1 = L.qudaAmdTuneSweeps()
        qa.load_gauge(gauge, qa.gauge_param(X, cuda_prec=8))
        ip = qa.invert_param(qa.QUDA_TWISTED_MASS_DSLASH, kappa, mu, +1, "ee", 0, cuda_prec=8)
        ip.tune = 1
        assert qc.rel_err(qa.dslash(spinor[:nh].copy(), ip, 0), want) < TOL[8]
        assert L.qudaAmdTuneSweeps() == s1
    finally:
        L.qudaAmdSetPartitionMask(0)
        qa.end()
        del os.environ["QUDA_RESOURCE_PATH"]
        qa.init(0)
