"""The solve loop of the QKXTM drivers through the C ABI (reference lib/interface_quda.cpp:6018-6531):
Gaussian smearing against oracle/qo_qkxtm.c, and the up / down propagators of a smeared point source, each checked the way
the reference checks a solve (tests/multigrid_invert_test.cpp:529-577): the oracle's host tm_mat applied to the returned
solution must reproduce the source to the solver tolerance."""
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


def _lex_gauge(oracle, gauge, X):
    return np.stack([oracle.eo_to_lex(np.ascontiguousarray(gauge[d]), list(X), 18) for d in range(4)])


@pytest.mark.parametrize("X", [(4, 4, 4, 4), (6, 4, 2, 8)])
@pytest.mark.parametrize("mask", [0, 0b0110])
def test_gaussian_smearing_matches_oracle(qa, oracle, X, mask):
    """fp64, same operation order up to the summation order of the six hops: 1e-12 relative per site"""
    gauge, _, _ = oracle.make_fields(list(X), seed=7, antiperiodic_t=False, clover=False)
    gp = qa.gauge_param(X, t_boundary=qa.QUDA_PERIODIC_T)
    qa.load_gauge(gauge, gp)
    g_lex = _lex_gauge(oracle, gauge, X)
    rng = np.random.default_rng(2)
    v = rng.standard_normal(int(np.prod(X)) * 24)
    qa.lib().qudaAmdSetPartitionMask(mask)
    try:
        got = qa.gaussian_smear(v, g_lex, 5, 0.8)
    finally:
        qa.lib().qudaAmdSetPartitionMask(0)
    want = oracle.gauss_smear(v, g_lex, list(X), 0.8, 5)
    assert np.max(np.abs(got - want)) < 1e-12 * np.max(np.abs(want))
    # zero steps: the identity (through the site reordering and both spin rotations)
    assert np.max(np.abs(qa.gaussian_smear(v, g_lex, 0, 0.8) - v)) < 1e-14 * np.max(np.abs(v))


@pytest.mark.parametrize("X", [(4, 4, 4, 4), (6, 4, 2, 8)])
@pytest.mark.parametrize("mask,prec,recon", [(0, 8, 18), (0b0111, 8, 18), (0b1010, 4, 12)])
def test_plaquette_and_ape_smearing_match_oracle(qa, oracle, X, mask, prec, recon):
    """plaqQuda / performAPEnStep / saveGaugeQuda (reference lib/gauge_plaq.cu, lib/gauge_ape.cu) against oracle/qo_qkxtm.c; with a
    partition mask the staples come through the ghost-aware shifts, exactly as on a decomposed lattice.  fp64 links: 1e-12;
    fp32 recon-12 resident links: the smearing itself runs in fp64, the input rounding gives 1e-5."""
    gauge, _, _ = oracle.make_fields(list(X), seed=31, antiperiodic_t=True, clover=False)
    gp = qa.gauge_param(X, cuda_prec=prec, recon=recon)
    tol = 1e-12 if prec == 8 else 2e-5
    qa.lib().qudaAmdSetPartitionMask(mask)
    try:
        qa.load_gauge(gauge, gp)
        assert np.max(np.abs(qa.save_gauge(gp) - gauge)) < (1e-15 if prec == 8 else 1e-6)
        pl, want = np.array(qa.plaquette()), oracle.plaquette(gauge, list(X))
        assert np.max(np.abs(pl - want)) < tol, (pl, want)
        qa.perform_ape(3, 0.5)
        got = qa.save_smeared_gauge(int(np.prod(X)))
        want = oracle.ape_smear(gauge, list(X), 0.5, 3)
        assert np.max(np.abs(got - want)) < tol
        # the lexicographic copy is the layout gauge_APE takes, and NULL means "use the resident smeared field"
        lex = qa.save_smeared_gauge(int(np.prod(X)), lexicographic=True)
        assert np.array_equal(lex, _lex_gauge(oracle, got, X))
        v = np.random.default_rng(3).standard_normal(int(np.prod(X)) * 24)
        assert np.array_equal(qa.gaussian_smear(v, None, 2, 0.7), qa.gaussian_smear(v, lex, 2, 0.7))
    finally:
        qa.lib().qudaAmdSetPartitionMask(0)


def _check_propagators(qa, oracle, gauge, g_lex, X, kappa, mu, ip, pos, nsmear, alpha, normalized, clover=None):
    V = int(np.prod(X))
    up, dn = qa.calc_mg_propagators(g_lex, ip, pos, nsmear, alpha, V)
    worst = 0.0
    oracle.set_threads(8)
    try:
        for isc in range(12):
            src = np.zeros(V * 24)
            iv = ((pos[3] * X[2] + pos[2]) * X[1] + pos[1]) * X[0] + pos[0]
            src[iv * 24 + isc * 2] = 1.0
            b_lex = oracle.gauss_smear(src, g_lex, list(X), alpha, nsmear) if nsmear else src
            b = oracle.lex_to_eo(oracle.ukqcd_to_dr(b_lex.reshape(-1, 24)).reshape(-1), list(X), 24)
            for flavor, prop in ((+1, up), (-1, dn)):
                x = oracle.lex_to_eo(oracle.ukqcd_to_dr(prop[isc].reshape(-1, 24)).reshape(-1), list(X), 24)
                if normalized:
                    x = x / (2 * kappa)
                mx = oracle.tm_mat(gauge, x, list(X), kappa, mu, flavor, 0) if clover is None else oracle.tmc_mat(gauge, clover, x, list(X), kappa, mu, flavor, 0)
                res = float(np.linalg.norm(b - mx) / np.linalg.norm(b))
                worst = max(worst, res)
    finally:
        oracle.set_threads(1)
    return worst


def test_propagators_plain_gcr(qa, oracle):
    """no multigrid: the loop itself (source, smearing, flavour flip, prepare / reconstruct, 2 kappa rescaling, layouts)"""
    X, kappa, mu = (4, 4, 4, 8), 0.12, 0.05
    gauge, _, _ = oracle.make_fields(list(X), seed=11, antiperiodic_t=False, clover=False)
    gp = qa.gauge_param(X, t_boundary=qa.QUDA_PERIODIC_T)
    qa.load_gauge(gauge, gp)
    g_lex = _lex_gauge(oracle, gauge, X)
    for matpc, norm in (("ee", qa.QUDA_KAPPA_NORMALIZATION), ("oo", qa.QUDA_MASS_NORMALIZATION)):
        ip = qa.invert_param(qa.QUDA_TWISTED_MASS_DSLASH, kappa, mu, +1, matpc, 0, cuda_prec=8, solution_type=qa.QUDA_MAT_SOLUTION,
                             gamma_basis=qa.QUDA_UKQCD_GAMMA_BASIS)
        ip.solve_type, ip.inv_type, ip.gcrNkrylov, ip.tol, ip.maxiter = qa.QUDA_DIRECT_PC_SOLVE, qa.QUDA_GCR_INVERTER, 20, 1e-10, 4000
        ip.inv_type_precondition = qa.QUDA_INVALID_ENUM
        ip.mass_normalization = norm
        ip.verbosity = qa.QUDA_SILENT
        worst = _check_propagators(qa, oracle, gauge, g_lex, X, kappa, mu, ip, (1, 2, 3, 5), 4, 0.6, norm == qa.QUDA_MASS_NORMALIZATION)
        # the solver's own criterion is 1e-10 on the EVEN-ODD system (BASELINE's bar, met: QudaInvertParam.true_res); what is checked here
        # is stricter in kind — the residual of the reconstructed FULL solution against the oracle's tm_mat on the host — hence the factor
        assert worst < 5e-10, (matpc, worst)
        assert ip.twist_flavor == qa.QUDA_TWIST_MINUS and 24 < ip.iter < 24 * 4000   # last solve was the down quark; iterations are summed


def test_propagators_with_up_and_down_hierarchies(qa, oracle):
    """the production shape: one multigrid hierarchy per flavour in preconditionerUP / preconditionerDN"""
    X, kappa, mu = (8, 8, 8, 16), 0.124, 0.005
    gauge = smooth_gauge(X, 0.35)
    gp = qa.gauge_param(X, cuda_prec=8, prec_sloppy=4, prec_precondition=4, t_boundary=qa.QUDA_PERIODIC_T)
    qa.load_gauge(gauge, gp)
    g_lex = _lex_gauge(oracle, gauge, X)
    hier = {}
    try:
        for flavor in (+1, -1):
            ipm = qa.invert_param(qa.QUDA_TWISTED_MASS_DSLASH, kappa, mu, flavor, "ee", 0, cuda_prec=8, prec_sloppy=4, prec_precondition=4,
                                  solution_type=qa.QUDA_MAT_SOLUTION)
            ipm.solve_type = qa.QUDA_DIRECT_SOLVE
            ipm.inv_type, ipm.gcrNkrylov, ipm.tol, ipm.maxiter, ipm.reliable_delta, ipm.verbosity = qa.QUDA_GCR_INVERTER, 20, 1e-10, 2000, 1e-4, qa.QUDA_SILENT
            mp = qa.multigrid_param(ipm, n_level=3, geo_block=[(4, 4, 4, 4), (1, 1, 1, 2), (2, 2, 2, 2)], n_vec=8, setup_maxiter=300, setup_tol=1e-5, smoother_pc=True,
                                    coarse_matpc=True)
            hier[flavor] = (qa.Multigrid(mp), ipm, mp)
        ip = qa.invert_param(qa.QUDA_TWISTED_MASS_DSLASH, kappa, mu, +1, "ee", 0, cuda_prec=8, prec_sloppy=4, prec_precondition=4,
                             solution_type=qa.QUDA_MAT_SOLUTION, gamma_basis=qa.QUDA_UKQCD_GAMMA_BASIS)
        ip.solve_type, ip.inv_type, ip.gcrNkrylov, ip.tol, ip.maxiter, ip.reliable_delta = qa.QUDA_DIRECT_PC_SOLVE, qa.QUDA_GCR_INVERTER, 20, 1e-10, 2000, 1e-4
        ip.verbosity = qa.QUDA_SILENT
        ip.inv_type_precondition = qa.QUDA_MG_INVERTER
        ip.tol_precondition, ip.maxiter_precondition, ip.precondition_cycle, ip.omega = 1e-1, 1, 1, 1.0
        ip.preconditionerUP, ip.preconditionerDN = hier[+1][0].h, hier[-1][0].h
        s0 = qa.multi_src_stats()
        worst = _check_propagators(qa, oracle, gauge, g_lex, X, kappa, mu, ip, (3, 0, 7, 9), 6, 1.0, False)
        s1 = qa.multi_src_stats()
        print("24 MG-GCR solves: %d outer iterations in total, %.3f s in the solvers, worst true residual %.2e, %s" % (ip.iter, ip.secs, worst, s1))
        assert worst < 5e-10, worst
        assert ip.iter < 24 * 30, ip.iter
        # the twelve sources of a flavour went through ONE lockstep solve each, smoothed on block fields (csrc/block_solver.cpp)
        assert s1["solves"] - s0["solves"] == 2 and s1["block_smoothed"] > s0["block_smoothed"], (s0, s1)
    finally:
        for h, _, _ in hier.values():
            h.free()


def test_twisted_clover_propagators_with_up_and_down_hierarchies(qa, oracle):
    """the production ACTION in the production shape (qkxtm/CalcMG_2pt3pt_EvenOdd.cpp:222-240): QUDA_TWISTED_CLOVER_DSLASH, the clover term
    built on the device by loadCloverQuda(NULL, NULL), one hierarchy per flavour; all 24 solutions checked with the HOST tmc_mat on a clover
    field the oracle constructs independently from the same links"""
    X, kappa, mu = (8, 8, 8, 16), 0.124, 0.005
    coeff = kappa * 1.57551
    gauge = smooth_gauge(X, 0.35)
    gp = qa.gauge_param(X, cuda_prec=8, prec_sloppy=4, prec_precondition=4, t_boundary=qa.QUDA_PERIODIC_T)
    qa.load_gauge(gauge, gp)
    g_lex = _lex_gauge(oracle, gauge, X)
    clover = oracle.clover_compute(gauge, coeff, list(X))
    hier = {}
    try:
        for flavor in (+1, -1):
            ipm = qa.invert_param(qa.QUDA_TWISTED_CLOVER_DSLASH, kappa, mu, flavor, "ee", 0, cuda_prec=8, prec_sloppy=4, prec_precondition=4,
                                  solution_type=qa.QUDA_MAT_SOLUTION)
            ipm.solve_type, ipm.clover_coeff = qa.QUDA_DIRECT_SOLVE, coeff
            ipm.inv_type, ipm.gcrNkrylov, ipm.tol, ipm.maxiter, ipm.reliable_delta, ipm.verbosity = qa.QUDA_GCR_INVERTER, 20, 1e-10, 2000, 1e-4, qa.QUDA_SILENT
            if flavor == +1:
                qa.load_clover(None, None, ipm)
            mp = qa.multigrid_param(ipm, n_level=3, geo_block=[(4, 4, 4, 4), (1, 1, 1, 2), (2, 2, 2, 2)], n_vec=8, setup_maxiter=300, setup_tol=1e-5, smoother_pc=True,
                                    coarse_matpc=True)
            hier[flavor] = (qa.Multigrid(mp), ipm, mp)
        ip = qa.invert_param(qa.QUDA_TWISTED_CLOVER_DSLASH, kappa, mu, +1, "ee", 0, cuda_prec=8, prec_sloppy=4, prec_precondition=4,
                             solution_type=qa.QUDA_MAT_SOLUTION, gamma_basis=qa.QUDA_UKQCD_GAMMA_BASIS)
        ip.clover_coeff = coeff
        ip.solve_type, ip.inv_type, ip.gcrNkrylov, ip.tol, ip.maxiter, ip.reliable_delta = qa.QUDA_DIRECT_PC_SOLVE, qa.QUDA_GCR_INVERTER, 20, 1e-10, 2000, 1e-4
        ip.verbosity = qa.QUDA_SILENT
        ip.inv_type_precondition = qa.QUDA_MG_INVERTER
        ip.tol_precondition, ip.maxiter_precondition, ip.precondition_cycle, ip.omega = 1e-1, 1, 1, 1.0
        ip.preconditionerUP, ip.preconditionerDN = hier[+1][0].h, hier[-1][0].h
        worst = _check_propagators(qa, oracle, gauge, g_lex, X, kappa, mu, ip, (5, 1, 2, 11), 4, 0.8, False, clover=clover)
        print("24 twisted-clover MG-GCR solves: %d outer iterations in total, %.3f s in the solvers, worst true residual %.2e" % (ip.iter, ip.secs, worst))
        assert worst < 5e-10, worst
        assert ip.iter < 24 * 30, ip.iter
    finally:
        for h, _, _ in hier.values():
            h.free()


@pytest.mark.parametrize("case", ["calcmg_not_pc", "calcmg_not_ukqcd"])
def test_parameter_checks(case):
    """the reference's guards (lib/interface_quda.cpp:6041-6054) through the usual error convention; child process (tools/error_cases.py)"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "error_cases.py"), case], capture_output=True, text=True, timeout=300)
    assert r.returncode == 1, (case, r.returncode, r.stdout[-300:], r.stderr[-300:])
    assert "ERROR" in r.stdout + r.stderr and "works only with" in r.stdout + r.stderr


def test_reference_entry_point_names_from_a_cxx_driver(oracle, tmp_path):
    """tests/consumer/qkxtm_driver.cpp — a driver in the shape of qkxtm/CalcMG_2pt3pt_EvenOdd.cpp / CalcMG_Loops_w_oneD_TSM_EvenOdd.cpp —
    calls calcMG_threepTwop_EvenOdd (2 source positions = 48 solves), calcMG_loop_wOneD_TSM_EvenOdd (truncated solver method: 3 LP
    solves + 2 HP/LP pairs) and calcMG_loop_wOneD_TSM_wExact (nEv = 0, 2 sources, down flavour) by the reference's names; every
    solution that reaches the sink is checked with the oracle's tm_mat (the reference's own check of a solve,
    tests/multigrid_invert_test.cpp:529-577): full-precision solves to 5e-10, truncated ones between the two tolerances."""
    import os
    import subprocess
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_dropin_build import build_qkxtm_driver
    X, kappa, mu = (8, 8, 8, 8), 0.124, 0.005
    V = int(np.prod(X))
    gauge = smooth_gauge(X, 0.35)
    gfile, ofile = tmp_path / "gauge.bin", tmp_path / "out.bin"
    np.ascontiguousarray(gauge).tofile(str(gfile))
    exe = build_qkxtm_driver(str(tmp_path))
    r = subprocess.run([exe, str(gfile)] + [str(v) for v in X] + [str(ofile)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    assert "calcMG_threepTwop_EvenOdd" in r.stdout and "calcMG_loop_wOneD_TSM_wExact" in r.stdout
    g_ape_eo = oracle.ape_smear(gauge, list(X), 0.5, 2)
    g_ape = _lex_gauge(oracle, g_ape_eo, X)
    raw = open(str(ofile), "rb").read()
    off, recs = 0, []
    while off < len(raw):
        kind = raw[off:off + 16].split(b"\0")[0].decode(); off += 16
        index, flavor, has_src, nreal = np.frombuffer(raw, dtype=np.int32, count=4, offset=off); off += 16
        src = None
        if has_src:
            src = np.frombuffer(raw, dtype=np.float64, count=nreal, offset=off); off += 8 * nreal
        sol = np.frombuffer(raw, dtype=np.float64, count=nreal, offset=off); off += 8 * nreal
        recs.append((kind, int(index), int(flavor), src, sol))
    kinds = [k for k, *_ in recs]
    assert kinds.count("prop_up") == 24 and kinds.count("prop_dn") == 24
    assert kinds.count("loop_LP") == 3 and kinds.count("loop_HP") == 2 and kinds.count("loop_HP_LP") == 2 and kinds.count("loop_stoch") == 2
    pos = [(1, 2, 3, 5), (0, 3, 1, 2)]
    oracle.set_threads(8)
    try:
        def residual(src_lex, sol_lex, flavor):
            b = oracle.lex_to_eo(oracle.ukqcd_to_dr(src_lex.reshape(-1, 24)).reshape(-1), list(X), 24)
            x = oracle.lex_to_eo(oracle.ukqcd_to_dr(sol_lex.reshape(-1, 24)).reshape(-1), list(X), 24)
            return float(np.linalg.norm(b - oracle.tm_mat(gauge, x, list(X), kappa, mu, flavor, 0)) / np.linalg.norm(b))
        seen_lp = []
        for kind, index, flavor, src, sol in recs:
            if kind.startswith("prop"):
                isource, isc = divmod(index, 12)
                p = [pos[isource][d] % X[d] for d in range(4)]
                point = np.zeros(V * 24)
                point[(((p[3] * X[2] + p[2]) * X[1] + p[1]) * X[0] + p[0]) * 24 + isc * 2] = 1.0
                src = oracle.gauss_smear(point, g_ape, list(X), 0.8, 3)
                assert flavor == (+1 if kind == "prop_up" else -1)
                assert residual(src, sol, flavor) < 5e-10, (kind, index)
            else:
                # Z4 noise: every component is one of 1, -1, i, -i
                c = src.reshape(-1, 2)
                assert np.all(np.abs(c).sum(axis=1) == 1.0) and set(np.unique(c)) <= {-1.0, 0.0, 1.0}
                res = residual(src, sol, flavor)
                if kind in ("loop_LP", "loop_HP_LP"):
                    assert 1e-8 < res < 5e-3, (kind, index, res)   # stopped at TSM_tol = 1e-3 (even-odd system), well short of 1e-10
                    seen_lp.append(res)
                else:
                    assert res < 5e-10, (kind, index, res)
                assert flavor == (-1 if kind == "loop_stoch" else +1)
        assert len(seen_lp) == 5
        # the HP / LP pair of the truncated solver method is solved from the SAME source
        hp = {i: s for k, i, f, s, x in recs if k == "loop_HP"}
        lp = {i: s for k, i, f, s, x in recs if k == "loop_HP_LP"}
        assert all(np.array_equal(hp[i], lp[i]) for i in hp)
    finally:
        oracle.set_threads(1)
