"""The multigrid cycle below level 0 as ONE persistent kernel (include/coarse_cycle.h, csrc/coarse_cycle.hip) against the
kernel-per-operation path it replaces (MG::cycleUnfused: the reference's MG::operator(), lib/multigrid.cpp:488-604, with MR smoothers
lib/inv_mr_quda.cpp:40-200 and the coarsest-grid GCR lib/inv_gcr_quda.cpp:235-516): the same cycle x = K b on a coarse level, the same
outer iteration count, the true residual of the MG-GCR solution recomputed on the host with the oracle's tm_mat.  Partition masks run
the in-kernel halo exchange (flag-in-data words pushed into the — here: own — peer windows, polled by the sites that hop across)."""
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
from test_mg_gpu import _setup, _true_residual  # noqa: E402

BLOCKS3 = [(4, 4, 4, 4), (2, 2, 2, 2), (2, 2, 2, 2)]


@pytest.mark.parametrize("mask", [0, 14, 15, 9], ids=["unpartitioned", "self-neighbour-yzt", "self-neighbour-xyzt", "self-neighbour-xt"])
@pytest.mark.parametrize("X,levels,blocks,nvec", [((16, 8, 8, 16), 3, BLOCKS3, 8), ((16, 16, 16, 16), 3, BLOCKS3, 24), ((8, 8, 8, 16), 2, (4, 4, 4, 4), 32)],
                         ids=["3-levels-n16", "3-levels-n48", "2-levels-n64"])
def test_fused_coarse_cycle_matches_the_kernel_per_operation_path(qa, oracle, X, levels, blocks, nvec, mask):
    kappa, mu = 0.124, 0.005
    qa.lib().qudaAmdSetPartitionMask(mask)
    qa.lib().qudaAmdMultigridSetFused(1)
    try:
        gauge, ip = _setup(qa, X, kappa, mu)
        mp = qa.multigrid_param(ip, n_level=levels, geo_block=blocks, n_vec=nvec, setup_maxiter=300, setup_tol=1e-5, smoother_pc=True, cycle=qa.QUDA_MG_CYCLE_VCYCLE)
        mg = qa.Multigrid(mp)
        try:
            i0 = mg.level_info(0)
            Vc = int(np.prod(i0["Xc"]))
            rng = np.random.default_rng(23)
            for trial in range(2):
                bc = (rng.standard_normal((Vc, 2, nvec)) + 1j * rng.standard_normal((Vc, 2, nvec))).astype(np.complex64)
                qa.lib().qudaAmdMultigridSetFused(0)
                want = mg.apply(1, "K", bc)
                qa.lib().qudaAmdMultigridSetFused(1)
                got = mg.apply(1, "K", bc)      # trial 0: first use (checked inside the library too), trial 1: steady state
                st = mg.fused_stats(1)
                assert st is not None, "level 1 did not get its persistent cycle kernel"
                err = float(np.max(np.abs(got - want)) / np.max(np.abs(want)))
                assert err < 2e-4, (err, st)
                assert st["barriers"] > 0 and st["gcr_iters"] > 0 and (st["halo_exchanges"] > 0) == (mask != 0), st
            # the whole solve: same outer iteration count with and without the fused kernel, residual on the host
            b = rng.random(int(np.prod(X)) * 24)
            ip.inv_type_precondition, ip.preconditioner = qa.QUDA_MG_INVERTER, mg.h
            ip.tol_precondition, ip.maxiter_precondition, ip.precondition_cycle, ip.omega = 1e-1, 1, 1, 1.0
            qa.lib().qudaAmdMultigridSetFused(0)
            x0 = qa.invert(b, ip)
            it0 = ip.iter
            qa.lib().qudaAmdMultigridSetFused(1)
            x1 = qa.invert(b, ip)
            it1 = ip.iter
            assert abs(it1 - it0) <= 1, (it0, it1)
            res = _true_residual(oracle, gauge, X, kappa, mu, x1, b)
            assert res < 1e-10, res
            print("fused cycle %s mask %d: %s, outer iterations %d (kernel-per-operation %d), residual %.2e" % (X, mask, mg.fused_stats(1), it1, it0, res))
        finally:
            mg.free()
    finally:
        qa.lib().qudaAmdSetPartitionMask(0)
        qa.lib().qudaAmdMultigridSetFused(1)


def test_fused_cycle_falls_back_when_its_first_use_check_fails(qa, oracle, monkeypatch):
    """the first-use comparison is what protects a multi-GPU run from a transport that does not behave: forced to fail, the hierarchy must
    keep the kernel-per-operation path and still solve"""
    X, kappa, mu = (16, 8, 8, 16), 0.124, 0.005
    monkeypatch.setenv("QUDA_AMD_MG_FUSED_VERIFY_FAIL", "1")
    qa.lib().qudaAmdSetPartitionMask(14)
    try:
        gauge, ip = _setup(qa, X, kappa, mu)
        mp = qa.multigrid_param(ip, n_level=3, geo_block=BLOCKS3, n_vec=8, setup_maxiter=300, setup_tol=1e-5, smoother_pc=True, cycle=qa.QUDA_MG_CYCLE_VCYCLE)
        mg = qa.Multigrid(mp)
        try:
            b = np.random.default_rng(3).random(int(np.prod(X)) * 24)
            ip.inv_type_precondition, ip.preconditioner = qa.QUDA_MG_INVERTER, mg.h
            ip.tol_precondition, ip.maxiter_precondition, ip.precondition_cycle, ip.omega = 1e-1, 1, 1, 1.0
            x = qa.invert(b, ip)
            assert mg.fused_stats(1) is None
            assert _true_residual(oracle, gauge, X, kappa, mu, x, b) < 1e-10
        finally:
            mg.free()
    finally:
        qa.lib().qudaAmdSetPartitionMask(0)


@pytest.mark.parametrize("mask", [0, 14], ids=["unpartitioned", "self-neighbour-yzt"])
def test_fused_coarsest_gcr_through_restarts(qa, oracle, mask):
    """a tight coarsest-grid tolerance drives the in-kernel GCR through full Krylov spaces, restarts and true-residual updates (reference
    lib/inv_gcr_quda.cpp:300-470): same solution as the host-driven GCR of the kernel-per-operation path"""
    X, kappa, mu, nvec = (8, 8, 8, 16), 0.124, 0.005, 8
    qa.lib().qudaAmdSetPartitionMask(mask)
    try:
        gauge, ip = _setup(qa, X, kappa, mu)
        mp = qa.multigrid_param(ip, n_level=2, geo_block=(4, 4, 4, 4), n_vec=nvec, setup_maxiter=300, setup_tol=1e-5, smoother_pc=True, cycle=qa.QUDA_MG_CYCLE_VCYCLE, smoother_tol=1e-9)
        mg = qa.Multigrid(mp)
        try:
            Vc = int(np.prod(mg.level_info(0)["Xc"]))
            rng = np.random.default_rng(29)
            bc = (rng.standard_normal((Vc, 2, nvec)) + 1j * rng.standard_normal((Vc, 2, nvec))).astype(np.complex64)
            qa.lib().qudaAmdMultigridSetFused(0)
            want = mg.apply(1, "K", bc)
            qa.lib().qudaAmdMultigridSetFused(1)
            for _ in range(2):
                got = mg.apply(1, "K", bc)
            st = mg.fused_stats(1)
            assert st is not None and st["gcr_iters"] > 20 and st["gcr_restarts"] >= 1, st
            # both are solutions of the coarsest system to 1e-6: compare through the operator, |M x - b| / |b|
            for x in (got, want):
                r = mg.apply(1, "M", x) - bc
                assert np.linalg.norm(r) < 1e-5 * np.linalg.norm(bc), (np.linalg.norm(r) / np.linalg.norm(bc), st)
            assert np.linalg.norm(got - want) < 1e-4 * np.linalg.norm(want), st
            print("fused coarsest GCR mask %d: %s" % (mask, st))
        finally:
            mg.free()
    finally:
        qa.lib().qudaAmdSetPartitionMask(0)
        qa.lib().qudaAmdMultigridSetFused(1)
