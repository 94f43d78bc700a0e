#!/usr/bin/env python3
"""Validate the RCCL call sequence of the multi-GPU path on ONE GPU: a one-rank RCCL communicator with
QUDA_AMD_RCCL_SELFTEST=1 routes every self-neighbour halo message through grouped ncclSend/ncclRecv (peer = own rank)
and every reduction through ncclAllReduce.  Checks golden Dslash/Mat cases in three precisions, a GCR solve and an
MG-GCR solve on the fully self-partitioned lattice.  Run by tests/test_dslash_gpu.py::test_rccl_call_sequence_self_loop
in a child process (the communicator must exist before initQuda)."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["QUDA_AMD_RCCL_SELFTEST"] = "1"
os.environ.setdefault("QUDA_AMD_HALO", "rccl")   # the Dslash halo through RCCL too (default would be direct peer stores)
os.environ["QUDA_AMD_FORCE_GAUGE_HALO"] = "1"
import multi_gpu  # noqa: E402
import oracle_api  # noqa: E402
import qa_cases as qc  # noqa: E402
from synth import smooth_gauge  # noqa: E402


def main():
    qa = importlib.import_module("quda-qkxtm-multigrid_amd")
    z, X, kappa, mu, gauge = qc.load(qc.FILES[1])
    multi_gpu.setup(qa, 0, 1, 0, X, grid=[1, 1, 1, 1])
    qa.lib().qudaAmdSetPartitionMask(15)
    names = ["wil_dslash_p0_d0", "tm_dslash_fp_ee_d0_p0", "tm_dslash_fm_oo_d1_p0", "tmc_dslash_fp_ee_d0_p0", "tm_matpc_fp_ee_d0", "tm_mat_fp_d0",
             "tmc_matpc_fm_ee_d1"]
    for prec, tol in ((8, 1e-12), (4, 2e-5), (2, 1e-2)):
        qa.load_gauge(gauge, qa.gauge_param(X, cuda_prec=prec))
        ipc = qa.invert_param(qa.QUDA_TWISTED_CLOVER_DSLASH, kappa, mu, cuda_prec=prec)
        qa.load_clover(z["clover"], None, ipc)
        for name in names:
            err = qc.rel_err(qc.run_abi(qa, name, z["spinor"], X, kappa, mu, prec), z[name])
            assert err < tol, (name, prec, err)
    # solver + multigrid: halo of fine and coarse operators and all reductions through RCCL
    oracle = oracle_api.load()
    Xm, km, mum = (8, 8, 8, 8), 0.124, 0.005
    g = smooth_gauge(Xm, 0.35)
    qa.load_gauge(g, qa.gauge_param(Xm, cuda_prec=8, prec_sloppy=4, prec_precondition=4, t_boundary=qa.QUDA_PERIODIC_T))
    ip = qa.invert_param(qa.QUDA_TWISTED_MASS_DSLASH, km, mum, +1, "ee", 0, cuda_prec=8, prec_sloppy=4, prec_precondition=4, solution_type=qa.QUDA_MAT_SOLUTION)
    ip.solve_type, ip.inv_type, ip.gcrNkrylov, ip.tol, ip.maxiter, ip.reliable_delta = qa.QUDA_DIRECT_SOLVE, qa.QUDA_GCR_INVERTER, 20, 1e-10, 2000, 1e-4
    b = np.random.default_rng(5).random(int(np.prod(Xm)) * 24)
    ip.inv_type_precondition = qa.QUDA_INVALID_ENUM
    qa.invert(b, ip)
    plain = ip.iter
    mg = qa.Multigrid(qa.multigrid_param(ip, n_level=2, geo_block=(4, 4, 4, 4), n_vec=8, setup_maxiter=300, setup_tol=1e-5, smoother_pc=True))
    dev = mg.verify()
    assert max(dev) < 1e-4, dev
    ip.inv_type_precondition, ip.preconditioner = qa.QUDA_MG_INVERTER, mg.h
    ip.tol_precondition, ip.maxiter_precondition, ip.precondition_cycle, ip.omega = 1e-1, 1, 1, 1.0
    x = qa.invert(b, ip)
    oracle.set_threads(8)
    res = float(np.linalg.norm(b - oracle.tm_mat(g, x, list(Xm), km, mum, +1, 0)) / np.linalg.norm(b))
    assert res < 5e-10 and ip.iter < plain // 2, (res, ip.iter, plain)
    mg.free()
    qa.end()
    print("RCCL self-loop OK: golden cases in 3 precisions, MG-GCR %d iterations (plain %d), residual %.2e" % (ip.iter, plain, res))


if __name__ == "__main__":
    main()
