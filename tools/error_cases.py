#!/usr/bin/env python3
"""Error convention of the C ABI (reference include/util_quda.h:51-61): any failure prints `ERROR: ... (rank, file:line in func())`
and exits with status 1.  usage: error_cases.py <case>"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
qa = importlib.import_module("quda-qkxtm-multigrid_amd")
case = sys.argv[1]
X = [4, 4, 4, 4]
V = int(np.prod(X))
if case == "not_initialized":
    gp = qa.gauge_param(X)
    qa.load_gauge(np.zeros((4, V * 18)), gp)
qa.init(0)
if case == "sentinel_gauge_param":
    gp = qa.lib().newQudaGaugeParam()     # every field still "invalid"
    gp.X[0] = gp.X[1] = gp.X[2] = gp.X[3] = 4
    qa.load_gauge(np.zeros((4, V * 18)), gp)
elif case == "dslash_without_gauge":
    ip = qa.invert_param(qa.QUDA_TWISTED_MASS_DSLASH, 0.1, 0.01)
    qa.dslash(np.zeros(V // 2 * 24), ip, 0)
elif case == "unsupported_dslash_type":
    unit = np.zeros((4, V, 9, 2))
    unit[:, :, [0, 4, 8], 0] = 1
    qa.load_gauge(unit.reshape(4, -1), qa.gauge_param(X))
    ip = qa.invert_param(qa.QUDA_TWISTED_MASS_DSLASH, 0.1, 0.01)
    ip.dslash_type = 5   # QUDA_DOMAIN_WALL_DSLASH: not on this library's path
    qa.dslash(np.ones(V // 2 * 24), ip, 0)
elif case == "clover_without_coefficient":
    unit = np.zeros((4, V, 9, 2))
    unit[:, :, [0, 4, 8], 0] = 1
    qa.load_gauge(unit.reshape(4, -1), qa.gauge_param(X))
    ip = qa.invert_param(qa.QUDA_TWISTED_CLOVER_DSLASH, 0.1, 0.01)
    ip.clover_coeff = 0.0
    qa.load_clover(None, None, ip)
elif case == "mg_outer_pc_with_full_smoother":
    sys.path.insert(0, ROOT)
    from synth import smooth_gauge
    Xm = (8, 8, 8, 8)
    qa.load_gauge(smooth_gauge(Xm, 0.35), qa.gauge_param(Xm, cuda_prec=8, prec_sloppy=4, prec_precondition=4, t_boundary=qa.QUDA_PERIODIC_T))
    ip = qa.invert_param(qa.QUDA_TWISTED_MASS_DSLASH, 0.124, 0.005, +1, "ee", 0, cuda_prec=8, prec_sloppy=4, prec_precondition=4, solution_type=qa.QUDA_MAT_SOLUTION)
    ip.solve_type, ip.inv_type, ip.gcrNkrylov, ip.tol, ip.maxiter = qa.QUDA_DIRECT_SOLVE, qa.QUDA_GCR_INVERTER, 20, 1e-10, 100
    mg = qa.Multigrid(qa.multigrid_param(ip, n_level=2, geo_block=(4, 4, 4, 4), n_vec=4, setup_maxiter=20, setup_tol=1e-2, smoother_pc=False))
    ip.solve_type = qa.QUDA_DIRECT_PC_SOLVE
    ip.inv_type_precondition, ip.preconditioner = qa.QUDA_MG_INVERTER, mg.h
    ip.tol_precondition, ip.maxiter_precondition, ip.precondition_cycle, ip.omega = 1e-1, 1, 1, 1.0
    qa.invert(np.ones(int(np.prod(Xm)) * 24), ip)
elif case in ("calcmg_not_pc", "calcmg_not_ukqcd"):
    unit = np.zeros((4, V, 9, 2))
    unit[:, :, [0, 4, 8], 0] = 1
    qa.load_gauge(unit.reshape(4, -1), qa.gauge_param(X, t_boundary=qa.QUDA_PERIODIC_T))
    ip = qa.invert_param(qa.QUDA_TWISTED_MASS_DSLASH, 0.1, 0.01, solution_type=qa.QUDA_MAT_SOLUTION,
                         gamma_basis=qa.QUDA_DEGRAND_ROSSI_GAMMA_BASIS if case == "calcmg_not_ukqcd" else qa.QUDA_UKQCD_GAMMA_BASIS)
    ip.solve_type = qa.QUDA_DIRECT_SOLVE if case == "calcmg_not_pc" else qa.QUDA_DIRECT_PC_SOLVE
    ip.inv_type, ip.gcrNkrylov, ip.tol, ip.maxiter = qa.QUDA_GCR_INVERTER, 10, 1e-6, 100
    qa.calc_mg_propagators(unit.reshape(4, -1), ip, (0, 0, 0, 0), 1, 0.5, V)
print("NOT REACHED: %s did not abort" % case)
