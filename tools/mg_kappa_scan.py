import importlib, sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
from synth import smooth_gauge
qa = importlib.import_module("quda-qkxtm-multigrid_amd")
qa.init(0)
X=(16,16,16,16)
eps = float(sys.argv[1]) if len(sys.argv) > 1 else 0.35
gauge = smooth_gauge(X, eps)
b = np.random.default_rng(5).random(int(np.prod(X)) * 24)
ks = [float(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0.125]
mu0 = float(sys.argv[3]) if len(sys.argv) > 3 else 0.003
for kappa, mu in [(k, mu0) for k in ks]:
    gp = qa.gauge_param(X, cuda_prec=8, prec_sloppy=4, prec_precondition=4, t_boundary=qa.QUDA_PERIODIC_T)
    qa.load_gauge(gauge, gp)
    ip = qa.invert_param(qa.QUDA_TWISTED_MASS_DSLASH, kappa, mu, +1, "ee", 0, cuda_prec=8, prec_sloppy=4, prec_precondition=4, solution_type=qa.QUDA_MAT_SOLUTION)
    ip.solve_type, ip.inv_type, ip.gcrNkrylov, ip.tol, ip.maxiter = qa.QUDA_DIRECT_SOLVE, qa.QUDA_GCR_INVERTER, 20, 1e-10, 3000
    ip.inv_type_precondition = qa.QUDA_INVALID_ENUM
    t0=time.perf_counter(); qa.invert(b, ip); tp=time.perf_counter()-t0; itp=ip.iter
    mp = qa.multigrid_param(ip, n_level=3, geo_block=[(4,4,4,4),(2,2,2,2),(2,2,2,2)], n_vec=24, setup_maxiter=500, setup_tol=5e-6, smoother_pc=True)
    mg = qa.Multigrid(mp)
    ip.inv_type_precondition = qa.QUDA_MG_INVERTER; ip.preconditioner = mg.h
    ip.tol_precondition, ip.maxiter_precondition, ip.precondition_cycle, ip.omega = 1e-1, 1, 1, 1.0
    qa.invert(b, ip)
    t0=time.perf_counter(); x=qa.invert(b, ip); tm=time.perf_counter()-t0
    res=float(np.linalg.norm(b - qa.mat(x, ip))/np.linalg.norm(b))
    print("kappa %.4f mu %.4f: plain GCR %d its %.3f s | MG-GCR %d its %.3f s (setup %.2f s) res %.1e" % (kappa, mu, itp, tp, ip.iter, tm, mp.secs, res), flush=True)
    mg.free()
qa.end()
