#!/usr/bin/env python3
"""bench.py — the reference's headline benchmark on MI355X: even-odd twisted-mass Dslash throughput on a 32^4 lattice.

A "step" is ONE application of DiracTwistedMassPC::Dslash (stencil + fused inverse twist) on device-resident
fields, exactly what tests/dslash_test.cpp times with transfer=0 (:455-616).  `value` is GFLOP/s with the
reference's kernel-level flop count (1368 / checkerboard site, lib/dslash_twisted_mass.cu:144-160).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--prec 8|4|2] [--recon 18|12] [--dslash tm|tmc|wilson] [--lattice X,Y,Z,T]

N > 1 (launched by torch.distributed.run): the global lattice is 4-D block-decomposed over the ranks (strong scaling: the
global volume is fixed); the halo goes through direct peer stores over xGMI where the start-up probe and the first-use check
allow it, else through grouped RCCL send/recv.  After the Dslash measurement the ranks run the MG-GCR half of the metric on the
decomposed lattice (extra.mg_gcr; --no-mg skips it).  N = 1 adds the other precisions / actions, the 48^3 x 96 legs, the partitioned
sub-lattice kernel of an 8-GPU split, the MG-GCR legs (32^4, its critical kappa, 16^4, 48^3 x 96) and the CPU baselines under `extra`.

One JSON line on stdout; `roofline` is computed from the ALGORITHMIC bytes of the stencil kernel
divided by its average duration measured with HIP events on the stream the kernel runs on; `cpu_baseline` is
the oracle (CPU restatement of the reference host path, kind "port") timed on this box's host cores.
"""
import argparse
import importlib
import json
import os
import sys
import time

# the CPU-baseline legs run OpenMP loops with many short parallel regions (BLAS-1 of the host GCR): on a box whose CPU share is a
# cgroup quota, spinning at the barriers eats the quota and the solve took 229 s instead of 2 s — make idle threads sleep
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
os.environ.setdefault("GOMP_SPINCOUNT", "0")

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md


from synth import make_clover, make_gauge, smooth_gauge, tiled_gauge  # noqa: E402


class _LineGuard:
    """Keeps the finished JSON line (rank 0) safe while an optional leg runs: a library error inside the leg writes it before the
    process ends (qudaAmdSetExitLine), and a watchdog thread writes it if the leg does not come back within `timeout` seconds (ctypes
    calls release the GIL).  Either way the process ends with a NON-ZERO status (3): the line is delivered, but a failure inside a
    process that has touched the GPU is never reported to the launcher as success."""

    def __init__(self, qa, line, timeout):
        import threading
        self.qa, self.line, self.done = qa, line, False
        text = None
        if line is not None:
            failed = dict(line)
            failed["extra"] = dict(line.get("extra") or {}, mg_gcr=dict(failed="the library ended the process inside this leg (see stderr)"))
            text = json.dumps(failed).encode()
        qa.lib().qudaAmdSetExitLine(text if text is not None else b"", 3)
        self._timer = threading.Timer(timeout, self._timeout, [timeout])
        self._timer.daemon = True
        self._timer.start()

    def _timeout(self, timeout):
        if self.done:
            return
        self.done = True
        if self.line is not None:
            self.line.setdefault("extra", {})["mg_gcr"] = dict(failed="no result after %.0f s" % timeout)
            sys.stdout.write(json.dumps(self.line) + "\n")
            sys.stdout.flush()
        os._exit(3)

    def disarm(self):
        self.done = True
        self._timer.cancel()
        self.qa.lib().qudaAmdSetExitLine(None, 1)


def run_mg_ranks(qa, dist, X, kappa=0.124, mu=0.005):
    """MG-GCR to 1e-10 on the lattice decomposed over the ranks of `dist` (N > 1 leg of the metric): same field, source, solver and
    hierarchy shape as run_mg at N = 1; seconds are the slowest rank's, the residual is the global |b - M x| / |b| from MatQuda."""
    import ctypes as C
    import multi_gpu as mgpu
    hook = os.environ.get("QUDA_AMD_BENCH_MG_TEST")   # rehearsal of the line guard: a library error / a leg that never returns
    if hook == "error" and dist.rank == dist.world - 1:
        qa.lib().qudaAmdSetDslashTune(b"no-such-key", 0)
    if hook == "hang":
        time.sleep(1e6)
    Xl = dist.local_dims
    gauge = dist.scatter_gauge(smooth_gauge(tuple(X), 0.35))
    qa.lib().freeCloverQuda()
    qa.load_gauge(gauge, qa.gauge_param(Xl, cuda_prec=8, prec_sloppy=4, prec_precondition=4, t_boundary=qa.QUDA_PERIODIC_T))
    del gauge
    ip = qa.invert_param(qa.QUDA_TWISTED_MASS_DSLASH, kappa, mu, +1, "ee", 0, cuda_prec=8, prec_sloppy=4, prec_precondition=4, solution_type=qa.QUDA_MAT_SOLUTION)
    ip.solve_type, ip.inv_type, ip.gcrNkrylov, ip.tol, ip.maxiter, ip.reliable_delta = qa.QUDA_DIRECT_SOLVE, qa.QUDA_GCR_INVERTER, 20, 1e-10, 5000, 1e-4
    b = mgpu.scatter_field(np.random.default_rng(5).random(int(np.prod(X)) * 24), X, dist.grid, dist.coords, 24)

    xbuf = np.zeros_like(b)   # the caller's solution array, touched (see run_mg)
    xbuf.fill(0.0)

    def global_res(x):
        r = b - qa.mat(x, ip)
        n2 = np.array([np.dot(r, r), np.dot(b, b)])
        qa.lib().qudaAmdCommAllreduce(n2.ctypes.data_as(C.POINTER(C.c_double)), 2)
        return float(np.sqrt(n2[0] / n2[1]))

    def timed_solve():
        qa.invert(b, ip, out=xbuf)
        best = None
        for _ in range(2):
            dist.barrier()
            t0 = time.perf_counter()
            x = qa.invert(b, ip, out=xbuf)
            wall = dist.max_over_ranks(time.perf_counter() - t0)
            if best is None or wall < best[0]:
                best = (wall, dist.max_over_ranks(ip.secs), ip.iter, x)
        return best

    ip.inv_type_precondition = qa.QUDA_INVALID_ENUM
    wall, inner, iters, xp = timed_solve()
    plain = dict(iters=iters, secs=round(wall, 4), solver_secs=round(inner, 4), true_res=global_res(xp))
    # second-level aggregates by the reference's rule on the LOCAL coarse extents: the largest of 4 / 2 that leaves an even extent
    lvl1 = [x // 4 for x in Xl]
    b1 = tuple(2 if (x % 2 == 0 and (x // 2) % 2 == 0) else 1 for x in lvl1)
    levels = 3 if all(x % 4 == 0 for x in Xl) and all(v == 2 for v in b1) else 2
    dist.barrier()
    t0 = time.perf_counter()
    mp = qa.multigrid_param(ip, n_level=levels, geo_block=[(4, 4, 4, 4), (2, 2, 2, 2), (2, 2, 2, 2)][:levels], n_vec=24, setup_maxiter=500, setup_tol=5e-6, smoother_pc=True,
                            cycle=qa.QUDA_MG_CYCLE_VCYCLE)   # the plain V-cycle of BASELINE.json configs[4], as extra.mg_gcr at N = 1
    mg = qa.Multigrid(mp)
    setup = dist.max_over_ranks(time.perf_counter() - t0)
    lv0 = mg.level_info(0)
    ip.inv_type_precondition = qa.QUDA_MG_INVERTER
    ip.preconditioner = mg.h
    ip.tol_precondition, ip.maxiter_precondition, ip.precondition_cycle, ip.omega = 1e-1, 1, 1, 1.0
    wall, inner, iters, x = timed_solve()
    out = dict(lattice="x".join(map(str, X)), local_lattice="x".join(map(str, Xl)), process_grid=list(dist.grid), kappa=kappa, mu=mu, levels=levels, n_vec=24, cycle="V-cycle",
               null_vectors_level0={0: "sequential BiCGstab solves", 1: "lockstep block BiCGstab on the multi-rhs stencil (ghost zones behind the block fields)"}.get(lv0["null_method"], "?"),
               setup_secs=round(setup, 3), solve_secs=round(wall, 4), solver_secs=round(inner, 4), iters=iters, true_res=global_res(x), plain_gcr=plain,
               timing="slowest rank, best of 2 after 1 warm-up solve; residual = global |b - M x| / |b| through MatQuda")
    mg.free()
    return out


def run_mg(qa, X=(16, 16, 16, 16), blocks=((4, 4, 4, 4), (2, 2, 2, 2), (2, 2, 2, 2)), gauge=None, extras=True, kappa=0.124, mu=0.005, plain_maxiter=5000,
           coarse_bench=True, setup_repeats=1, dslash="tm", csw=1.57551, cycle="V", refine=0, recon_sloppy=None, recon_precondition=None, multi_src=0):
    """MG-preconditioned GCR to |r|/|b| <= 1e-10 (the second half of the metric) on one GPU: 3-level K-cycle, 24 null
    vectors, 4^4 then 2^4 aggregates, even-odd preconditioned MR smoother — the reference harness' shape (tests/multigrid_invert_test.cpp:224-286)
    with the plain V-cycle BASELINE.json configs[4] names (cycle="K": the harness' default K-cycle, reported next to it) on a smooth synthetic gauge field (synth.smooth_gauge: far easier than a production
    configuration — plain GCR needs only 76 iterations — so the MG / plain ratio here understates what MG buys at the physical point).  Setup (null
    vectors + Galerkin operators) and solve are timed separately; the residual is re-computed with MatQuda."""
    qa.lib().freeCloverQuda()
    if gauge is None:
        gauge = smooth_gauge(X, 0.35)
    gp = qa.gauge_param(X, cuda_prec=8, prec_sloppy=4, prec_precondition=4, t_boundary=qa.QUDA_PERIODIC_T, recon_sloppy=recon_sloppy, recon_precondition=recon_precondition)   # recon_sloppy: also the preconditioner links' unless given
    qa.load_gauge(gauge, gp)
    del gauge
    ip = qa.invert_param(qa.QUDA_TWISTED_CLOVER_DSLASH if dslash == "tmc" else qa.QUDA_TWISTED_MASS_DSLASH, kappa, mu, +1, "ee", 0, cuda_prec=8, prec_sloppy=4,
                         prec_precondition=4, solution_type=qa.QUDA_MAT_SOLUTION)
    ip.solve_type = qa.QUDA_DIRECT_SOLVE
    ip.inv_type = qa.QUDA_GCR_INVERTER
    ip.gcrNkrylov = 20
    ip.tol = 1e-10
    ip.maxiter = plain_maxiter
    if dslash == "tmc":
        # the production ETMC action the way the QKXTM drivers set it up (reference qkxtm/CalcMG_2pt3pt_EvenOdd.cpp:222-240): the clover
        # term is built on the device from the resident links, loadCloverQuda(NULL, NULL), clover_coeff = kappa csw
        ip.clover_coeff = kappa * csw
        qa.load_clover(None, None, ip)
    b = np.random.default_rng(5).random(int(np.prod(X)) * 24)
    # the caller's solution array, existing and touched as a C caller's would be: a fresh numpy array per call is untouched memory whose page
    # faults (0.1 s for 2 GB at 48^3 x 96) land in the download of the solution and would be timed as part of invertQuda
    xbuf = np.zeros_like(b)
    xbuf.fill(0.0)

    def timed_solve():
        """best of three after one warm-up: (wall seconds of invertQuda, GCR-loop seconds, iterations, solution).  The first solve
        through a new path pays one-off lazy initialisation, and single calls on the shared test boxes occasionally stall
        for 50-100 ms on the host side (seen inside otherwise 0.1 ms stages), hence the minimum."""
        qa.invert(b, ip, out=xbuf)
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            x = qa.invert(b, ip, out=xbuf)
            wall = time.perf_counter() - t0
            if best is None or wall < best[0]:
                best = (wall, ip.secs, ip.iter, x)
        return best

    ip.inv_type_precondition = qa.QUDA_INVALID_ENUM
    if plain_maxiter > 5000:   # thousands of iterations: one solve, not best-of-three
        t0 = time.perf_counter(); xp = qa.invert(b, ip); wall = time.perf_counter() - t0
        inner, iters = ip.secs, ip.iter
    else:
        wall, inner, iters, xp = timed_solve()
    plain = dict(iters=iters, secs=round(wall, 4), solver_secs=round(inner, 4), true_res=float(np.linalg.norm(b - qa.mat(xp, ip)) / np.linalg.norm(b)))
    ip.maxiter = 5000
    # the hierarchy is built `setup_repeats` times and the fastest build is reported (all of them listed): the first seconds of a
    # process on a fresh box pay host-side stalls (the image still paging in) that have nothing to do with the set-up
    setups = []
    for rep in range(setup_repeats):
        if rep:
            mg.free()
        mp = qa.multigrid_param(ip, n_level=3, geo_block=[tuple(bk) for bk in blocks], n_vec=24, setup_maxiter=500, setup_tol=5e-6, smoother_pc=True,
                                cycle=qa.QUDA_MG_CYCLE_VCYCLE if cycle == "V" else qa.QUDA_MG_CYCLE_RECURSIVE)
        mg = qa.Multigrid(mp)
        setups.append(round(mp.secs, 3))
    ip.inv_type_precondition = qa.QUDA_MG_INVERTER
    ip.preconditioner = mg.h
    ip.tol_precondition, ip.maxiter_precondition, ip.precondition_cycle, ip.omega = 1e-1, 1, 1, 1.0
    wall, inner, iters, x = timed_solve()
    res = float(np.linalg.norm(b - qa.mat(x, ip)) / np.linalg.norm(b))
    refined = None
    if refine:
        # set-up refinement (not in the reference): inverse iteration of the null vectors through the hierarchy, hierarchy rebuilt; the
        # numbers of the plain set-up stay in the line next to the refined ones
        refined = dict(plain_setup=dict(iters=iters, solve_secs=round(wall, 4), solver_secs=round(inner, 4), true_res=res), passes=refine)
        refined["refine_secs"] = round(mg.refine(refine, 1), 3)
        wall, inner, iters, x = timed_solve()
        res = float(np.linalg.norm(b - qa.mat(x, ip)) / np.linalg.norm(b))
    # solve_secs: wall clock of invertQuda (host source in, host solution out, as SURVEY 8d defines it: includes the two
    # PCIe transfers and the operator / field set-up); solver_secs: the GCR loop alone (QudaInvertParam.secs)
    lv0, lv1 = mg.level_info(0), mg.level_info(1)
    out = dict(lattice="x".join(map(str, X)), action={"tm": "twisted mass", "tmc": "twisted clover (device-built clover, csw %g)" % csw}[dslash], kappa=kappa, mu=mu, levels=3, n_vec=24, blocks=[list(bk) for bk in blocks[:2]],
               cycle={"V": "V-cycle (QUDA_MG_CYCLE_VCYCLE)", "K": "K-cycle (QUDA_MG_CYCLE_RECURSIVE)"}[cycle], null_vectors=dict(level0={0: "preset (refined from the first hierarchy)" if refined else "sequential BiCGstab solves", 1: "lockstep block BiCGstab on the multi-rhs stencil", 2: "lockstep block BiCGstab on the MFMA coarse operator"}[lv0["null_method"]],
                                 level0_lockstep_iters=lv0["null_iters"], level1={0: "sequential BiCGstab solves", 1: "lockstep (fine stencil)", 2: "lockstep block BiCGstab on the MFMA coarse operator"}[lv1["null_method"]],
                                 level1_lockstep_iters=lv1["null_iters"]),
               setup_secs=min(setups), setup_secs_all=setups, solve_secs=round(wall, 4),
               solver_secs=round(inner, 4), iters=iters, true_res=res, plain_gcr=plain, timing="best of 3 after 1 warm-up solve")
    if refined:
        out["setup_refinement"] = refined
    # the multi-right-hand-side coarse operator on the matrix cores (level 1: 2 Nvec = 48 rows, 9 dense matrices per site) against
    # the single-vector kernel: seconds per application, HBM rate on the ALGORITHMIC bytes (links once + in/out panels) and MFMA rate
    try:
        if not coarse_bench:
            raise RuntimeError("not requested for this leg")
        info = mg.level_info(0)
        Vc, nn = int(np.prod(info["Xc"])), 2 * info["Nvec"]
        single = mg.time_apply(1, 20)
        cb = dict(coarse_lattice="x".join(map(str, info["Xc"])), n=nn, single_vector_us=round(1e6 * single, 1),
                  single_vector_hbm_gbs=round(Vc * (9 * nn * nn * 8 + 10 * nn * 8) / single * 1e-9, 1))
        rng = np.random.default_rng(2)
        for nrhs in (8, 16, 24):
            eta = (rng.standard_normal((nrhs, Vc, 2, info["Nvec"])) + 1j * rng.standard_normal((nrhs, Vc, 2, info["Nvec"]))).astype(np.complex64)
            _, secs = mg.apply_block(1, eta, niter=20)
            nbytes = Vc * (9 * nn * nn * 8 + 2 * nn * nrhs * 8)
            flops = Vc * 9 * 8 * nn * nn * nrhs
            cb["nrhs_%d" % nrhs] = dict(us=round(1e6 * secs, 1), us_per_rhs=round(1e6 * secs / nrhs, 2), hbm_gbs=round(nbytes / secs * 1e-9, 1),
                                        hbm_frac=round(nbytes / secs * 1e-9 / HBM_PEAK_GBS, 4), mfma_tflops=round(flops / secs * 1e-12, 2),
                                        mfma_frac_of_157=round(flops / secs * 1e-12 / 157.3, 4), speedup_vs_single=round(single * nrhs / secs, 2))
        cb["bound"] = "hbm (links read once per site: AI = nrhs flop/B; fp32 MFMA peak 157 TFLOP/s is reached only near nrhs = 24; at 24 the matrix pipe is busy 75 % of all cycles at the 1.95 GHz the device sustains under matrix load: profiles/r03_coarse_block_kernel_pmc.log)"
        out["coarse_block_mfma"] = cb
    except Exception as e:  # e.g. a hierarchy whose level 1 does not qualify (n not a multiple of 16)
        out["coarse_block_mfma"] = dict(skipped=str(e)[:200])
    if not extras:
        mg.free()
        return out
    # the QKXTM production shape: the same hierarchy under an outer GCR on the even-odd preconditioned system
    # (solve_type = QUDA_DIRECT_PC_SOLVE, reference lib/interface_quda.cpp:6041), full-field solution via prepare / reconstruct
    ip.solve_type = qa.QUDA_DIRECT_PC_SOLVE
    wall, inner, iters, x = timed_solve()
    ip.solve_type = qa.QUDA_DIRECT_SOLVE
    out["outer_even_odd"] = dict(solve_secs=round(wall, 4), solver_secs=round(inner, 4), iters=iters,
                                 true_res=float(np.linalg.norm(b - qa.mat(x, ip)) / np.linalg.norm(b)))
    # a propagator's worth of sources through ONE lockstep solve (invertMultiSrcQuda, csrc/block_solver.cpp) against the same sources through
    # invertQuda one after the other: even-odd outer solve as the QKXTM drivers run it, best of 3 passes each, solver seconds (the GCR loops) and
    # wall seconds (with the host transfers) side by side
    try:
        nsrc = multi_src
        if not nsrc:
            raise RuntimeError("not requested for this leg")
        rng = np.random.default_rng(7)
        bs = [rng.random(int(np.prod(X)) * 24) for _ in range(nsrc)]
        xs = [np.zeros_like(v) for v in bs]
        ip.solve_type = qa.QUDA_DIRECT_PC_SOLVE
        qa.invert_multi_src(bs, ip, out=xs)
        seq, blk = [], []
        for _ in range(3):
            sw, ss, its = 0.0, 0.0, []
            for v in bs:
                t0 = time.perf_counter(); qa.invert(v, ip, out=xbuf); sw += time.perf_counter() - t0
                ss += ip.secs; its.append(ip.iter)
            seq.append((ss, sw))
            t0 = time.perf_counter(); qa.invert_multi_src(bs, ip, out=xs); bw = time.perf_counter() - t0
            blk.append((ip.secs, bw))
        blk_iters = ip.iter
        ip.solve_type = qa.QUDA_DIRECT_SOLVE
        worst = max(float(np.linalg.norm(v - qa.mat(x, ip)) / np.linalg.norm(v)) for x, v in zip(xs, bs))
        st = qa.multi_src_stats()
        out["multi_src"] = dict(sources=nsrc, outer="even-odd (QUDA_DIRECT_PC_SOLVE)", sequential=dict(solver_secs=round(min(seq)[0], 4), wall_secs=round(min(seq)[1], 4), iters=its),
                                lockstep=dict(solver_secs=round(min(blk)[0], 4), wall_secs=round(min(blk)[1], 4), iters=blk_iters, worst_true_res=worst,
                                              fine_smoother="block fields (multi-rhs stencil)" if st["block_smoothed"] else "per source", quad_transfers=st["quad_transfers"] > 0),
                                speedup_solver=round(min(seq)[0] / min(blk)[0], 3), speedup_wall=round(min(seq)[1] / min(blk)[1], 3))
        del bs, xs
    except Exception as e:
        ip.solve_type = qa.QUDA_DIRECT_SOLVE
        out["multi_src"] = dict(skipped=str(e)[:200])
    # opt-in half-precision storage inside the cycle: fp16 V and coarse links, 16-bit level-0 smoother (the outer solve is unchanged)
    mg.set_half_storage(True)
    wall, inner, iters, x = timed_solve()
    out["half_precision_cycle"] = dict(solve_secs=round(wall, 4), solver_secs=round(inner, 4), iters=iters,
                                      true_res=float(np.linalg.norm(b - qa.mat(x, ip)) / np.linalg.norm(b)))
    mg.set_half_storage(False)
    mg.free()
    # the other cycle type on the same problem (one more set-up)
    other = "K" if cycle == "V" else "V"
    mp = qa.multigrid_param(ip, n_level=3, geo_block=[tuple(bk) for bk in blocks], n_vec=24, setup_maxiter=500, setup_tol=5e-6, smoother_pc=True,
                            cycle=qa.QUDA_MG_CYCLE_VCYCLE if other == "V" else qa.QUDA_MG_CYCLE_RECURSIVE)
    mg = qa.Multigrid(mp)
    ip.preconditioner = mg.h
    wall, inner, iters, x = timed_solve()
    out["%s_cycle" % other.lower()] = dict(solve_secs=round(wall, 4), solver_secs=round(inner, 4), iters=iters, true_res=float(np.linalg.norm(b - qa.mat(x, ip)) / np.linalg.norm(b)))
    mg.free()
    return out


def launch_ranks(n, argv):
    """`python bench.py --gpus N` started plainly (no torch.distributed.run): this parent, which never touches the GPU, starts N
    fresh children of this same script with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set (one process per GPU),
    relays rank 0's stdout (the JSON line) and everybody's stderr, and returns the first non-zero status — after which the
    remaining ranks are ended, since a partner that will never arrive only leaves them waiting in a collective."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    status, pending = 0, set(range(n))
    while pending:
        for r in sorted(pending):
            rc = procs[r].poll()
            if rc is None:
                continue
            pending.discard(r)
            if rc != 0 and status == 0:
                status = rc if rc > 0 else 1
                sys.stderr.write("bench.py: rank %d ended with status %d, ending the other ranks\n" % (r, rc))
                deadline = time.time() + 20      # rank 0 may still be writing its guarded line
                while time.time() < deadline and any(procs[q].poll() is None for q in pending):
                    time.sleep(0.2)
                for q in pending:
                    if procs[q].poll() is None:
                        procs[q].terminate()
        time.sleep(0.1)
    return status


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--prec", type=int, default=8, choices=[8, 4, 2])
    ap.add_argument("--recon", type=int, default=18, choices=[18, 12, 8])
    ap.add_argument("--dslash", default="tm", choices=["tm", "tmc", "wilson"])
    ap.add_argument("--lattice", default="32,32,32,32")
    ap.add_argument("--fast-gauge", action="store_true", help="links = periodic repetition of 65536 random SU(3) matrices (profiling of big lattices)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-extra", action="store_true", help="skip the other-precision sweep")
    ap.add_argument("--no-mg", action="store_true", help="N > 1: skip the MG-GCR leg on the decomposed lattice")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch plainly (python bench.py --gpus N) or with torch.distributed.run --nproc-per-node N" % (args.gpus, world))

    qa = importlib.import_module("quda-qkxtm-multigrid_amd")
    X = [int(v) for v in args.lattice.split(",")]
    kappa, mu = 0.1, 0.01  # tests/dslash_test.cpp:124-129

    dist = None
    if world > 1:
        import multi_gpu  # repo-root helper: process grid, RCCL bootstrap through torch.distributed
        dist = multi_gpu.setup(qa, rank, world, local_rank, X)
        Xl = dist.local_dims
    else:
        qa.init(0)
        Xl = X
    Vh_local = int(np.prod(Xl)) // 2
    Vh_global = int(np.prod(X)) // 2

    dtype_name = {8: "f64", 4: "f32", 2: "i16+f32scale"}
    kinds = {"tm": qa.QUDA_TWISTED_MASS_DSLASH, "tmc": qa.QUDA_TWISTED_CLOVER_DSLASH, "wilson": qa.QUDA_WILSON_DSLASH}

    full_gauge = tiled_gauge(X) if args.fast_gauge else make_gauge(X)
    gauge = full_gauge if dist is None else dist.scatter_gauge(full_gauge)
    del full_gauge
    clover = None
    rng = np.random.default_rng(1234 + rank)
    src_h = rng.random(Vh_local * 24)

    def run(prec, recon, kind, steps, warmup, prewarm=0.0):
        gp = qa.gauge_param(Xl, cuda_prec=prec, recon=recon)
        qa.load_gauge(gauge, gp)
        ip = qa.invert_param(kinds[kind], kappa, mu, +1, "ee", 0, cuda_prec=prec)
        if kind == "tmc":
            nonlocal clover
            if clover is None:
                clover = make_clover(Xl, seed=11 + rank)
            qa.load_clover(clover, None, ip)
        src, dst = qa.Spinor(prec), qa.Spinor(prec)
        src.load(src_h, ip)
        d = qa.Dirac(ip, pc=True)
        # untimed: bring the device to its steady clocks first — the driver's `--steps 20 --warmup 5` is a 3 ms measurement right after
        # the process started, and a GPU that idled through the host-side set-up needs tens of ms to ramp up.  Then the W warm-up
        # steps of the contract, then exactly K timed ones.
        n_prewarm = 0
        if prewarm > 0:
            # the NUMBER of applications has to be the same on every rank (a partitioned application pairs with its neighbours'):
            # time 100 of them, agree on the count through the slowest rank, then run that many
            t_pre = time.perf_counter()
            d.time_dslash(dst, src, 0, 100)
            t100 = time.perf_counter() - t_pre
            n_pre = int(min(20000, max(0, (prewarm - t100) / max(t100, 1e-6) * 100)))
            if dist is not None:
                n_pre = int(dist.max_over_ranks(float(n_pre)))
            if n_pre > 0:
                d.time_dslash(dst, src, 0, n_pre)
            n_prewarm = 100 + n_pre
        d.time_dslash(dst, src, 0, max(1, warmup))
        if dist is not None:
            dist.barrier()
        qa.lib().qudaAmdDeviceSynchronize()
        t0 = time.perf_counter()
        sec_kernel = d.time_dslash(dst, src, 0, steps)  # HIP events on the compute stream, per application
        qa.lib().qudaAmdDeviceSynchronize()
        if dist is not None:
            dist.barrier()
        wall = time.perf_counter() - t0
        sec_min = sec_kernel
        if dist is not None:
            wall = dist.max_over_ranks(wall)
            sec_min = -dist.max_over_ranks(-sec_kernel)   # fastest and slowest rank: the spread shows an uneven halo wait
            sec_kernel = dist.max_over_ranks(sec_kernel)
        flops_site = qa.lib().qudaAmdDslashFlopsPerSite(ip, 0)
        bytes_site = qa.lib().qudaAmdDslashBytesPerSite(ip, 0, 0)
        n2 = dst.norm2()
        for f in (src, dst):
            f.free()
        d.free()
        return dict(wall=wall, sec=sec_kernel, sec_min=sec_min, flops_site=flops_site, bytes_site=bytes_site, norm2=n2, prewarm=n_prewarm)

    r = run(args.prec, args.recon, args.dslash, args.steps, args.warmup, prewarm=float(os.environ.get("QUDA_AMD_BENCH_PREWARM", "0.3")))
    ms_per_step = 1e3 * r["wall"] / args.steps
    gflops = r["flops_site"] * Vh_global / (r["wall"] / args.steps) * 1e-9
    achieved = r["bytes_site"] * Vh_local / r["sec"] * 1e-9  # GB/s of ONE GPU's kernel (per launch)

    extra = {}
    if not args.no_extra and rank == 0 and world == 1:
        for prec, recon, kind in ((8, 12, "tm"), (4, 18, "tm"), (4, 12, "tm"), (4, 8, "tm"), (2, 18, "tm"), (2, 8, "tm"), (8, 18, "tmc"), (4, 18, "tmc"), (2, 18, "tmc")):
            e = run(prec, recon, kind, max(20, args.steps // 2), 5)
            extra["%s_%s_r%d" % (kind, dtype_name[prec].split("+")[0], recon)] = dict(
                gflops=round(e["flops_site"] * Vh_global / e["sec"] * 1e-9, 1), hbm_gbs=round(e["bytes_site"] * Vh_local / e["sec"] * 1e-9, 1),
                frac=round(e["bytes_site"] * Vh_local / e["sec"] * 1e-9 / HBM_PEAK_GBS, 4), bytes_per_site=e["bytes_site"], us=round(1e6 * e["sec"], 2))

        # streaming yardstick: y += a x on full-lattice fp64 fields (3 x 201 MB per call, beyond the 256 MB Infinity Cache)
        sx, sy = qa.Spinor(8, qa.QUDA_FULL_SITE_SUBSET), qa.Spinor(8, qa.QUDA_FULL_SITE_SUBSET)
        qa.lib().qudaAmdTimeAxpy(0.5, sx.h, sy.h, 5)
        sec = qa.lib().qudaAmdTimeAxpy(0.5, sx.h, sy.h, 50)
        extra["stream_axpy_f64"] = dict(hbm_gbs=round(3 * 2 * Vh_local * 24 * 8 / sec * 1e-9, 1), us=round(1e6 * sec, 2))
        sx.free(); sy.free()

        # the halo path on one GPU: the local lattice of an 8-GPU strong-scaling split of the 32^4 problem (1 x 2 x 2 x 2 -> 32 x 16 x 16 x 16)
        # with y, z, t partitioned through the self-neighbour emulation (every face goes through the ghost window and back, as the
        # reference's --partition flag does): one fused launch [pack blocks | sites], against the same lattice unpartitioned
        if X == [32, 32, 32, 32]:
            Xs = [32, 16, 16, 16]
            gs = make_gauge(Xs)
            hs = np.random.default_rng(1).random(int(np.prod(Xs)) // 2 * 24)
            halo = dict(local_lattice="x".join(map(str, Xs)), partitioned="y,z,t (self-neighbour emulation)")
            for prec in (8, 4, 2):
                row = {}
                # the partitioned kernel in both wire formats of the peer-store ghost zones: flag-in-data {word, flag, word, flag} vectors and the
                # compact self-validating 16-byte atoms {3 words, flag} = one 128-byte line per fp64 face site (QUDA_AMD_HALO_FORMAT=atom, the
                # default between devices): 2/3 of the bytes on the links
                for mask, fmt, name in ((0, 0, "unpartitioned_us"), (0b1110, 0, "partitioned_us"), (0b1110, 1, "partitioned_atom_format_us")):
                    qa.lib().qudaAmdSetPartitionMask(mask)
                    qa.lib().qudaAmdSetDslashTune(b"halo_format", fmt)
                    qa.load_gauge(gs, qa.gauge_param(Xs, cuda_prec=prec))
                    ip = qa.invert_param(qa.QUDA_TWISTED_MASS_DSLASH, kappa, mu, +1, "ee", 0, cuda_prec=prec)
                    src, dst = qa.Spinor(prec), qa.Spinor(prec)
                    src.load(hs, ip)
                    d = qa.Dirac(ip, pc=True)
                    d.time_dslash(dst, src, 0, 50)
                    row[name] = round(1e6 * min(d.time_dslash(dst, src, 0, 500) for _ in range(3)), 2)
                    if mask and not fmt:
                        row["transport"] = {1: "direct peer stores", 0: "RCCL send/recv", -1: "none"}[int(qa.lib().qudaAmdHaloTransport())]
                    src.free(); dst.free(); d.free()
                qa.lib().qudaAmdSetDslashTune(b"halo_format", -1)
                site_bytes = {8: (192, 128), 4: (96, 64), 2: (64, 48)}[prec]
                row["wire_bytes_per_face_site"] = {"flag_in_data": site_bytes[0], "atom_format": site_bytes[1], "payload": {8: 96, 4: 48, 2: 28}[prec]}
                row["y_face_bytes"] = {"flag_in_data": site_bytes[0] * 32 * 16 * 16 // 2, "atom_format": site_bytes[1] * 32 * 16 * 16 // 2}
                # what an 8-GPU strong-scaling run would make of it if xGMI behaved like the emulation: the 32^4 kernel of this precision on
                # one GPU over the partitioned sub-lattice kernel (8 = ideal)
                one = {8: 1e6 * r["sec"] if (args.prec, args.recon, args.dslash) == (8, 18, "tm") else None,
                       4: extra.get("tm_f32_r18", {}).get("us"), 2: extra.get("tm_i16_r18", {}).get("us")}[prec]
                if one:
                    row["one_gpu_32x4_us"] = round(one, 2)
                    row["projected_speedup_8_gpus"] = round(one / row["partitioned_us"], 2)
                    row["projected_speedup_8_gpus_atom_format"] = round(one / row["partitioned_atom_format_us"], 2)
                halo[dtype_name[prec].split("+")[0]] = row
            qa.lib().qudaAmdSetPartitionMask(0)
            extra["halo_8gpu_sublattice"] = halo

    g16 = None
    if not args.no_extra and rank == 0 and world == 1:
        g32 = smooth_gauge((32, 32, 32, 32), 0.35)
        extra["mg_gcr"] = run_mg(qa, (32, 32, 32, 32), gauge=g32, multi_src=12)
        # the same problem with the production action (twisted CLOVER): lockstep set-up on the clover variant of the multi-rhs stencil
        extra["mg_gcr_tmc"] = run_mg(qa, (32, 32, 32, 32), gauge=g32, dslash="tmc", coarse_bench=False)
        # where multigrid matters: the same field at its critical kappa (tools/mg_kappa_scan.py, profiles/r02_mg_kappa_scan_32x4_c.json:
        # plain GCR(20) needs > 10^4 iterations there and stagnates beyond it; the twisted mass keeps the operator regular)
        extra["mg_gcr_critical"] = run_mg(qa, (32, 32, 32, 32), gauge=g32, extras=False, kappa=0.147, mu=0.001, plain_maxiter=30000, coarse_bench=False, refine=3)
        del g32
        # the SAME small problem on the GPU and, below, on the host cores (cpu_baseline.solver): 16^4, same field family, kappa, mu
        g16 = smooth_gauge((16, 16, 16, 16), 0.35)
        extra["mg_gcr_16x4"] = run_mg(qa, (16, 16, 16, 16), gauge=g16, extras=False, coarse_bench=False)
        # BASELINE.json configs[4] (48^3 x 96, quoted by the reference on 8 GPUs) resident on this one GPU: 288 GB holds the whole
        # hierarchy; levels by the reference's blocking rule 48^3 x 96 -> 12^3 x 24 -> 6^4 (lib/transfer.cpp:31-44)
        from synth import smooth_gauge_cayley
        Xc5 = (48, 48, 48, 96)
        gc5 = smooth_gauge_cayley(Xc5, 0.35, workers=min(16, os.cpu_count() or 8))
        extra["mg_gcr_c5_one_gpu"] = run_mg(qa, Xc5, blocks=((4, 4, 4, 4), (2, 2, 2, 4), (2, 2, 2, 2)), gauge=gc5, setup_repeats=2, multi_src=12)
        # ... with the sloppy and preconditioner links stored as 12 reals (reconstruct_sloppy = reconstruct_precondition = 12, the usual production choice;
        # the precise links stay at 18): every fp32 stencil of the cycle moves 576 instead of 768 B per site
        r12 = run_mg(qa, Xc5, blocks=((4, 4, 4, 4), (2, 2, 2, 4), (2, 2, 2, 2)), gauge=gc5, extras=False, coarse_bench=False, recon_sloppy=qa.QUDA_RECONSTRUCT_12)
        extra["mg_gcr_c5_one_gpu"]["recon12_sloppy_links"] = {k: r12[k] for k in ("setup_secs", "solve_secs", "solver_secs", "iters", "true_res")}
        # ... and with 8-real sloppy links (reconstruct_sloppy = 8: the fp32 operator of the outer GCR moves 448 B per site; the hierarchy keeps 12)
        r8 = run_mg(qa, Xc5, blocks=((4, 4, 4, 4), (2, 2, 2, 4), (2, 2, 2, 2)), gauge=gc5, extras=False, coarse_bench=False, recon_sloppy=qa.QUDA_RECONSTRUCT_8, recon_precondition=qa.QUDA_RECONSTRUCT_12)
        extra["mg_gcr_c5_one_gpu"]["recon8_sloppy_links"] = {k: r8[k] for k in ("setup_secs", "solve_secs", "solver_secs", "iters", "true_res")}
        # ... and with the production action (twisted clover, clover term built on the device) at the production volume
        extra["mg_gcr_c5_tmc_one_gpu"] = run_mg(qa, Xc5, blocks=((4, 4, 4, 4), (2, 2, 2, 4), (2, 2, 2, 2)), gauge=gc5, dslash="tmc", extras=False, coarse_bench=False)
        # the stencil at that production volume (Vh = 5.3 M sites: one time slice is 10.6 MB, the fields no longer sit in the 256 MB
        # Infinity Cache as they do at 32^4), same byte model
        Vh5 = int(np.prod(Xc5)) // 2
        h5 = np.random.default_rng(3).random(Vh5 * 24)
        row5 = {}
        for prec, recon5 in ((8, 18), (4, 18), (2, 18), (8, 12), (4, 12), (4, 8), (2, 8)):
            qa.load_gauge(gc5, qa.gauge_param(list(Xc5), cuda_prec=prec, recon=recon5))
            ip = qa.invert_param(qa.QUDA_TWISTED_MASS_DSLASH, kappa, mu, +1, "ee", 0, cuda_prec=prec)
            src, dst = qa.Spinor(prec), qa.Spinor(prec)
            src.load(h5, ip)
            d = qa.Dirac(ip, pc=True)
            d.time_dslash(dst, src, 0, 5)
            sec = min(d.time_dslash(dst, src, 0, 30) for _ in range(2))
            bs = qa.lib().qudaAmdDslashBytesPerSite(ip, 0, 0)
            row5["tm_%s_r%d" % (dtype_name[prec].split("+")[0], recon5)] = dict(us=round(1e6 * sec, 1), hbm_gbs=round(bs * Vh5 / sec * 1e-9, 1), frac=round(bs * Vh5 / sec * 1e-9 / HBM_PEAK_GBS, 4),
                                                                    gflops=round(qa.lib().qudaAmdDslashFlopsPerSite(ip, 0) * Vh5 / sec * 1e-9, 1), bytes_per_site=bs)
            src.free(); dst.free(); d.free()
        extra["dslash_48x48x48x96"] = row5
        del gc5, h5

    cpu = None
    if rank == 0 and not args.no_cpu:
        import oracle_api  # TEST INFRASTRUCTURE used as the reported CPU baseline ("port" of the reference host path)
        oracle = oracle_api.load()
        cores = os.cpu_count() or 1
        try:
            cores = len(os.sched_getaffinity(0))
        except Exception:
            pass
        # a one-GPU box owns a 16-core share of the host (more threads only oversubscribe it)
        cores = int(os.environ.get("QUDA_AMD_CPU_THREADS", min(cores, 16)))
        Xc = Xl
        g_cpu = gauge if dist is None else gauge
        inp = src_h.copy()
        # protocol of SURVEY 8d / BASELINE.md 4.3: fields resident, >= 10 warm-up + >= 50 timed applications, wall clock.  All cores: the bench
        # workload itself (same lattice, same inputs).  One thread (the reference's host path has no threading): the same operator on a
        # 16^4 sample of the workload (60 applications of 32^4 would be a minute of one core; the loop nest is compute-bound, its rate does not
        # depend on the volume: 0.86 / 0.89 GFLOP/s at 8^4 / 16^4 in BASELINE.md 3)
        n_warm, n_timed = 10, 50
        oracle.set_threads(cores)
        for _ in range(n_warm):
            oracle.tm_dslash(g_cpu, inp, Xc, kappa, mu, +1, 0, "ee", 0)
        t0 = time.perf_counter()
        for _ in range(n_timed):
            oracle.tm_dslash(g_cpu, inp, Xc, kappa, mu, +1, 0, "ee", 0)
        tall = (time.perf_counter() - t0) / n_timed
        Xs = [min(16, v) for v in Xc]
        Vh_s = int(np.prod(Xs)) // 2
        g_s, inp_s = make_gauge(Xs), np.random.default_rng(77).random(Vh_s * 24)
        oracle.set_threads(1)
        for _ in range(n_warm):
            oracle.tm_dslash(g_s, inp_s, Xs, kappa, mu, +1, 0, "ee", 0)
        t0 = time.perf_counter()
        for _ in range(n_timed):
            oracle.tm_dslash(g_s, inp_s, Xs, kappa, mu, +1, 0, "ee", 0)
        t1 = (time.perf_counter() - t0) / n_timed
        cpu = dict(value=round(1368.0 * Vh_local / tall * 1e-9, 3), unit="GFLOP/s", cores=cores, kind="port",
                   sample="%d warm-up + %d timed tm_dslash fp64 on %s (oracle/liboracle.so, outer parallel-for over sites, %d threads); 1 thread, same protocol on a %s sample = %.3f GFLOP/s"
                   % (n_warm, n_timed, "x".join(str(v) for v in Xc), cores, "x".join(str(v) for v in Xs), 1368.0 * Vh_s / t1 * 1e-9),
                   one_thread=dict(value=round(1368.0 * Vh_s / t1 * 1e-9, 3), unit="GFLOP/s", cores=1, lattice="x".join(str(v) for v in Xs), warmup=n_warm, timed=n_timed),
                   warmup=n_warm, timed=n_timed)
        if g16 is not None:
            # the solver half of the metric on the host: the reference's restarted GCR(20) (lib/inv_gcr_quda.cpp, plainest configuration)
            # on the host tm_mat with lib/blas_cpu.cpp-style BLAS (oracle/qo_solver.c), fp64, on the SAME 16^4 problem (field, kappa, mu,
            # source, tolerance) as extra.mg_gcr_16x4 on the GPU — a size the host finishes in seconds (at 32^4 it took 244 s on 16 cores)
            bsol = np.random.default_rng(5).random(16 ** 4 * 24)
            cores_solver = min(cores, 8)   # 12.6 MB vectors: more threads only add barrier cost
            oracle.set_threads(cores_solver)
            _, it_cpu, secs_cpu, res_cpu = oracle.gcr_tm(g16, bsol, [16, 16, 16, 16], 0.124, 0.005, +1, tol=1e-10, nkrylov=20, maxiter=5000)
            g = extra.get("mg_gcr_16x4", {})
            cpu["solver"] = dict(what="plain GCR(20) to 1e-10 on tm_mat, fp64, 16x16x16x16, kappa 0.124 mu 0.005 (same problem as extra.mg_gcr_16x4)",
                                 secs=round(secs_cpu, 3), iters=it_cpu, true_res=res_cpu, cores=cores_solver,
                                 gpu_plain_gcr_secs=g.get("plain_gcr", {}).get("secs"), gpu_mg_gcr_secs=g.get("solve_secs"), gpu_mg_setup_secs=g.get("setup_secs"))
        oracle.set_threads(1)

    traffic, traffic_source = None, None
    if rank == 0 and world == 1:
        # HBM bytes per launch of the same kernel from the committed rocprofv3 PMC summary (tools/summarize_profiles.py)
        import glob
        tname = {8: "double", 4: "float", 2: "short"}[args.prec]
        variant = {"tm": 0, "wilson": 0, "tmc": 2}[args.dslash]
        key = "dslash_kernel<%s, %d, %d," % (tname, args.recon, variant)
        lat_tag = "_%s_" % ("32x4" if X == [32, 32, 32, 32] else "x".join(map(str, X)))
        for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_hbm_traffic.json"))):
            if lat_tag not in os.path.basename(path):   # the summaries are named after their lattice: never quote another volume's traffic
                continue
            try:
                for k, v in json.load(open(path))["kernels"].items():
                    if key in k and "hbm_bytes_per_launch" in v:
                        traffic, traffic_source = round(v["hbm_bytes_per_launch"]), os.path.basename(path)
            except Exception:
                pass
    def per_rank_report():
        """every rank's transport and exchange counters (a rank that fell back to the staged transport, or global sums that went through
        the collective library, are visible in the line): gathered with one sum all-reduce into rank-indexed slots"""
        stats = qa.comm_stats()
        mine = [float(qa.lib().qudaAmdHaloTransport())] + [float(stats[k]) for k in qa.COMM_STATS_KEYS]
        if dist is None:
            rows = [mine]
        else:
            import ctypes as C
            buf = np.zeros(world * len(mine))
            buf[rank * len(mine):(rank + 1) * len(mine)] = mine
            qa.lib().qudaAmdCommAllreduce(buf.ctypes.data_as(C.POINTER(C.c_double)), buf.size)
            rows = buf.reshape(world, len(mine)).tolist()
        names = {1: "peer stores", 0: "RCCL send/recv", -1: "none"}
        return [dict(rank=i, halo_transport=names.get(int(row[0]), "?"), **{k: int(v) for k, v in zip(qa.COMM_STATS_KEYS, row[1:])}) for i, row in enumerate(rows)]

    ranks_dslash = per_rank_report() if world > 1 else None
    if rank == 0:
        out = {
            "metric": "twisted-mass Dslash GFLOP/s (even-odd, 32^4)" if X == [32, 32, 32, 32] else "twisted-mass Dslash GFLOP/s",
            "value": round(gflops, 2), "unit": "GFLOP/s", "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
            "prewarm_applications": r["prewarm"],   # untimed, in front of the W contract warm-ups: brings the device to steady clocks
            "ms_per_step": round(ms_per_step, 5), "higher_is_better": True, "scaling": "strong",   # the global lattice is fixed (BASELINE: 32^4 on 1/2/4/8 GPUs), N ranks cut it into N sub-lattices
            "vs_baseline": None, "dtype": dtype_name[args.prec], "data": "synthetic",
            "config": {"workload": "%s even-odd Dslash (DiracTwistedMassPC::Dslash, kappa=%g mu=%g), %s lattice, recon-%d, fields resident in HBM"
                       % ({"tm": "twisted-mass", "tmc": "twisted-clover", "wilson": "Wilson"}[args.dslash], kappa, mu, "x".join(map(str, X)), args.recon),
                       "local_lattice": Xl, "halo_transport": {1: "direct peer stores (IPC-mapped ghost zones over xGMI)", 0: "RCCL send/recv", -1: "none (single rank)"}[int(qa.lib().qudaAmdHaloTransport()) if int(qa.lib().qudaAmdCommSize()) > 1 else -1], "halo_wire_format": {0: "flag-in-data", 1: "self-validating 16-byte atoms"}[int(qa.lib().qudaAmdHaloWireFormat())] if dist else "none", "process_grid": dist.grid if dist else [1, 1, 1, 1], "flops_per_site": r["flops_site"],
                       "ranks_in_communicator": int(qa.lib().qudaAmdCommSize()), "per_rank_kernel_us": {"slowest": round(1e6 * r["sec"], 2), "fastest": round(1e6 * r["sec_min"], 2)},
                       "per_rank": ranks_dslash,   # transport + exchange / global-sum counters of every rank after the Dslash measurement
                       "other_configs": "BASELINE configs[3] (32^3 x 64 over 8 GPUs): --lattice 32,32,32,64 (grid 1x2x2x2, local 32x16x16x32)"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": traffic, "traffic_unit": "B/launch", "traffic_source": traffic_source, "bytes_per_site": r["bytes_site"], "kernel_us": round(1e6 * r["sec"], 3)},
            "cpu_baseline": cpu,
            "extra": extra,
        }
    if world > 1 and not args.no_mg and (X == [32, 32, 32, 32] or os.environ.get("QUDA_AMD_BENCH_MG_ANY")):
        # The second half of the metric on the decomposed lattice: the same 32^4 MG-GCR problem as extra.mg_gcr at N = 1, the
        # hierarchy built and the K-cycle run across the ranks (aggregates never straddle ranks, coarse halos through the same
        # transport, global sums through the collective path).  The Dslash line above is complete at this point; a guard makes sure
        # it is printed whatever happens in this leg — the library ends the process on an error (errorQuda -> exit, as the
        # reference), and a collective that never returns would otherwise take the line with it.
        guard = _LineGuard(qa, out if rank == 0 else None, float(os.environ.get("QUDA_AMD_BENCH_MG_TIMEOUT", "420")))
        try:
            res = run_mg_ranks(qa, dist, X)
        except Exception as e:
            res = dict(failed=str(e)[:300])
        guard.disarm()
        ranks_mg = per_rank_report()
        if rank == 0:
            if isinstance(res, dict):
                res["per_rank"] = ranks_mg   # counters now include the hierarchy set-up and the solves
            out["extra"]["mg_gcr"] = res
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.finalize()
    else:
        qa.end()


if __name__ == "__main__":
    main()
