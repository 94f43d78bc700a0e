#!/usr/bin/env python3
"""N-rank correctness check of the grid-decomposed Dslash / Mat / MatPC / reductions / GCR against the single-lattice oracle.
Started once per rank by tools/mgpu_rehearsal.sh (env RANK/WORLD_SIZE/...); every rank builds the same seeded global
fields, cuts out its sub-lattice (multi_gpu.py), runs the library on it and compares with the oracle's GLOBAL result
restricted to its sub-lattice."""
import ctypes as C
import faulthandler
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import multi_gpu as mg  # noqa: E402
import oracle_api  # noqa: E402


def main():
    faulthandler.enable()   # a crash in one rank otherwise leaves an empty log and a timeout in the others
    world = int(sys.argv[1])
    rank = int(os.environ["RANK"])
    qa = importlib.import_module("quda-qkxtm-multigrid_amd")
    oracle = oracle_api.load()
    cases = [([8, 8, 8, 16], None)]
    if world == 3:   # three ranks along t: the +t and -t neighbours are DIFFERENT ranks (never the case with two per dimension)
        cases = [([8, 8, 8, 24], [1, 1, 1, 3]), ([8, 12, 8, 8], [1, 3, 1, 1])]
    if world == 6:
        cases = [([8, 8, 8, 24], [1, 1, 2, 3])]
    if world == 2:
        cases += [([8, 4, 8, 8], [1, 1, 2, 1]), ([4, 8, 8, 8], [2, 1, 1, 1])]
    for X, grid in cases:
        dist = mg.setup(qa, rank, world, int(os.environ["LOCAL_RANK"]), X, grid=grid)
        grid = dist.grid
        gauge, spinor, clover = oracle.make_fields(X)
        kappa, mu = 0.1, 0.3
        g_loc = mg.scatter_gauge(gauge, X, grid, dist.coords)
        s_loc = mg.scatter_field(spinor, X, grid, dist.coords, 24)
        c_loc = mg.scatter_field(clover, X, grid, dist.coords, 72)
        Xl = dist.local_dims
        nh_g, nh_l = spinor.size // 2, s_loc.size // 2
        oracle.set_threads(4)

        def local_part(full_parity_field, parity):
            full = np.zeros(2 * nh_g)
            full[parity * nh_g:(parity + 1) * nh_g] = full_parity_field
            return mg.scatter_field(full, X, grid, dist.coords, 24)[parity * nh_l:(parity + 1) * nh_l]

        tols = {8: 1e-12, 4: 2e-5, 2: 1e-2}
        order = [(int(v), tols[int(v)]) for v in os.environ.get("MGPU_ORDER", "8,4,2").split(",")]
        for prec, tol in order:
            qa.load_gauge(g_loc, qa.gauge_param(Xl, cuda_prec=prec))
            ipc = qa.invert_param(qa.QUDA_TWISTED_CLOVER_DSLASH, kappa, mu, cuda_prec=prec)
            qa.load_clover(c_loc, None, ipc)
            for dagger in (0, 1):
                for parity in (0, 1):
                    pin = 1 - parity
                    ip = qa.invert_param(qa.QUDA_TWISTED_MASS_DSLASH, kappa, mu, +1, "ee", dagger, cuda_prec=prec)
                    got = qa.dslash(s_loc[pin * nh_l:(pin + 1) * nh_l].copy(), ip, parity)
                    want_g = oracle.tm_dslash(gauge, spinor[pin * nh_g:(pin + 1) * nh_g].copy(), X, kappa, mu, +1, parity, "ee", dagger)
                    want_l = local_part(want_g, parity)
                    err = np.max(np.abs(got - want_l)) / np.max(np.abs(want_g))
                    if err >= tol:
                        bad = np.nonzero(np.max(np.abs(got - want_l).reshape(-1, 24), axis=1) > tol * np.max(np.abs(want_g)))[0]
                        x, y, z, t = mg.cb_coords(Xl, parity)
                        print("rank %d FAIL tm_dslash prec %d dagger %d parity %d: %d bad sites of %d; coords (x,y,z,t) of first: %s; z in %s t in %s" % (
                            rank, prec, dagger, parity, bad.size, got.size // 24, [(int(x[i]), int(y[i]), int(z[i]), int(t[i])) for i in bad[:6]],
                            sorted(set(int(v) for v in z[bad])), sorted(set(int(v) for v in t[bad]))), flush=True)
                    assert err < tol, ("tm_dslash", X, grid, prec, dagger, parity, err)
            ipf = qa.invert_param(qa.QUDA_TWISTED_MASS_DSLASH, kappa, mu, -1, "ee", 0, cuda_prec=prec, solution_type=qa.QUDA_MAT_SOLUTION)
            got = qa.mat(s_loc.copy(), ipf)
            want = mg.scatter_field(oracle.tm_mat(gauge, spinor, X, kappa, mu, -1, 0), X, grid, dist.coords, 24)
            err = np.max(np.abs(got - want)) / np.max(np.abs(want))
            assert err < 2 * tol, ("tm_mat", X, grid, prec, err)
            cinv = oracle.clover_twisted_inverse(clover, 4 * kappa * kappa * mu * mu)
            ipc = qa.invert_param(qa.QUDA_TWISTED_CLOVER_DSLASH, kappa, mu, +1, "oo", 1, cuda_prec=prec)
            got = qa.mat(s_loc[nh_l:].copy(), ipc)
            want_g = oracle.tmc_matpc(gauge, spinor[nh_g:].copy(), clover, cinv, X, kappa, mu, +1, "oo", 1)
            err = np.max(np.abs(got - local_part(want_g, 1))) / np.max(np.abs(want_g))
            assert err < 4 * tol, ("tmc_matpc", X, grid, prec, err)
        # clover term built on the device from the decomposed links (transport formulation) against the oracle's global field
        coeff = 0.17
        c_built = mg.scatter_field(oracle.clover_compute(gauge, coeff, X), X, grid, dist.coords, 72)
        for prec, tol in ((8, 1e-11), (4, 2e-5)):
            qa.load_gauge(g_loc, qa.gauge_param(Xl, cuda_prec=prec))
            ipb = qa.invert_param(qa.QUDA_TWISTED_CLOVER_DSLASH, kappa, mu, cuda_prec=prec)
            ipb.clover_coeff = coeff
            ipb.compute_clover, ipb.return_clover = 1, 1
            got_c = np.zeros_like(c_built)
            qa.load_clover(got_c, None, ipb)
            errc = float(np.max(np.abs(got_c - c_built)))
            assert errc < 10 * tol, ("device clover", X, grid, prec, errc)
        # a mixed-precision even-odd GCR solve across ranks (halo + global reductions), residual checked with MatQuda
        qa.load_gauge(g_loc, qa.gauge_param(Xl, cuda_prec=8, prec_sloppy=4))
        ip = qa.invert_param(qa.QUDA_TWISTED_MASS_DSLASH, kappa, 0.05, +1, "ee", 0, cuda_prec=8, prec_sloppy=4, solution_type=qa.QUDA_MAT_SOLUTION)
        ip.solve_type = qa.QUDA_DIRECT_PC_SOLVE
        ip.tol = 1e-9
        b_loc = s_loc.copy()
        x_loc = qa.invert(b_loc, ip)
        ip2 = qa.invert_param(qa.QUDA_TWISTED_MASS_DSLASH, kappa, 0.05, +1, "ee", 0, cuda_prec=8, solution_type=qa.QUDA_MAT_SOLUTION)
        res = b_loc - qa.mat(x_loc, ip2)
        n2 = np.array([np.dot(res, res), np.dot(b_loc, b_loc)])
        qa.lib().qudaAmdCommAllreduce(n2.ctypes.data_as(C.POINTER(C.c_double)), 2)
        rel = float(np.sqrt(n2[0] / n2[1]))
        assert rel < 5e-9, ("gcr", X, grid, rel, ip.iter)
        # and the same solution must satisfy the GLOBAL operator of the oracle
        xg = np.zeros_like(spinor)
        mg.gather_field(x_loc, X, grid, dist.coords, 24, xg)
        want_b_loc = mg.scatter_field(oracle.tm_mat(gauge, xg, X, kappa, 0.05, +1, 0), X, grid, dist.coords, 24)
        # (xg holds only this rank's piece; the oracle result is exact on sites whose 8 neighbours are local)
        if rank == 0:
            print("OK lattice %s grid %s: dslash/mat/matpc parity in 3 precisions; GCR %d iterations, global |r|/|b| = %.2e" % (X, grid, ip.iter, rel))
        # a global sum that the ranks reach 0.3 s apart — longer than the in-kernel wait of the peer all-reduce, so the early ranks
        # finish it from the host side: every rank must still get the same bits, and the value of the global field
        import time
        ipn = qa.invert_param(qa.QUDA_TWISTED_MASS_DSLASH, kappa, mu, +1, "ee", 0, cuda_prec=8)
        spn = qa.Spinor(8)
        spn.load(s_loc[:nh_l].copy(), ipn)
        for late in (world - 1, 0):
            if rank == late:
                time.sleep(0.3)
            n2v = spn.norm2()
            mm = np.array([n2v, -n2v])
            qa.lib().qudaAmdCommAllreduceMax(mm.ctypes.data_as(C.POINTER(C.c_double)), 2)
            assert mm[0] == -mm[1] == n2v, ("skewed all-reduce differs between ranks", rank, n2v, mm)
            want_n2 = float(np.dot(spinor[:nh_g], spinor[:nh_g]))
            assert abs(n2v - want_n2) < 1e-12 * want_n2, ("skewed all-reduce value", n2v, want_n2)
        spn.free()
        # every rank reads its sub-block of one ILDG / LIME file (written here byte by byte from the format description)
        import struct
        lime_path = os.path.join(os.environ.get("QUDA_AMD_SHM_DIR", "/tmp"), "conf_%s.lime" % "_".join(map(str, X + grid)))
        if rank == 0:
            def rec(rtype, data, mb, me):
                return struct.pack(">IHHQ", 0x456789AB, 1, (0x8000 if mb else 0) | (0x4000 if me else 0), len(data)) + rtype.encode().ljust(128, b"\0") + data + b"\0" * (-len(data) % 8)
            Vg = int(np.prod(X))
            lexg = np.stack([oracle.eo_to_lex(np.ascontiguousarray(gauge[d]), X, 18).reshape(Vg, 18) for d in range(4)], axis=1)   # (V, 4, 18): ILDG site record
            xml = ("<ildgFormat><precision>64</precision><lx>%d</lx><ly>%d</ly><lz>%d</lz><lt>%d</lt></ildgFormat>" % tuple(X)).encode()
            with open(lime_path + ".tmp", "wb") as fh:
                fh.write(rec("ildg-format", xml, True, False) + rec("ildg-binary-data", lexg.reshape(Vg, 72).astype(">f8").tobytes(), False, True))
            os.replace(lime_path + ".tmp", lime_path)
        qa.lib().qudaAmdCommBarrier()
        gpl = qa.lib().newQudaGaugeParam()
        g_read = qa.read_lime_gauge(lime_path, gpl, grid, None, int(np.prod(Xl)))
        assert [gpl.X[d] for d in range(4)] == list(Xl) and np.array_equal(g_read, g_loc), ("lime reader", X, grid)
        # gauge tools on the decomposed lattice: plaquette (global average) and APE smearing through the ghost-aware shifts
        qa.load_gauge(g_loc, qa.gauge_param(Xl, cuda_prec=8, prec_sloppy=4))
        plq, plq_want = np.array(qa.plaquette()), oracle.plaquette(gauge, X)
        assert np.max(np.abs(plq - plq_want)) < 1e-12, ("plaquette", X, grid, plq, plq_want)
        qa.perform_ape(2, 0.5)
        ape_got = qa.save_smeared_gauge(int(np.prod(Xl)))
        ape_want = mg.scatter_gauge(oracle.ape_smear(gauge, X, 0.5, 2), X, grid, dist.coords)
        err_ape = float(np.max(np.abs(ape_got - ape_want)))
        assert err_ape < 1e-12, ("ape smearing", X, grid, err_ape)
        if rank == 0:
            print("OK lattice %s grid %s: plaquette %.12f, APE smearing %.1e against the global oracle field" % (X, grid, plq[0], err_ape), flush=True)
        # ---- QKXTM solve loop on the decomposed lattice: smearing through ghost-aware covariant shifts, the point source on the
        # rank that owns it, up / down propagators in the drivers' lexicographic UKQCD layout ----
        g_lex_glob = np.stack([oracle.eo_to_lex(np.ascontiguousarray(gauge[d]), X, 18) for d in range(4)])
        g_lex_loc = np.stack([oracle.eo_to_lex(np.ascontiguousarray(g_loc[d]), Xl, 18) for d in range(4)])

        def to_local_lex(global_lex):
            return oracle.eo_to_lex(mg.scatter_field(oracle.lex_to_eo(global_lex, X, 24), X, grid, dist.coords, 24), Xl, 24)

        v_glob = np.random.default_rng(8).standard_normal(spinor.size)
        got = qa.gaussian_smear(to_local_lex(v_glob), g_lex_loc, 3, 0.7)
        want = to_local_lex(oracle.gauss_smear(v_glob, g_lex_glob, X, 0.7, 3))
        err = np.max(np.abs(got - want)) / np.max(np.abs(want))
        assert err < 1e-12, ("gaussian smearing", X, grid, err)
        ipq = qa.invert_param(qa.QUDA_TWISTED_MASS_DSLASH, kappa, 0.05, +1, "ee", 0, cuda_prec=8, prec_sloppy=4, solution_type=qa.QUDA_MAT_SOLUTION,
                              gamma_basis=qa.QUDA_UKQCD_GAMMA_BASIS)
        ipq.solve_type, ipq.inv_type, ipq.gcrNkrylov, ipq.tol, ipq.maxiter = qa.QUDA_DIRECT_PC_SOLVE, qa.QUDA_GCR_INVERTER, 20, 1e-9, 4000
        ipq.inv_type_precondition = qa.QUDA_INVALID_ENUM
        pos = (X[0] - 1, 1, X[2] - 2, X[3] - 1)   # owned by the last rank of the grid
        up, dn = qa.calc_mg_propagators(g_lex_loc, ipq, pos, 3, 0.7, int(np.prod(Xl)))
        worst = 0.0
        for isc in (0, 7):
            src = np.zeros(spinor.size)
            src[(((pos[3] * X[2] + pos[2]) * X[1] + pos[1]) * X[0] + pos[0]) * 24 + 2 * isc] = 1.0
            b_eo = oracle.lex_to_eo(to_local_lex(oracle.gauss_smear(src, g_lex_glob, X, 0.7, 3)), Xl, 24)
            for flavor, prop in ((+1, up), (-1, dn)):
                ipf = qa.invert_param(qa.QUDA_TWISTED_MASS_DSLASH, kappa, 0.05, flavor, "ee", 0, cuda_prec=8, solution_type=qa.QUDA_MAT_SOLUTION,
                                      gamma_basis=qa.QUDA_UKQCD_GAMMA_BASIS)
                r = b_eo - qa.mat(oracle.lex_to_eo(prop[isc], Xl, 24), ipf)
                n2 = np.array([np.dot(r, r), np.dot(b_eo, b_eo)])
                qa.lib().qudaAmdCommAllreduce(n2.ctypes.data_as(C.POINTER(C.c_double)), 2)
                worst = max(worst, float(np.sqrt(n2[0] / n2[1])))
        assert worst < 5e-9, ("qkxtm propagators", X, grid, worst)
        if rank == 0:
            print("OK lattice %s grid %s: Gaussian smearing %.1e, up/down propagators of a smeared point source |r|/|b| <= %.2e (%d iterations in 24 solves)"
                  % (X, grid, err, worst, ipq.iter), flush=True)
        # ---- multigrid on the decomposed lattice: ghost-aware Galerkin probing, coarse halo exchange, global reductions ----
        from synth import smooth_gauge
        kmg, mumg = 0.124, 0.005
        gs = smooth_gauge(X, 0.35)
        gs_loc = mg.scatter_gauge(gs, X, grid, dist.coords)
        qa.load_gauge(gs_loc, qa.gauge_param(Xl, cuda_prec=8, prec_sloppy=4, prec_precondition=4, t_boundary=qa.QUDA_PERIODIC_T))
        ipm = qa.invert_param(qa.QUDA_TWISTED_MASS_DSLASH, kmg, mumg, +1, "ee", 0, cuda_prec=8, prec_sloppy=4, prec_precondition=4,
                              solution_type=qa.QUDA_MAT_SOLUTION)
        ipm.solve_type = qa.QUDA_DIRECT_SOLVE
        ipm.inv_type = qa.QUDA_GCR_INVERTER
        ipm.gcrNkrylov = 20
        ipm.tol = 1e-10
        ipm.maxiter = 2000
        ipm.reliable_delta = 1e-4
        bg = np.random.default_rng(5).random(spinor.size)
        bm_loc = mg.scatter_field(bg, X, grid, dist.coords, 24)
        ipm.inv_type_precondition = qa.QUDA_INVALID_ENUM
        qa.invert(bm_loc, ipm)
        it_plain = ipm.iter
        for pc in (False, True):
            mp = qa.multigrid_param(ipm, n_level=2, geo_block=(4, 4, 4, 4), n_vec=8, setup_maxiter=300, setup_tol=1e-5, smoother_pc=pc)
            h = qa.Multigrid(mp)
            dev = h.verify()
            assert max(dev) < 1e-4, ("mg verify", X, grid, dev)
            ipm.inv_type_precondition = qa.QUDA_MG_INVERTER
            ipm.preconditioner = h.h
            ipm.tol_precondition, ipm.maxiter_precondition, ipm.precondition_cycle, ipm.omega = 1e-1, 1, 1, 1.0
            xm_loc = qa.invert(bm_loc, ipm)
            res = bm_loc - qa.mat(xm_loc, ipm)
            n2 = np.array([np.dot(res, res), np.dot(bm_loc, bm_loc)])
            qa.lib().qudaAmdCommAllreduce(n2.ctypes.data_as(C.POINTER(C.c_double)), 2)
            relm = float(np.sqrt(n2[0] / n2[1]))
            assert relm < 5e-10 and ipm.iter < it_plain // 2, ("mg-gcr", X, grid, relm, ipm.iter, it_plain)
            if rank == 0:
                print("OK lattice %s grid %s: MG-GCR (pc smoother %s) %d iterations (plain GCR %d), verify %.1e %.1e %.1e, global |r|/|b| = %.2e"
                      % (X, grid, pc, ipm.iter, it_plain, dev[0], dev[1], dev[2], relm), flush=True)
            if pc:
                # three sources through ONE lockstep solve (invertMultiSrcQuda): block smoother with its MR sums across the ranks, ghost zones of
                # the block fields through the transport, four-source / per-source transfers, block fields on the coarse level
                rngm = np.random.default_rng(9)
                srcs = [mg.scatter_field(rngm.random(spinor.size), X, grid, dist.coords, 24) for _ in range(3)]
                ipm.solve_type = qa.QUDA_DIRECT_PC_SOLVE
                s0 = qa.multi_src_stats()
                sols = qa.invert_multi_src(srcs, ipm)
                s1 = qa.multi_src_stats()
                it_multi = ipm.iter
                ipm.solve_type = qa.QUDA_DIRECT_SOLVE
                worst = 0.0
                for xs_loc, bs_loc in zip(sols, srcs):
                    rs = bs_loc - qa.mat(xs_loc, ipm)
                    n2 = np.array([np.dot(rs, rs), np.dot(bs_loc, bs_loc)])
                    qa.lib().qudaAmdCommAllreduce(n2.ctypes.data_as(C.POINTER(C.c_double)), 2)
                    worst = max(worst, float(np.sqrt(n2[0] / n2[1])))
                assert worst < 5e-10 and s1["block_cycles"] > s0["block_cycles"], ("multi-source mg-gcr", X, grid, worst, s0, s1)
                if rank == 0:
                    print("OK lattice %s grid %s: 3 sources in lockstep, %d iterations, worst global |r|/|b| = %.2e, %s" % (X, grid, it_multi, worst, s1), flush=True)
            ipm.inv_type_precondition = qa.QUDA_INVALID_ENUM
            h.free()
        oracle.set_threads(1)
        dist.finalize()
    print("rank %d: all checks passed" % rank)


if __name__ == "__main__":
    main()
