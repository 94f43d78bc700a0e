"""world_size-2 CPU test (gloo) of the N > 1 path's host logic: process grid, rank <-> coordinate map, cutting the
global even-odd fields into local sub-lattices, the face (ghost-zone) indexing and the neighbour exchange pattern.
Each rank runs the oracle's grid-decomposed Wilson hop (restatement of the reference's MULTI_GPU host branch) on its
sub-lattice with faces received from the other rank, and the result must equal the single-rank global operator."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, grid, X, q):
    import ctypes as C

    import torch
    import torch.distributed as dist

    import multi_gpu as mg
    import oracle_api

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        oracle = oracle_api.load()
        gauge, spinor, _ = oracle.make_fields(X, clover=False)  # identical on both ranks (seeded glibc rand)
        coords = mg.rank_to_coords(rank, grid)
        assert mg.coords_to_rank(coords, grid) == rank
        Xl = mg.local_dims(X, grid)
        g_loc = mg.scatter_gauge(gauge, X, grid, coords)
        s_loc = mg.scatter_field(spinor, X, grid, coords, 24)
        Vh = int(np.prod(Xl)) // 2
        worst = 0.0
        for parity in (0, 1):
            for dagger in (0, 1):
                pin = 1 - parity
                src = s_loc.reshape(2, Vh, 24)[pin]
                part = [int(grid[d] > 1) for d in range(4)]
                fwd = [np.zeros(1)] * 4
                back = [np.zeros(1)] * 4
                gg = [np.zeros(1)] * 4
                for d in range(4):
                    if not part[d]:
                        continue
                    up = mg.coords_to_rank([(coords[k] + (k == d)) % grid[k] for k in range(4)], grid)
                    dn = mg.coords_to_rank([(coords[k] - (k == d)) % grid[k] for k in range(4)], grid)
                    # my x_d = L-1 slice goes forward (it is the neighbour's "back" ghost), my x_d = 0 slice goes backward
                    send_f = torch.from_numpy(np.ascontiguousarray(src[mg.face_cb_indices(Xl, d, 1, pin)]))
                    send_b = torch.from_numpy(np.ascontiguousarray(src[mg.face_cb_indices(Xl, d, 0, pin)]))
                    gl = g_loc[d].reshape(2, Vh, 18)
                    send_g = torch.from_numpy(np.ascontiguousarray(np.stack([gl[p][mg.face_cb_indices(Xl, d, 1, p)] for p in (0, 1)])))
                    rb, rf, rg = torch.empty_like(send_f), torch.empty_like(send_b), torch.empty_like(send_g)
                    # same posting order as libquda's commExchange: send forward, send backward, recv from behind, recv from ahead
                    reqs = [dist.isend(send_f, up, tag=10 * d), dist.isend(send_b, dn, tag=10 * d + 1), dist.isend(send_g, up, tag=10 * d + 2),
                            dist.irecv(rb, dn, tag=10 * d), dist.irecv(rf, up, tag=10 * d + 1), dist.irecv(rg, dn, tag=10 * d + 2)]
                    for r in reqs:
                        r.wait()
                    back[d], fwd[d], gg[d] = rb.numpy().reshape(-1), rf.numpy().reshape(-1), rg.numpy().reshape(-1)
                out = np.empty(Vh * 24)
                dp = C.POINTER(C.c_double)
                arr = lambda xs: (dp * 4)(*[np.ascontiguousarray(a).ctypes.data_as(dp) for a in xs])
                keep = [np.ascontiguousarray(a) for a in list(g_loc) + gg + fwd + back]
                oracle.lib.qo_wil_dslash_halo_d(out.ctypes.data_as(dp), arr(keep[0:4]), arr(keep[4:8]), np.ascontiguousarray(src).ctypes.data_as(dp),
                                                arr(keep[8:12]), arr(keep[12:16]), parity, dagger, (C.c_int * 4)(*Xl), (C.c_int * 4)(*part))
                nh = spinor.size // 2
                want_g = oracle.wil_dslash(gauge, spinor[pin * nh:(pin + 1) * nh].copy(), X, parity, dagger)
                full = np.zeros(2 * nh)
                full[parity * nh:(parity + 1) * nh] = want_g
                want = mg.scatter_field(full, X, grid, coords, 24).reshape(2, Vh * 24)[parity]
                worst = max(worst, float(np.max(np.abs(out - want))))
        # round trip of the scatter/gather maps
        back_g = np.zeros_like(spinor)
        mg.gather_field(s_loc, X, grid, coords, 24, back_g)
        idx = mg.local_to_global_cb(X, grid, coords)
        assert np.array_equal(back_g.reshape(2, -1, 24)[0][idx[0]], spinor.reshape(2, -1, 24)[0][idx[0]])
        q.put((rank, worst))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("grid", [[1, 1, 1, 2], [1, 1, 2, 1], [1, 2, 1, 1], [2, 1, 1, 1]])
def test_two_rank_decomposition_matches_global_operator(grid):
    import multiprocessing as mp  # NOT torch.multiprocessing: the pytest process may hold libquda.so (second HIP runtime)

    X = [4, 4, 4, 8]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, grid, X, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, worst in res:
        assert worst == 0.0, (rank, worst)  # same operation order as the single-rank loop -> bit identical


def test_grid_helpers():
    import multi_gpu as mg

    assert mg.choose_grid(1) == [1, 1, 1, 1] and mg.choose_grid(2) == [1, 1, 1, 2]
    assert mg.choose_grid(4) == [1, 1, 2, 2] and mg.choose_grid(8) == [1, 2, 2, 2]
    for n in (2, 4, 8):
        g = mg.choose_grid(n)
        seen = set()
        for r in range(n):
            c = mg.rank_to_coords(r, g)
            assert mg.coords_to_rank(c, g) == r
            seen.add(tuple(c))
        assert len(seen) == n
    # t fastest, as the reference's default map (lib/interface_quda.cpp:261-270)
    assert mg.rank_to_coords(1, [1, 2, 2, 2]) == [0, 0, 0, 1] and mg.rank_to_coords(4, [1, 2, 2, 2]) == [0, 1, 0, 0]
    with pytest.raises(ValueError):
        mg.local_dims([6, 4, 4, 4], [4, 1, 1, 1])
