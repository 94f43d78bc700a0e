"""multi_gpu.py — host-side logic of the 4-D grid decomposition (one process per GPU).

Mirrors what the reference's harness does around initCommsGridQuda (tests/test_util.cpp:50-92, lib/interface_quda.cpp:261-285):
choose a process grid, map rank <-> grid coordinates (t fastest), cut the global even-odd ordered host fields into
each rank's local sub-lattice, and bootstrap the transport.  The data path (halo exchange, all-reduce) is RCCL inside
libquda.so; torch.distributed (gloo) is used ONLY as the out-of-band control plane that carries the 128-byte RCCL id,
the barrier and the max-over-ranks of the timings.

The pure-numpy helpers are exercised on CPU by tests/test_multirank_cpu.py (gloo, world_size 2) against the oracle.
"""
import ctypes as C
import os

import numpy as np


def choose_grid(n):
    """(x, y, z, t) process grid: split t first, then z, then y (x stays whole: it is the contiguous direction)."""
    grid = [1, 1, 1, 1]
    d = 3
    while n > 1:
        if n % 2:
            raise ValueError("number of ranks must be a power of two")
        grid[d] *= 2
        n //= 2
        d = d - 1 if d > 1 else 3
    return grid


def rank_to_coords(rank, grid):
    c = [0, 0, 0, 0]
    for d in (3, 2, 1, 0):
        c[d] = rank % grid[d]
        rank //= grid[d]
    return c


def coords_to_rank(c, grid):
    return ((c[0] * grid[1] + c[1]) * grid[2] + c[2]) * grid[3] + c[3]


def cb_coords(X, parity):
    """coordinates of every checkerboard site of `parity` in index order (reference tests/test_util.cpp:419-443)"""
    Vh = int(np.prod(X)) // 2
    i = np.arange(Vh)
    Xh = X[0] // 2
    za, xh = i // Xh, i % Xh
    zb, y = za // X[1], za % X[1]
    t, z = zb // X[2], zb % X[2]
    x = 2 * xh + ((y + z + t + parity) & 1)
    return x, y, z, t


def cb_index(X, x, y, z, t):
    return (((t * X[2] + z) * X[1] + y) * X[0] + x) // 2


def local_dims(X, grid):
    for d in range(4):
        if X[d] % grid[d] or (X[d] // grid[d]) % 2:
            raise ValueError("extent %d of dimension %d does not split evenly (and into even parts) over %d ranks" % (X[d], d, grid[d]))
    return [X[d] // grid[d] for d in range(4)]


def local_to_global_cb(X, grid, coords):
    """for each parity: global checkerboard index of every local checkerboard site (local parity == global parity)"""
    Xl = local_dims(X, grid)
    off = [coords[d] * Xl[d] for d in range(4)]
    out = []
    for p in (0, 1):
        x, y, z, t = cb_coords(Xl, p)
        out.append(cb_index(X, x + off[0], y + off[1], z + off[2], t + off[3]))
    return out


def scatter_field(glob, X, grid, coords, nreal):
    """global even-odd ordered host field (2*Vh_global*nreal reals) -> this rank's local even-odd ordered field"""
    Vh_g = int(np.prod(X)) // 2
    g = glob.reshape(2, Vh_g, nreal)
    idx = local_to_global_cb(X, grid, coords)
    return np.ascontiguousarray(np.stack([g[0][idx[0]], g[1][idx[1]]])).reshape(-1)


def gather_field(local, X, grid, coords, nreal, out):
    """inverse of scatter_field into a preallocated global array (used by tests)"""
    Vh_g = int(np.prod(X)) // 2
    idx = local_to_global_cb(X, grid, coords)
    l = local.reshape(2, -1, nreal)
    o = out.reshape(2, Vh_g, nreal)
    o[0][idx[0]] = l[0]
    o[1][idx[1]] = l[1]


def scatter_gauge(gauge, X, grid, coords):
    """(4, V*18) global QDP links -> (4, V_local*18).  Boundary signs are already folded into the global field, so the
    anti-periodic sign automatically sits on the last-t ranks only (reference tests/test_util.cpp:699-706)."""
    return np.stack([scatter_field(gauge[mu], X, grid, coords, 18) for mu in range(4)])


class Dist:
    def __init__(self, qa, rank, world, grid, X, tdist):
        self.qa, self.rank, self.world, self.grid, self.X, self.tdist = qa, rank, world, grid, X, tdist
        self.coords = rank_to_coords(rank, grid)
        self.local_dims = local_dims(X, grid)

    def barrier(self):
        self.tdist.barrier()

    def max_over_ranks(self, v):
        import torch

        t = torch.tensor([float(v)], dtype=torch.float64)
        self.tdist.all_reduce(t, op=self.tdist.ReduceOp.MAX)
        return float(t[0])

    def scatter_gauge(self, gauge_or_none):
        # every rank regenerates the same seeded global field (cheaper than shipping 1.2 GB through the control plane)
        assert gauge_or_none is not None
        return scatter_gauge(gauge_or_none, self.X, self.grid, self.coords)

    def finalize(self):
        self.qa.end()
        self.tdist.barrier()
        self.tdist.destroy_process_group()


def setup(qa, rank, world, local_rank, X, grid=None):
    """initQudaDevice -> RCCL bootstrap -> initCommsGridQuda -> initQuda, returning the Dist helper"""
    import torch
    import torch.distributed as tdist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    if not tdist.is_initialized():
        tdist.init_process_group("gloo", rank=rank, world_size=world)
    grid = grid or choose_grid(world)
    L = qa.lib()
    L.setVerbosityQuda(qa.QUDA_SILENT, b"", None)
    L.initQudaDevice(int(local_rank))
    uid = torch.zeros(128, dtype=torch.uint8)
    if rank == 0:
        buf = (C.c_char * 128)()
        L.qudaAmdCommGetUniqueId(buf)
        uid = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8).clone()
    tdist.broadcast(uid, src=0)
    idb = (C.c_char * 128).from_buffer_copy(bytes(uid.numpy().tobytes()))
    L.qudaAmdCommInit(idb, int(rank), int(world))
    dims = (C.c_int * 4)(*grid)
    L.initCommsGridQuda(4, dims, None, None)
    L.initQuda(int(local_rank))
    return Dist(qa, rank, world, grid, X, tdist)


def face_cb_indices(Xl, d, side, parity):
    """checkerboard indices of the parity-`parity` sites on the face x_d = 0 (side 0) or x_d = L-1 (side 1), ordered by the
    face index both the pack kernel and the ghost lookup use: lexicographic over the other three coordinates, halved."""
    others = [k for k in range(4) if k != d]
    L = [Xl[k] for k in others]
    nf = int(np.prod(Xl)) // Xl[d] // 2
    l = 2 * np.arange(nf)
    c0, l = l % L[0], l // L[0]
    c1, c2 = l % L[1], l // L[1]
    c = [None] * 4
    c[d] = np.full(nf, Xl[d] - 1 if side else 0)
    c[others[0]], c[others[1]], c[others[2]] = c0, c1, c2
    c[others[0]] = c[others[0]] + ((parity + c[0] + c[1] + c[2] + c[3]) & 1)
    return cb_index(Xl, c[0], c[1], c[2], c[3])
