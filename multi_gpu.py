"""multi_gpu.py — host-side logic of the 4-D grid decomposition (one process per GPU).

Mirrors what the reference's harness does around initCommsGridQuda (tests/test_util.cpp:50-92, lib/interface_quda.cpp:261-285):
choose a process grid, map rank <-> grid coordinates (t fastest), cut the global even-odd ordered host fields into
each rank's local sub-lattice, and bootstrap the transport.  The data path (halo exchange, all-reduce, barrier, max over
ranks) is RCCL inside libquda.so; the only out-of-band step is a 128-byte TCP broadcast of the RCCL id on MASTER_ADDR.

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
    """Handle on the initialised multi-rank run.  Barrier and max-over-ranks go through libquda's own RCCL communicator."""

    def __init__(self, qa, rank, world, grid, X):
        self.qa, self.rank, self.world, self.grid, self.X = qa, rank, world, grid, X
        self.coords = rank_to_coords(rank, grid)
        self.local_dims = local_dims(X, grid)

    def barrier(self):
        self.qa.lib().qudaAmdCommBarrier()

    def max_over_ranks(self, v):
        buf = (C.c_double * 1)(float(v))
        self.qa.lib().qudaAmdCommAllreduceMax(buf, 1)
        return float(buf[0])

    def scatter_gauge(self, gauge):
        # every rank regenerates the same seeded global field (cheaper than shipping 1.2 GB through a control plane)
        return scatter_gauge(gauge, self.X, self.grid, self.coords)

    def finalize(self):
        self.barrier()
        self.qa.end()


def _launcher_token():
    """identifies the process that launched this rank (torch.distributed.run's agent, bench.py's own launcher, a shell script): its pid
    and its start time in clock ticks (/proc/<pid>/stat field 22) — the same for all sibling ranks, never the same for two launches"""
    ppid = os.getppid()
    try:
        with open("/proc/%d/stat" % ppid) as f:
            start = f.read().rsplit(")", 1)[1].split()[19]
    except OSError:
        start = "0"
    return "%d_%s" % (ppid, start)


def _broadcast_id(rank, world, payload):
    """Out-of-band broadcast of the 128-byte RCCL id from rank 0.  Deliberately NOT torch.distributed: importing torch would load a
    second HIP runtime (torch bundles its own ROCm libraries) next to the /opt/rocm one libquda.so is linked against.
    Two channels, both always open: a file in the node's temp directory named after the launcher (pid + start time: sibling ranks agree
    on it, no earlier launch can have left one) — the ranks of one node need nothing else, and no port can be taken — and a TCP socket on
    MASTER_ADDR, port MASTER_PORT + 1 (QUDA_AMD_BOOTSTRAP_PORT), for ranks that do not share the launcher or the file system.  A rank
    takes the file if it appears within QUDA_AMD_BOOTSTRAP_FILE_WAIT seconds (default 20), the socket otherwise."""
    import socket
    import tempfile
    import threading
    import time

    addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
    port = int(os.environ.get("QUDA_AMD_BOOTSTRAP_PORT", int(os.environ.get("MASTER_PORT", "29500")) + 1))
    path = os.path.join(tempfile.gettempdir(), ".quda_amd_rccl_id_%s_%s" % (os.environ.get("MASTER_PORT", "0"), _launcher_token()))
    if rank == 0:
        tmp = path + ".part"
        with open(tmp, "wb") as f:
            f.write(payload)
        os.replace(tmp, path)   # appears whole or not at all
        import atexit
        atexit.register(lambda: os.path.exists(path) and os.remove(path))

        def serve():
            try:
                srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
                srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                srv.bind((addr, port))
                srv.listen(world)
            except OSError:
                return   # port taken: the file is the channel
            while True:
                try:
                    c, _a = srv.accept()
                    c.sendall(payload)
                    c.close()
                except OSError:
                    return
        threading.Thread(target=serve, daemon=True).start()
        return payload
    file_wait = float(os.environ.get("QUDA_AMD_BOOTSTRAP_FILE_WAIT", "20"))
    t0 = time.time()
    deadline = t0 + 300
    while True:
        if os.path.exists(path):
            with open(path, "rb") as f:
                buf = f.read()
            if len(buf) == 128:
                return buf
        if time.time() - t0 > file_wait:
            try:
                c = socket.create_connection((addr, port), timeout=5)
                c.settimeout(10)
                buf = b""
                while len(buf) < 128:
                    chunk = c.recv(128 - len(buf))
                    if not chunk:
                        break
                    buf += chunk
                c.close()
                if len(buf) == 128:
                    return buf
            except OSError:
                pass
        if time.time() > deadline:
            raise RuntimeError("rank %d: no RCCL id from rank 0 after 300 s (file %s, socket %s:%d)" % (rank, path, addr, port))
        time.sleep(0.05)


def setup(qa, rank, world, local_rank, X, grid=None):
    """initQudaDevice -> RCCL bootstrap -> initCommsGridQuda -> initQuda, returning the Dist helper"""
    grid = grid or choose_grid(world)
    L = qa.lib()
    L.setVerbosityQuda(qa.QUDA_SILENT, b"", None)
    local_rank = int(os.environ.get("QUDA_AMD_FORCE_DEVICE", local_rank))  # rehearsal of N ranks on fewer GPUs
    L.initQudaDevice(int(local_rank))
    payload = b""
    if rank == 0:
        buf = (C.c_char * 128)()
        L.qudaAmdCommGetUniqueId(buf)
        payload = bytes(buf.raw)
    payload = _broadcast_id(rank, world, payload)
    idb = (C.c_char * 128).from_buffer_copy(payload)
    L.qudaAmdCommInit(idb, int(rank), int(world))
    dims = (C.c_int * 4)(*grid)
    L.initCommsGridQuda(4, dims, None, None)
    L.initQuda(int(local_rank))
    return Dist(qa, rank, world, grid, X)


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
