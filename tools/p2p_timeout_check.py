#!/usr/bin/env python3
"""A neighbour that never delivers its face must produce an error, not a hang: rank 1 sets up the decomposition and then
leaves without applying the operator; rank 0 applies it and has to come back with `halo wait ran out` (and the record of what it waited for) after
QUDA_AMD_P2P_TIMEOUT_S.  Started once per rank (env RANK / WORLD_SIZE = 2 / file transport) by the GPU test."""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import multi_gpu as mg  # noqa: E402
from synth import make_gauge  # noqa: E402


def main():
    rank = int(os.environ["RANK"])
    qa = importlib.import_module("quda-qkxtm-multigrid_amd")
    X = [8, 8, 8, 16]
    dist = mg.setup(qa, rank, 2, 0, X, grid=[1, 1, 1, 2])
    Xl = dist.local_dims
    qa.load_gauge(dist.scatter_gauge(make_gauge(X)), qa.gauge_param(Xl, cuda_prec=4))
    ip = qa.invert_param(qa.QUDA_TWISTED_MASS_DSLASH, 0.1, 0.01, +1, "ee", 0, cuda_prec=4)
    src = np.random.default_rng(rank).random(int(np.prod(Xl)) // 2 * 24)
    qa.dslash(src, ip, 0)           # both ranks: a normal exchange first (also decides the transport)
    print("rank %d: transport %d" % (rank, qa.lib().qudaAmdHaloTransport()), flush=True)
    if rank == 1:
        time.sleep(4.0)
        os._exit(0)
    # rank 0 alone, through the device-field operator interface (dslashQuda itself meets the other ranks on the host first and
    # would report the missing rank there): its neighbour never packs, the in-kernel wait has to run out
    s_in, s_out = qa.Spinor(4), qa.Spinor(4)
    s_in.load(src, ip)
    d = qa.Dirac(ip, pc=True)
    d.dslash(s_out, s_in, 0)
    s_out.save(ip, src)             # reading the result back checks the device error word
    print("NOT REACHED: the missing face went unnoticed", flush=True)


if __name__ == "__main__":
    main()
