#!/usr/bin/env python3
"""Per-call cost of the halo exchange and the reductions through RCCL (self-loop mode, one GPU) against the local-copy
self-neighbour path: usage rccl_latency.py [0|1]  (1 = through RCCL)."""
import ctypes as C
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
use_rccl = len(sys.argv) > 1 and sys.argv[1] == "1"
if use_rccl:
    os.environ["QUDA_AMD_RCCL_SELFTEST"] = "1"
import multi_gpu  # noqa: E402
from synth import make_gauge  # noqa: E402


def main():
    qa = importlib.import_module("quda-qkxtm-multigrid_amd")
    X = [32, 16, 16, 16]
    t0 = time.perf_counter()
    multi_gpu.setup(qa, 0, 1, 0, X, grid=[1, 1, 1, 1])
    print("setup (incl. ncclCommInitRank: %s) %.2f s" % (use_rccl, time.perf_counter() - t0), flush=True)
    gauge = make_gauge(X, seed=3)
    masks = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else (0, 8, 14)
    precs = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else (8, 4)
    for mask in masks:
        qa.lib().qudaAmdSetPartitionMask(mask)
        for prec in precs:
            qa.load_gauge(gauge, qa.gauge_param(X, cuda_prec=prec))
            ip = qa.invert_param(qa.QUDA_TWISTED_MASS_DSLASH, 0.1, 0.01, +1, "ee", 0, cuda_prec=prec)
            a, b = qa.Spinor(prec), qa.Spinor(prec)
            a.load(np.random.default_rng(1).random(int(np.prod(X)) // 2 * 24), ip)
            d = qa.Dirac(ip, pc=True)
            qa.lib().qudaAmdTimeDslash(d.h, b.h, a.h, 0, 5)
            sec = qa.lib().qudaAmdTimeDslash(d.h, b.h, a.h, 0, 100)
            print("mask %2d prec %d: %.1f us per Dslash" % (mask, prec, 1e6 * sec), flush=True)
            t0 = time.perf_counter()
            for _ in range(100):
                a.norm2()
            print("            norm2 (reduction + host sync): %.1f us" % (1e4 * (time.perf_counter() - t0)), flush=True)
            d.free(); a.free(); b.free()
    qa.end()


if __name__ == "__main__":
    main()
