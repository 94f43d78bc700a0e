#!/usr/bin/env python3
"""Launch-geometry sweep of the fine-grid stencil on one GPU: every (precision, action) x every tuning setting in one process
(qudaAmdSetDslashTune), HIP-event time per application and the fraction of the 8 TB/s HBM roofline on ALGORITHMIC bytes.

usage: dslash_sweep.py LATTICE "PREC:KIND,..." "key=v,key=v;key=v,..." [steps]
  e.g. dslash_sweep.py 48,48,48,96 "4:tm,8:tm" "tiled=0,order=1;tiled=0,order=4;tiled=1,nxz=8,tt=2"
The links are a periodic tiling of a small random SU(3) set (timing does not depend on the values); results are not checked here
(tests/test_dslash_gpu.py checks every order against the oracle)."""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from synth import make_clover, tiled_gauge  # noqa: E402


def main():
    X = [int(v) for v in sys.argv[1].split(",")]
    cases = [(int(c.split(":")[0]), c.split(":")[1]) for c in sys.argv[2].split(",")]
    settings = [dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in s.split(",") if kv) for s in sys.argv[3].split(";")]
    steps = int(sys.argv[4]) if len(sys.argv) > 4 else 30
    qa = importlib.import_module("quda-qkxtm-multigrid_amd")
    qa.init(0)
    L = qa.lib()
    t0 = time.time()
    gauge = tiled_gauge(X)
    Vh = int(np.prod(X)) // 2
    src_h = np.random.default_rng(1).random(Vh * 24)
    clover = None
    print("inputs %.1f s" % (time.time() - t0), flush=True)
    kinds = {"tm": qa.QUDA_TWISTED_MASS_DSLASH, "tmc": qa.QUDA_TWISTED_CLOVER_DSLASH, "wilson": qa.QUDA_WILSON_DSLASH}
    defaults = dict(block=0, remap=1, order=1, store_aux=-1, tiled=-1, nxz=0, tz=0, tt=0, lds_pad=0, ygroups=-1)
    for prec, kind in cases:
        qa.load_gauge(gauge, qa.gauge_param(X, cuda_prec=prec))
        ip = qa.invert_param(kinds[kind], 0.1, 0.01, +1, "ee", 0, cuda_prec=prec)
        if kind == "tmc":
            if clover is None:
                clover = make_clover(X, seed=11)
            qa.load_clover(clover, None, ip)
        src, dst = qa.Spinor(prec), qa.Spinor(prec)
        src.load(src_h, ip)
        d = qa.Dirac(ip, pc=True)
        nbytes = L.qudaAmdDslashBytesPerSite(ip, 0, 0) * Vh
        ref = None
        for s in settings:
            for k, v in defaults.items():
                L.qudaAmdSetDslashTune(k.encode(), s.get(k, v))
            d.time_dslash(dst, src, 0, 5)
            best = min(d.time_dslash(dst, src, 0, steps) for _ in range(3))
            n2 = dst.norm2()
            if ref is None:
                ref = n2
            ok = abs(n2 - ref) <= 1e-9 * abs(ref)
            print("%s prec %d %-6s %-48s %9.2f us  %6.3f TB/s  frac %.3f  %s" % ("x".join(map(str, X)), prec, kind, ",".join("%s=%d" % kv for kv in s.items()) or "default",
                                                                                1e6 * best, nbytes / best * 1e-12, nbytes / best / 8e12, "" if ok else "NORM MISMATCH %r %r" % (n2, ref)), flush=True)
        for k, v in defaults.items():
            L.qudaAmdSetDslashTune(k.encode(), v)
        src.free(); dst.free(); d.free()
    qa.end()


if __name__ == "__main__":
    main()
