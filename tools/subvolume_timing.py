#!/usr/bin/env python3
"""Dslash time on the 8-GPU strong-scaling sub-lattice (32x16x16x16 of the 32^4 bench lattice) on one GPU: unpartitioned and
with y,z,t partitioned through the self-neighbour emulation (qudaAmdSetPartitionMask).  Launch geometry is read from the
environment (QUDA_AMD_DSLASH_BLOCK, ...), so run once per setting.  usage: subvolume_timing.py [prec] [lattice] [mask]"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from synth import make_gauge  # noqa: E402

prec = int(sys.argv[1]) if len(sys.argv) > 1 else 8
X = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "32,16,16,16").split(",")]
qa = importlib.import_module("quda-qkxtm-multigrid_amd")
qa.init(0)
gauge = make_gauge(X)
Vh = int(np.prod(X)) // 2
src_h = np.random.default_rng(1).random(Vh * 24)
out = []
pmask = int(sys.argv[3]) if len(sys.argv) > 3 else 0b1110
# optional 4th argument: stencil tuning variants "key=value[+key=value...],..." measured alternately in this one process (A/B on the
# same box and clock state), e.g. "p2p_fold=1,p2p_fold=0"; 5th: rounds
variants = [v for v in (sys.argv[4].split(",") if len(sys.argv) > 4 else [""])]
rounds = int(sys.argv[5]) if len(sys.argv) > 5 else 1
for mask in (0, pmask):
    qa.lib().qudaAmdSetPartitionMask(mask)
    qa.load_gauge(gauge, qa.gauge_param(X, cuda_prec=prec))
    ip = qa.invert_param(qa.QUDA_TWISTED_MASS_DSLASH, 0.1, 0.01, +1, "ee", 0, cuda_prec=prec)
    src, dst = qa.Spinor(prec), qa.Spinor(prec)
    src.load(src_h, ip)
    d = qa.Dirac(ip, pc=True)
    d.time_dslash(dst, src, 0, 50)
    if mask == 0 or variants == [""]:
        best = min(d.time_dslash(dst, src, 0, 500) for _ in range(3))
        out.append("mask %2d: %.2f us" % (mask, 1e6 * best))
    else:
        res = {v: [] for v in variants}
        for _ in range(rounds):
            for v in variants:
                for kv in v.split("+"):
                    k, val = kv.split("=")
                    qa.lib().qudaAmdSetDslashTune(k.encode(), int(val))
                d.time_dslash(dst, src, 0, 20)
                res[v].append(1e6 * min(d.time_dslash(dst, src, 0, 500) for _ in range(2)))
        out.append("mask %2d: " % mask + "; ".join("%s: %s" % (v, " ".join("%.2f" % t for t in res[v])) for v in variants))
    src.free(); dst.free(); d.free()
print("prec %d lattice %s remap %s: %s" % (prec, X, os.environ.get("QUDA_AMD_XCD_REMAP", "1"), "; ".join(out)))
qa.lib().qudaAmdSetPartitionMask(0)
qa.end()
