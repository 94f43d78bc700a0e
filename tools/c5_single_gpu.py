#!/usr/bin/env python3
"""BASELINE.json configs[4] ("C5": 48^3 x 96 MG-GCR, quoted by the reference on 8 GPUs) on ONE MI355X: with 288 GB of
HBM the whole three-level hierarchy (fp64 + fp32 links, 24 null vectors per level, the Galerkin operators) is resident on a
single device, so the run needs no halo exchange at all.  Levels follow the reference's blocking rule (lib/transfer.cpp:31-44;
SURVEY 8d): 48^3 x 96 -> 12^3 x 24 (4^4 aggregates) -> 6^3 x 6 (2^3 x 4, because 12 / 4 is odd).  Same solver set-up and
the same synthetic warm-start field family as bench.py's 32^4 leg (Cayley-map variant of synth.smooth_gauge, which at this
size would spend minutes in batched eigen-decompositions).  Prints one JSON object.

usage: tools/c5_single_gpu.py [--lattice 48,48,48,96] [--no-extras]"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lattice", default="48,48,48,96")
    ap.add_argument("--no-extras", action="store_true", help="skip the outer even-odd and half-precision-cycle solves")
    args = ap.parse_args()
    X = tuple(int(v) for v in args.lattice.split(","))
    import bench
    from synth import smooth_gauge_cayley
    qa = importlib.import_module("quda-qkxtm-multigrid_amd")
    t0 = time.perf_counter()
    gauge = smooth_gauge_cayley(X, 0.35, workers=min(16, os.cpu_count() or 8))
    t_gauge = time.perf_counter() - t0
    print("gauge field generated in %.1f s" % t_gauge, flush=True)
    # second-level aggregates by the reference's rule: largest of 4 / 2 that leaves an even coarse extent
    lvl1 = [x // 4 for x in X]
    b1 = tuple(4 if (x % 4 == 0 and (x // 4) % 2 == 0) else 2 for x in lvl1)
    qa.init(0)
    t0 = time.perf_counter()
    out = bench.run_mg(qa, X, blocks=((4, 4, 4, 4), b1, (2, 2, 2, 2)), gauge=gauge, extras=not args.no_extras, setup_repeats=2)
    out["wall_secs_total"] = round(time.perf_counter() - t0, 1)
    out["gauge_gen_secs"] = round(t_gauge, 1)
    out["n_gpus"] = 1
    free, total = qa.device_memory() if hasattr(qa, "device_memory") else (None, None)
    if total:
        out["device_memory_GiB"] = dict(total=round(total / 2**30, 1), free_after=round(free / 2**30, 1))
    qa.end()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
