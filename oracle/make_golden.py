#!/usr/bin/env python3
"""oracle/make_golden.py — TEST INFRASTRUCTURE.

Regenerates tests/golden/ from the REFERENCE's own host operators (oracle/_ref/ref_driver, built by
`make -C oracle ref` from the sources under /root/reference).  Runs only in the build container
(the GPU box has no /root/reference); the resulting fixtures are committed.

  tests/golden/ref_<X>x<Y>x<Z>x<T>.npz  inputs (gauge, spinor, clover, clover_inv) + outputs of
                                        wil_dslash, wil_mat, wil_matpc, apply_clover, tm_dslash, tm_matpc,
                                        tm_mat, tmc_dslash, tmc_matpc, tmc_mat   (fp64, element-wise)
  tests/golden/ref_checksums.json       ||out||^2 of selected operators on 8^4 / 16^4 with inputs drawn
                                        from glibc rand() after srand(137) (re-creatable by oracle/liboracle.so)
"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
DRIVER = os.path.join(HERE, "_ref", "ref_driver")
GOLD = os.path.join(ROOT, "tests", "golden")

LATTICES = [(4, 4, 4, 4), (6, 4, 2, 8)]
CHECKSUM_LATTICES = [(8, 8, 8, 8), (16, 16, 16, 16)]


def main():
    subprocess.check_call(["make", "-C", HERE, "ref"])
    os.makedirs(GOLD, exist_ok=True)
    for X in LATTICES:
        with tempfile.TemporaryDirectory() as d:
            subprocess.check_call([DRIVER, "golden", d] + [str(x) for x in X])
            arrays = {}
            with open(os.path.join(d, "manifest.txt")) as f:
                header = f.readline().split()
                meta = dict(X=[int(v) for v in header[2:6]], kappa=float(header[7]), mu=float(header[9]))
                for line in f:
                    name, n = line.split()
                    a = np.fromfile(os.path.join(d, name + ".f64"), dtype="<f8")
                    assert a.size == int(n)
                    arrays[name] = a
            arrays["meta_X"] = np.array(meta["X"], dtype=np.int32)
            arrays["meta_kappa_mu"] = np.array([meta["kappa"], meta["mu"]])
            out = os.path.join(GOLD, "ref_%dx%dx%dx%d.npz" % X)
            np.savez_compressed(out, **arrays)
            print("wrote", out, "%d arrays, %.2f MB" % (len(arrays), os.path.getsize(out) / 1e6))
    sums = []
    for X in CHECKSUM_LATTICES:
        line = subprocess.check_output([DRIVER, "checksum"] + [str(x) for x in X] + ["3"]).decode()
        sums.append(json.loads(line))
    with open(os.path.join(GOLD, "ref_checksums.json"), "w") as f:
        json.dump(sums, f, indent=1)
    print(json.dumps(sums))


if __name__ == "__main__":
    sys.exit(main())
