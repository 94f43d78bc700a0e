#!/usr/bin/env python3
"""Turn rocprofv3 output directories (kernel-trace --stats, and separate --pmc FETCH_SIZE / WRITE_SIZE passes) under
gpurun_out/ into the committed summaries under profiles/.

usage: summarize_profiles.py <tag> <stats_dir> <pmc_fetch_dir> <pmc_write_dir>

FETCH_SIZE / WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE counts 16-byte-per-lane streaming reads at HALF their bytes
(MI355X_MICROARCH.md, HBM section); the factor is calibrated in-run on blas_kernel<double,2,false,Norm2F>, which reads
exactly 12 x Vh x 16 B, and applied to the stencil kernel's reads.  WRITE_SIZE is exact for 16-byte stores."""
import collections
import csv
import glob
import json
import os
import shutil
import sys


def counters(d):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    agg = collections.defaultdict(list)
    for path in f:
        for r in csv.DictReader(open(path)):
            agg[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
    return agg


def main():
    tag, stats_dir, fetch_dir, write_dir = sys.argv[1:5]
    os.makedirs("profiles", exist_ok=True)
    st = glob.glob(os.path.join(stats_dir, "**", "*kernel_stats.csv"), recursive=True)
    if st:
        shutil.copy(st[0], "profiles/%s_kernel_stats.csv" % tag)
    fetch, write = counters(fetch_dir), counters(write_dir)
    out = {"note": __doc__.split("\n\n")[2]}
    calib = None
    for (k, c), v in fetch.items():
        if "Norm2F" in k and "double" in k and c == "FETCH_SIZE":
            calib = sum(v) / len(v)
    vh = int(os.environ.get("VH", 524288))
    factor = (12 * vh * 16 / 1024.0) / calib if calib else 2.0
    out["fetch_calibration"] = {"kernel": "blas_kernel<double,2,false,Norm2F>", "true_KiB": 12 * vh * 16 / 1024.0, "reported_KiB": calib, "factor": factor}
    kernels = {}
    for (k, c), v in list(fetch.items()) + list(write.items()):
        if "dslash_kernel" not in k:
            continue
        e = kernels.setdefault(k, {})
        e[c] = {"dispatches": len(v), "mean_KiB": sum(v) / len(v)}
    for k, e in kernels.items():
        if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
            rd = e["FETCH_SIZE"]["mean_KiB"] * factor * 1024
            wr = e["WRITE_SIZE"]["mean_KiB"] * 1024
            e["hbm_read_bytes_per_launch_corrected"] = rd
            e["hbm_write_bytes_per_launch"] = wr
            e["hbm_bytes_per_launch"] = rd + wr
    out["kernels"] = kernels
    json.dump(out, open("profiles/%s_hbm_traffic.json" % tag, "w"), indent=1)
    print(json.dumps(out, indent=1)[:1500])


if __name__ == "__main__":
    main()
