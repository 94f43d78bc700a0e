"""Shared description of the golden cases (tests/golden/ref_*.npz): parse a case name and run it either on
the oracle (CPU restatement) or through the library's C ABI."""
import glob
import importlib
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FILES = sorted(glob.glob(os.path.join(GOLD, "ref_*x*.npz")))
FL = {"fp": +1, "fm": -1}
P0 = {"ee": 0, "eeasym": 0, "oo": 1, "ooasym": 1}


def load(path):
    z = np.load(path)
    X = [int(v) for v in z["meta_X"]]
    kappa, mu = [float(v) for v in z["meta_kappa_mu"]]
    gauge = np.stack([z["gauge%d" % d] for d in range(4)])
    return z, X, kappa, mu, gauge


def case_names(z):
    skip = ("gauge", "spinor", "clover", "meta")
    return [n for n in z.files if not n.startswith(skip)]


def run_abi(qa, name, spinor, X, kappa, mu, prec, host_dtype=np.float64):
    """Run golden case `name` through dslashQuda / MatQuda / cloverQuda.  Gauge (and clover) must be resident."""
    t = name.split("_")
    nh = spinor.size // 2
    cpu_prec = qa.QUDA_DOUBLE_PRECISION if host_dtype == np.float64 else qa.QUDA_SINGLE_PRECISION
    sp = spinor.astype(host_dtype)

    def ip(dslash_type, flavor=+1, matpc="ee", dagger=0, sol=qa.QUDA_MATPC_SOLUTION):
        return qa.invert_param(dslash_type, kappa, mu, flavor, matpc, dagger, cpu_prec=cpu_prec, cuda_prec=prec, solution_type=sol)

    if name.startswith("wil_dslash"):
        return qa.dslash(sp[:nh].copy(), ip(qa.QUDA_WILSON_DSLASH, dagger=int(t[3][1])), int(t[2][1]))
    if name.startswith("apply_clover"):
        par = C_int(int(t[2][1]))
        out = np.empty_like(sp[:nh])
        p = ip(qa.QUDA_TWISTED_CLOVER_DSLASH)
        import ctypes as C
        qa.lib().cloverQuda(out.ctypes.data_as(C.c_void_p), sp[:nh].copy().ctypes.data_as(C.c_void_p), C.byref(p), C.byref(par), 0)
        return out
    if name.startswith("wil_matpc"):
        return qa.mat(sp[:nh].copy(), ip(qa.QUDA_WILSON_DSLASH, matpc=t[2], dagger=int(t[3][1])))
    if name.startswith("wil_mat"):
        return qa.mat(sp.copy(), ip(qa.QUDA_WILSON_DSLASH, dagger=int(t[2][1]), sol=qa.QUDA_MAT_SOLUTION))
    kind = qa.QUDA_TWISTED_MASS_DSLASH if t[0] == "tm" else qa.QUDA_TWISTED_CLOVER_DSLASH
    if t[1] == "dslash":
        return qa.dslash(sp[:nh].copy(), ip(kind, FL[t[2]], t[3], int(t[4][1])), int(t[5][1]))
    if t[1] == "matpc":
        p0 = P0[t[3]]
        return qa.mat(sp[p0 * nh:(p0 + 1) * nh].copy(), ip(kind, FL[t[2]], t[3], int(t[4][1])))
    if t[1] == "mat":
        return qa.mat(sp.copy(), ip(kind, FL[t[2]], "ee", int(t[3][1]), sol=qa.QUDA_MAT_SOLUTION))
    raise KeyError(name)


def C_int(v):
    import ctypes as C
    return C.c_int(v)


def rel_err(got, want):
    """max over sites of |got - want| / max|want| — the per-site deviation bar of BASELINE.json"""
    return float(np.max(np.abs(got.astype(np.float64) - want)) / np.max(np.abs(want)))
