"""ctypes binding of oracle/liboracle.so — the CPU restatement of the reference host operators.

TEST INFRASTRUCTURE: importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg only.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ODIR = os.path.join(ROOT, "oracle")

MATPC = {"ee": 0, "oo": 1, "eeasym": 2, "ooasym": 3}
TWIST_DIRECT, TWIST_INVERSE = 0, 1

_dp = C.POINTER(C.c_double)
_fp = C.POINTER(C.c_float)


def _p(a):
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_dp if a.dtype == np.float64 else _fp)


def _g(gauge):
    """gauge: (4, V*18) array -> C array of 4 row pointers"""
    t = _dp if gauge.dtype == np.float64 else _fp
    arr = (t * 4)()
    for d in range(4):
        arr[d] = gauge[d].ctypes.data_as(t)
    return arr


def _x(X):
    return (C.c_int * 4)(*[int(v) for v in X])


class Oracle:
    def __init__(self, lib):
        self.lib = lib
        lib.qo_norm2_d.restype = C.c_double
        lib.qo_neighbor_index.restype = C.c_int
        lib.qo_full_lattice_index.restype = C.c_int

    def set_threads(self, n):
        self.lib.qo_set_threads(C.c_int(n))

    # -- geometry
    def neighbor_index(self, X, i, odd, dx4, dx3, dx2, dx1):
        return self.lib.qo_neighbor_index(_x(X), i, odd, dx4, dx3, dx2, dx1)

    def full_index(self, X, i, odd):
        return self.lib.qo_full_lattice_index(_x(X), i, odd)

    # -- operators (fp64 unless the arrays are float32)
    def _sfx(self, a):
        return "_d" if a.dtype == np.float64 else "_f"

    def wil_dslash(self, gauge, inp, X, parity, dagger):
        out = np.empty_like(inp)
        getattr(self.lib, "qo_wil_dslash" + self._sfx(inp))(_p(out), _g(gauge), _p(inp), parity, dagger, _x(X))
        return out

    def twist_gamma5(self, inp, kappa, mu, flavor, dagger, twist):
        out = np.empty_like(inp)
        T = C.c_double if inp.dtype == np.float64 else C.c_float
        getattr(self.lib, "qo_twist_gamma5" + self._sfx(inp))(_p(out), _p(inp), dagger, T(kappa), T(mu), flavor,
                                                               inp.size // 24, twist)
        return out

    def tm_dslash(self, gauge, inp, X, kappa, mu, flavor, parity, matpc, dagger):
        inp = inp.copy()  # the reference mutates-and-restores its input
        out = np.empty_like(inp)
        getattr(self.lib, "qo_tm_dslash" + self._sfx(inp))(_p(out), _g(gauge), _p(inp), C.c_double(kappa), C.c_double(mu),
                                                            flavor, parity, MATPC[matpc], dagger, _x(X))
        return out

    def tm_matpc(self, gauge, inp, X, kappa, mu, flavor, matpc, dagger):
        inp = inp.copy()
        out = np.empty_like(inp)
        getattr(self.lib, "qo_tm_matpc" + self._sfx(inp))(_p(out), _g(gauge), _p(inp), C.c_double(kappa), C.c_double(mu),
                                                           flavor, MATPC[matpc], dagger, _x(X))
        return out

    def tm_mat(self, gauge, inp, X, kappa, mu, flavor, dagger):
        out = np.empty_like(inp)
        getattr(self.lib, "qo_tm_mat" + self._sfx(inp))(_p(out), _g(gauge), _p(inp), C.c_double(kappa), C.c_double(mu),
                                                         flavor, dagger, _x(X))
        return out

    def gcr_tm(self, gauge, b, X, kappa, mu, flavor, tol=1e-10, nkrylov=20, maxiter=5000):
        """plain restarted GCR on tm_mat (oracle/qo_solver.c): returns (x, iterations, seconds, true residual)"""
        b = np.ascontiguousarray(b, dtype=np.float64)
        x = np.zeros_like(b)
        secs, res = C.c_double(0), C.c_double(0)
        self.lib.qo_gcr_tm_d.restype = C.c_int
        it = self.lib.qo_gcr_tm_d(_p(x), _g(gauge), _p(b), _x(X), C.c_double(kappa), C.c_double(mu), int(flavor), C.c_double(tol), int(nkrylov), int(maxiter),
                                  C.byref(secs), C.byref(res))
        return x, int(it), secs.value, res.value

    def wil_mat(self, gauge, inp, X, kappa, dagger):
        out = np.empty_like(inp)
        self.lib.qo_wil_mat_d(_p(out), _g(gauge), _p(inp), C.c_double(kappa), dagger, _x(X))
        return out

    def wil_matpc(self, gauge, inp, X, kappa, matpc, dagger):
        out = np.empty_like(inp)
        self.lib.qo_wil_matpc_d(_p(out), _g(gauge), _p(inp), C.c_double(kappa), MATPC[matpc], dagger, _x(X))
        return out

    def apply_clover(self, clover, inp, X, parity):
        out = np.empty_like(inp)
        getattr(self.lib, "qo_apply_clover" + self._sfx(inp))(_p(out), _p(clover), _p(inp), parity, _x(X))
        return out

    def twist_clover_gamma5(self, inp, clover, cinv, X, kappa, mu, flavor, parity, dagger, twist):
        out = np.empty_like(inp)
        self.lib.qo_twist_clover_gamma5_d(_p(out), _p(inp), _p(clover), _p(cinv) if cinv is not None else None, dagger,
                                          C.c_double(kappa), C.c_double(mu), flavor, parity, twist, _x(X))
        return out

    def tmc_dslash(self, gauge, inp, clover, cinv, X, kappa, mu, flavor, parity, matpc, dagger):
        out = np.empty_like(inp)
        self.lib.qo_tmc_dslash_d(_p(out), _g(gauge), _p(inp), _p(clover), _p(cinv), C.c_double(kappa), C.c_double(mu),
                                 flavor, parity, MATPC[matpc], dagger, _x(X))
        return out

    def tmc_matpc(self, gauge, inp, clover, cinv, X, kappa, mu, flavor, matpc, dagger):
        out = np.empty_like(inp)
        self.lib.qo_tmc_matpc_d(_p(out), _g(gauge), _p(inp), _p(clover), _p(cinv), C.c_double(kappa), C.c_double(mu),
                                flavor, MATPC[matpc], dagger, _x(X))
        return out

    def tmc_mat(self, gauge, clover, inp, X, kappa, mu, flavor, dagger):
        out = np.empty_like(inp)
        self.lib.qo_tmc_mat_d(_p(out), _g(gauge), _p(clover), _p(inp), C.c_double(kappa), C.c_double(mu), flavor, dagger,
                              _x(X))
        return out

    # -- synthetic inputs (glibc rand(), as the reference harness)
    def make_fields(self, X, seed=137, antiperiodic_t=True, clover=True):
        V = int(np.prod(X))
        self.lib.qo_srand(C.c_uint(seed))
        gauge = np.empty((4, V * 18))
        self.lib.qo_construct_gauge_field_d(_g(gauge), _x(X), C.c_double(1.0), int(antiperiodic_t))
        spinor = np.empty(V * 24)
        self.lib.qo_construct_spinor_field_d(_p(spinor), spinor.size)
        clv = None
        if clover:
            clv = np.empty(V * 72)
            self.lib.qo_construct_clover_field_d(_p(clv), V, C.c_double(0.1), C.c_double(1.0))
        return gauge, spinor, clv

    def clover_twisted_inverse(self, clover, mu2):
        out = np.empty_like(clover)
        self.lib.qo_clover_twisted_inverse_d(_p(out), _p(clover), clover.size // 72, C.c_double(mu2))
        return out

    def norm2(self, a):
        return self.lib.qo_norm2_d(_p(a), C.c_long(a.size))

    def clover_compute(self, gauge, coeff, X):
        """clover term from the gauge field, host packed order (oracle/qo_mg.c: computeFmunu + computeClover)"""
        out = np.zeros(int(np.prod(X)) * 72)
        self.lib.qo_clover_compute_d(_p(out), _g(gauge), C.c_double(coeff), _x(X))
        return out

    # -- QKXTM source preparation (oracle/qo_qkxtm.c); lexicographic QKXTM host layouts
    def gauss_smear(self, vec_lex, gauge_lex, X, alpha, nsmear):
        """vec_lex: (V*24,) lexicographic spin-colour vector, gauge_lex: (4, V*18) lexicographic links"""
        out = np.zeros_like(vec_lex)
        self.lib.qo_gauss_smear(_p(out), _p(np.ascontiguousarray(vec_lex)), _g(gauge_lex), _x(X), C.c_double(alpha), C.c_int(nsmear))
        return out

    def eo_to_lex(self, eo, X, n):
        out = np.zeros_like(eo)
        self.lib.qo_eo_to_lex(_p(out), _p(np.ascontiguousarray(eo)), _x(X), C.c_int(n))
        return out

    def lex_to_eo(self, lex, X, n):
        out = np.zeros_like(lex)
        self.lib.qo_lex_to_eo(_p(out), _p(np.ascontiguousarray(lex)), _x(X), C.c_int(n))
        return out

    def ape_smear(self, gauge, X, alpha, nsteps):
        """gauge: (4, V*18) QDP even-odd order -> APE-smeared links in the same order (oracle/qo_qkxtm.c)"""
        out = np.zeros_like(gauge)
        self.lib.qo_ape_smear(_g(out), _g(np.ascontiguousarray(gauge)), _x(X), C.c_double(alpha), C.c_int(nsteps))
        return out

    def plaquette(self, gauge, X):
        pl = np.zeros(3)
        self.lib.qo_plaquette(_p(pl), _g(np.ascontiguousarray(gauge)), _x(X))
        return pl

    @staticmethod
    def ukqcd_to_dr(v):
        """host spinor(s) (..., 24) UKQCD -> DeGrand-Rossi: the reference's RelBasis (lib/copy_color_spinor.cuh:73-91)"""
        k = 1.0 / np.sqrt(2.0)
        a = np.asarray(v).reshape(-1, 4, 6)
        o = np.empty_like(a)
        o[:, 0] = -k * (a[:, 1] + a[:, 3]); o[:, 1] = k * (a[:, 0] + a[:, 2])
        o[:, 2] = k * (a[:, 3] - a[:, 1]); o[:, 3] = k * (a[:, 0] - a[:, 2])
        return o.reshape(np.asarray(v).shape)

    @staticmethod
    def dr_to_ukqcd(v):
        """DeGrand-Rossi -> UKQCD: the reference's NonRelBasis (lib/copy_color_spinor.cuh:49-70)"""
        k = 1.0 / np.sqrt(2.0)
        a = np.asarray(v).reshape(-1, 4, 6)
        o = np.empty_like(a)
        o[:, 0] = k * (a[:, 1] + a[:, 3]); o[:, 1] = -k * (a[:, 0] + a[:, 2])
        o[:, 2] = k * (a[:, 1] - a[:, 3]); o[:, 3] = k * (a[:, 2] - a[:, 0])
        return o.reshape(np.asarray(v).shape)

    # -- multigrid pieces (oracle/qo_mg.c); complex128 arrays in the reference CPU orders
    @staticmethod
    def _c(a):
        a = np.ascontiguousarray(a, dtype=np.complex128)
        return a, a.ctypes.data_as(_dp)

    def mg_block_orthogonalize(self, V, X, geo_bs, Ns, Nc, Nvec, spin_bs):
        V, pv = self._c(np.array(V, dtype=np.complex128))
        self.lib.qo_mg_block_orthogonalize(pv, _x(X), _x(geo_bs), Ns, Nc, Nvec, spin_bs)
        return V

    def mg_restrict(self, inp, V, X, geo_bs, Ns, Nc, Nvec, spin_bs):
        Vc = int(np.prod(X)) // int(np.prod(geo_bs))
        out = np.zeros((Vc, Ns // spin_bs, Nvec), dtype=np.complex128)
        inp, pi = self._c(inp)
        V, pv = self._c(V)
        self.lib.qo_mg_restrict(out.ctypes.data_as(_dp), pi, pv, _x(X), _x(geo_bs), Ns, Nc, Nvec, spin_bs)
        return out

    def mg_prolongate(self, inp, V, X, geo_bs, Ns, Nc, Nvec, spin_bs):
        out = np.zeros((int(np.prod(X)), Ns, Nc), dtype=np.complex128)
        inp, pi = self._c(inp)
        V, pv = self._c(V)
        self.lib.qo_mg_prolongate(out.ctypes.data_as(_dp), pi, pv, _x(X), _x(geo_bs), Ns, Nc, Nvec, spin_bs)
        return out

    def mg_coarse_op_fine(self, V, gauge, clover, kappa, mu_tilde, X, geo_bs, Nvec):
        Vc, n = int(np.prod(X)) // int(np.prod(geo_bs)), 2 * Nvec
        Y = np.zeros((8, Vc, n, n), dtype=np.complex128)
        Xm = np.zeros((Vc, n, n), dtype=np.complex128)
        V, pv = self._c(V)
        self.lib.qo_mg_coarse_op_fine(Y.ctypes.data_as(_dp), Xm.ctypes.data_as(_dp), pv, _g(gauge), _p(clover) if clover is not None else None,
                                      C.c_double(kappa), C.c_double(mu_tilde), _x(X), _x(geo_bs), Nvec)
        return Y, Xm

    def mg_coarse_op_coarse(self, V, Yf, Xf, kappa, X, geo_bs, NcF, Nvec):
        Vc, n = int(np.prod(X)) // int(np.prod(geo_bs)), 2 * Nvec
        Y = np.zeros((8, Vc, n, n), dtype=np.complex128)
        Xm = np.zeros((Vc, n, n), dtype=np.complex128)
        V, pv = self._c(V)
        Yf, py = self._c(Yf)
        Xf, px = self._c(Xf)
        self.lib.qo_mg_coarse_op_coarse(Y.ctypes.data_as(_dp), Xm.ctypes.data_as(_dp), pv, py, px, C.c_double(kappa), _x(X), _x(geo_bs), NcF, Nvec)
        return Y, Xm

    def mg_coarse_apply(self, inp, Y, Xm, kappa, Xc, Nvec):
        inp, pi = self._c(inp)
        Y, py = self._c(Y)
        Xm, px = self._c(Xm)
        out = np.zeros((int(np.prod(Xc)), 2, Nvec), dtype=np.complex128)
        self.lib.qo_mg_coarse_apply(out.ctypes.data_as(_dp), pi, py, px, C.c_double(kappa), _x(Xc), Nvec)
        return out


def build():
    subprocess.check_call(["make", "-s", "-C", ODIR, "liboracle.so"])


def load():
    path = os.path.join(ODIR, "liboracle.so")
    if not os.path.exists(path):
        build()
    return Oracle(C.CDLL(path))
