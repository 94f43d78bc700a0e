/*
 * oracle/qo_dslash.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE (see qo_fields.h).
 *
 * CPU restatement of the reference host operators
 *   tests/wilson_dslash_reference.cpp, tests/clover_reference.cpp,
 *   tests/dslash_util.h, tests/test_util.cpp (index maps, field constructors),
 *   tests/blas_reference.cpp.
 * Parity pinned by tests/golden/ (generated from the reference itself by
 * oracle/make_golden.py via oracle/_ref/ref_driver).
 */
#include "qo_fields.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static int qo_nthreads = 1;
void qo_set_threads(int n) { qo_nthreads = n > 0 ? n : 1; }
int qo_get_threads(void) { return qo_nthreads; }

/* Spin projector table in the DeGrand-Rossi basis; index = 2*mu + (0: 1-gamma_mu type, 1: 1+gamma_mu type).
 * Values as in tests/wilson_dslash_reference.cpp:21-70 (data, required verbatim for parity). */
static const double qo_projector[8][4][4][2] = {
    {{{1, 0}, {0, 0}, {0, 0}, {0, -1}}, {{0, 0}, {1, 0}, {0, -1}, {0, 0}}, {{0, 0}, {0, 1}, {1, 0}, {0, 0}}, {{0, 1}, {0, 0}, {0, 0}, {1, 0}}},
    {{{1, 0}, {0, 0}, {0, 0}, {0, 1}}, {{0, 0}, {1, 0}, {0, 1}, {0, 0}}, {{0, 0}, {0, -1}, {1, 0}, {0, 0}}, {{0, -1}, {0, 0}, {0, 0}, {1, 0}}},
    {{{1, 0}, {0, 0}, {0, 0}, {1, 0}}, {{0, 0}, {1, 0}, {-1, 0}, {0, 0}}, {{0, 0}, {-1, 0}, {1, 0}, {0, 0}}, {{1, 0}, {0, 0}, {0, 0}, {1, 0}}},
    {{{1, 0}, {0, 0}, {0, 0}, {-1, 0}}, {{0, 0}, {1, 0}, {1, 0}, {0, 0}}, {{0, 0}, {1, 0}, {1, 0}, {0, 0}}, {{-1, 0}, {0, 0}, {0, 0}, {1, 0}}},
    {{{1, 0}, {0, 0}, {0, -1}, {0, 0}}, {{0, 0}, {1, 0}, {0, 0}, {0, 1}}, {{0, 1}, {0, 0}, {1, 0}, {0, 0}}, {{0, 0}, {0, -1}, {0, 0}, {1, 0}}},
    {{{1, 0}, {0, 0}, {0, 1}, {0, 0}}, {{0, 0}, {1, 0}, {0, 0}, {0, -1}}, {{0, -1}, {0, 0}, {1, 0}, {0, 0}}, {{0, 0}, {0, 1}, {0, 0}, {1, 0}}},
    {{{1, 0}, {0, 0}, {-1, 0}, {0, 0}}, {{0, 0}, {1, 0}, {0, 0}, {-1, 0}}, {{-1, 0}, {0, 0}, {1, 0}, {0, 0}}, {{0, 0}, {-1, 0}, {0, 0}, {1, 0}}},
    {{{1, 0}, {0, 0}, {1, 0}, {0, 0}}, {{0, 0}, {1, 0}, {0, 0}, {1, 0}}, {{1, 0}, {0, 0}, {1, 0}, {0, 0}}, {{0, 0}, {1, 0}, {0, 0}, {1, 0}}}};

/* cb index -> full lexicographic index: tests/test_util.cpp:419-443 */
int qo_full_lattice_index(const int X[4], int i, int oddBit) {
  int X1h = X[0] / 2;
  int za = i / X1h;
  int zb = za / X[1];
  int x2 = za - zb * X[1];
  int x4 = zb / X[2];
  int x3 = zb - x4 * X[2];
  int x1odd = (x2 + x3 + x4 + oddBit) & 1;
  return 2 * i + x1odd;
}

/* displaced cb index (lands on the other parity for |d|=1): tests/test_util.cpp:456-471 */
int qo_neighbor_index(const int X[4], int i, int oddBit, int dx4, int dx3, int dx2, int dx1) {
  int Y = qo_full_lattice_index(X, i, oddBit);
  int x4 = Y / (X[2] * X[1] * X[0]);
  int x3 = (Y / (X[1] * X[0])) % X[2];
  int x2 = (Y / X[0]) % X[1];
  int x1 = Y % X[0];
  x4 = (x4 + dx4 + X[3]) % X[3];
  x3 = (x3 + dx3 + X[2]) % X[2];
  x2 = (x2 + dx2 + X[1]) % X[1];
  x1 = (x1 + dx1 + X[0]) % X[0];
  return (x4 * (X[2] * X[1] * X[0]) + x3 * (X[1] * X[0]) + x2 * X[0] + x1) / 2;
}

#define REAL double
#define FN(x) x##_d
#include "qo_dslash_impl.inc"
#undef REAL
#undef FN

#define REAL float
#define FN(x) x##_f
#include "qo_dslash_impl.inc"
#undef REAL
#undef FN

double qo_norm2_d(const double *v, long n) {
  double s = 0.0;
  for (long i = 0; i < n; i++) s += v[i] * v[i];
  return s;
}
void qo_xpay_d(const double *x, double a, double *y, long n) { xpay_d(x, a, y, n); }

/* out = tmpH + i a gamma5 in : tests/clover_reference.cpp:175-186 */
static void apply_twist_d(double *out, const double *in, const double *tmpH, double a, int Vh) {
  for (int i = 0; i < Vh; i++)
    for (int s = 0; s < 4; s++) {
      double a5 = ((s / 2) ? -1.0 : +1.0) * a;
      for (int c = 0; c < 3; c++) {
        long k = (long)i * 24 + s * 6 + c * 2;
        out[k + 0] = tmpH[k + 0] - a5 * in[k + 1];
        out[k + 1] = tmpH[k + 1] + a5 * in[k + 0];
      }
    }
}

/* (A + i a g5) or cinv (A + i a g5): tests/clover_reference.cpp:203-232 */
void qo_twist_clover_gamma5_d(double *out, const double *in, const double *clover, const double *cinv, int dagger,
                              double kappa, double mu, int flavor, int parity, int twist, const int X[4]) {
  const int Vh = X[0] * X[1] * X[2] * X[3] / 2;
  double *tmp1 = (double *)malloc((size_t)Vh * 24 * sizeof(double));
  double *tmp2 = (double *)malloc((size_t)Vh * 24 * sizeof(double));
  double a;
  if (twist == QO_TWIST_DIRECT) {
    a = 2.0 * kappa * mu * flavor;
    if (dagger) a *= -1.0;
    qo_apply_clover_d(tmp1, clover, in, parity, X);
    apply_twist_d(out, in, tmp1, a, Vh);
  } else {
    a = -2.0 * kappa * mu * flavor;
    if (dagger) a *= -1.0;
    qo_apply_clover_d(tmp1, clover, in, parity, X);
    apply_twist_d(tmp2, in, tmp1, a, Vh);
    qo_apply_clover_d(out, cinv, tmp2, parity, X);
  }
  free(tmp2);
  free(tmp1);
}

/* tests/clover_reference.cpp:234-254 */
void qo_tmc_dslash_d(double *out, double *const gauge[4], const double *in, const double *clover, const double *cinv,
                     double kappa, double mu, int flavor, int parity, int matpc, int dagger, const int X[4]) {
  const int Vh = X[0] * X[1] * X[2] * X[3] / 2;
  double *tmp1 = (double *)malloc((size_t)Vh * 24 * sizeof(double));
  double *tmp2 = (double *)malloc((size_t)Vh * 24 * sizeof(double));
  if (dagger) {
    qo_twist_clover_gamma5_d(tmp1, in, clover, cinv, dagger, kappa, mu, flavor, 1 - parity, QO_TWIST_INVERSE, X);
    if (matpc == QO_MATPC_EVEN_EVEN_ASYM || matpc == QO_MATPC_ODD_ODD_ASYM) {
      qo_wil_dslash_d(tmp2, gauge, tmp1, parity, dagger, X);
      qo_twist_clover_gamma5_d(out, tmp2, clover, cinv, dagger, kappa, mu, flavor, parity, QO_TWIST_INVERSE, X);
    } else {
      qo_wil_dslash_d(out, gauge, tmp1, parity, dagger, X);
    }
  } else {
    qo_wil_dslash_d(tmp1, gauge, in, parity, dagger, X);
    qo_twist_clover_gamma5_d(out, tmp1, clover, cinv, dagger, kappa, mu, flavor, parity, QO_TWIST_INVERSE, X);
  }
  free(tmp2);
  free(tmp1);
}

/* tests/clover_reference.cpp:257-281 */
void qo_tmc_mat_d(double *out, double *const gauge[4], const double *clover, const double *in, double kappa, double mu,
                  int flavor, int dagger, const int X[4]) {
  const int Vh = X[0] * X[1] * X[2] * X[3] / 2;
  double *tmp = (double *)malloc((size_t)Vh * 48 * sizeof(double));
  const double *inE = in, *inO = in + (long)Vh * 24;
  double *outE = out, *outO = out + (long)Vh * 24;
  qo_wil_dslash_d(outO, gauge, inE, 1, dagger, X);
  qo_twist_clover_gamma5_d(tmp + (long)Vh * 24, inO, clover, NULL, dagger, kappa, mu, flavor, 1, QO_TWIST_DIRECT, X);
  qo_wil_dslash_d(outE, gauge, inO, 0, dagger, X);
  qo_twist_clover_gamma5_d(tmp, inE, clover, NULL, dagger, kappa, mu, flavor, 0, QO_TWIST_DIRECT, X);
  xpay_d(tmp, -kappa, out, (long)Vh * 48);
  free(tmp);
}

/* tests/clover_reference.cpp:284-341 */
void qo_tmc_matpc_d(double *out, double *const gauge[4], const double *in, const double *clover, const double *cinv,
                    double kappa, double mu, int flavor, int matpc, int dagger, const int X[4]) {
  const int Vh = X[0] * X[1] * X[2] * X[3] / 2;
  const double kappa2 = -kappa * kappa;
  double *tmp1 = (double *)malloc((size_t)Vh * 24 * sizeof(double));
  double *tmp2 = (double *)malloc((size_t)Vh * 24 * sizeof(double));
  /* p0 = parity the operator acts on, p1 = the intermediate parity */
  const int p0 = (matpc == QO_MATPC_EVEN_EVEN || matpc == QO_MATPC_EVEN_EVEN_ASYM) ? 0 : 1;
  const int p1 = 1 - p0;
#define TCG5(o, i, par, tw) qo_twist_clover_gamma5_d(o, i, clover, cinv, dagger, kappa, mu, flavor, par, tw, X)
  if (matpc == QO_MATPC_EVEN_EVEN || matpc == QO_MATPC_ODD_ODD) {
    if (!dagger) {
      qo_wil_dslash_d(out, gauge, in, p1, dagger, X);
      TCG5(tmp1, out, p1, QO_TWIST_INVERSE);
      qo_wil_dslash_d(tmp2, gauge, tmp1, p0, dagger, X);
      TCG5(out, tmp2, p0, QO_TWIST_INVERSE);
    } else {
      TCG5(out, in, p0, QO_TWIST_INVERSE);
      qo_wil_dslash_d(tmp1, gauge, out, p1, dagger, X);
      TCG5(tmp2, tmp1, p1, QO_TWIST_INVERSE);
      qo_wil_dslash_d(out, gauge, tmp2, p0, dagger, X);
    }
    xpay_d(in, kappa2, out, (long)Vh * 24);
  } else {
    qo_wil_dslash_d(tmp1, gauge, in, p1, dagger, X);
    TCG5(tmp2, tmp1, p1, QO_TWIST_INVERSE);
    qo_wil_dslash_d(out, gauge, tmp2, p0, dagger, X);
    if (matpc == QO_MATPC_EVEN_EVEN_ASYM) {
      TCG5(tmp2, in, p0, QO_TWIST_DIRECT);
      xpay_d(tmp2, kappa2, out, (long)Vh * 24);
    } else {
      TCG5(tmp1, in, p0, QO_TWIST_DIRECT);
      xpay_d(tmp1, kappa2, out, (long)Vh * 24);
    }
  }
#undef TCG5
  free(tmp2);
  free(tmp1);
}

/* ------------------------------------------------------------------------------------------
 * Synthetic inputs — the harness side of the reference tests.
 * ------------------------------------------------------------------------------------------ */
void qo_srand(unsigned seed) { srand(seed); }

static void normalize3(double *a) { /* tests/test_util.cpp:862-867 (complex<double>::operator/= by real) */
  double sum = 0.0;
  for (int i = 0; i < 3; i++) sum += a[2 * i] * a[2 * i] + a[2 * i + 1] * a[2 * i + 1];
  for (int i = 0; i < 3; i++) {
    a[2 * i] /= sqrt(sum);
    a[2 * i + 1] /= sqrt(sum);
  }
}
static void orthogonalize3(const double *a, double *b) { /* tests/test_util.cpp:870-875 */
  double dre = 0.0, dim = 0.0;
  for (int i = 0; i < 3; i++) { /* dot += conj(a)*b */
    double are = a[2 * i], aim = -a[2 * i + 1], bre = b[2 * i], bim = b[2 * i + 1];
    dre += are * bre - aim * bim;
    dim += are * bim + aim * bre;
  }
  for (int i = 0; i < 3; i++) { /* b -= dot*a */
    double are = a[2 * i], aim = a[2 * i + 1];
    b[2 * i] -= dre * are - dim * aim;
    b[2 * i + 1] -= dre * aim + dim * are;
  }
}
/* a += sign * conj(b*c) : tests/test_util.cpp accumulateConjugateProduct */
static void acc_conj_prod(double *a, const double *b, const double *c, int sign) {
  a[0] += sign * (b[0] * c[0] - b[1] * c[1]);
  a[1] -= sign * (b[0] * c[1] + b[1] * c[0]);
}

/* random SU(3): rows 1,2 uniform(0,1) complex drawn interleaved even/odd per element
 * (tests/test_util.cpp:879-956); scaling + anti-periodic T (:683-706). */
void qo_construct_gauge_field_d(double *const gauge[4], const int X[4], double anisotropy, int antiperiodic_t) {
  const int Vh = X[0] * X[1] * X[2] * X[3] / 2;
  for (int dir = 0; dir < 4; dir++) {
    double *ev = gauge[dir], *od = gauge[dir] + (long)Vh * 18;
    for (int i = 0; i < Vh; i++) {
      for (int m = 1; m < 3; m++)
        for (int n = 0; n < 3; n++) {
          ev[i * 18 + m * 6 + n * 2 + 0] = rand() / (double)RAND_MAX;
          ev[i * 18 + m * 6 + n * 2 + 1] = rand() / (double)RAND_MAX;
          od[i * 18 + m * 6 + n * 2 + 0] = rand() / (double)RAND_MAX;
          od[i * 18 + m * 6 + n * 2 + 1] = rand() / (double)RAND_MAX;
        }
      double *half[2] = {ev, od};
      for (int h = 0; h < 2; h++) {
        double *u = half[h] + (i * 3 + 1) * 6, *v = half[h] + (i * 3 + 2) * 6;
        normalize3(u);
        orthogonalize3(u, v);
        normalize3(v);
      }
      for (int h = 0; h < 2; h++) {
        double *w = half[h] + (i * 3 + 0) * 6, *u = half[h] + (i * 3 + 1) * 6, *v = half[h] + (i * 3 + 2) * 6;
        for (int n = 0; n < 6; n++) w[n] = 0.0;
        acc_conj_prod(w + 0, u + 2, v + 4, +1);
        acc_conj_prod(w + 0, u + 4, v + 2, -1);
        acc_conj_prod(w + 2, u + 4, v + 0, +1);
        acc_conj_prod(w + 2, u + 0, v + 4, -1);
        acc_conj_prod(w + 4, u + 0, v + 2, +1);
        acc_conj_prod(w + 4, u + 2, v + 0, -1);
      }
    }
  }
  for (int d = 0; d < 3; d++)
    for (long i = 0; i < 18L * Vh * 2; i++) gauge[d][i] /= anisotropy;
  if (antiperiodic_t) {
    for (int j = (X[0] / 2) * X[1] * X[2] * (X[3] - 1); j < Vh; j++)
      for (int i = 0; i < 18; i++) {
        gauge[3][(long)j * 18 + i] *= -1.0;
        gauge[3][((long)Vh + j) * 18 + i] *= -1.0;
      }
  }
}

void qo_construct_clover_field_d(double *res, int V, double norm, double diag) {
  double c = 2.0 * norm / RAND_MAX;
  for (int i = 0; i < V; i++) {
    for (int j = 0; j < 72; j++) res[(long)i * 72 + j] = c * rand() - norm;
    for (int j = 0; j < 6; j++) {
      res[(long)i * 72 + j] += diag;
      res[(long)i * 72 + j + 36] += diag;
    }
  }
}

void qo_construct_spinor_field_d(double *spinor, int nreal) {
  for (int i = 0; i < nreal; i++) spinor[i] = rand() / (double)RAND_MAX;
}

/* packed chiral block -> dense Hermitian 6x6 (M(row,col), row>col from L) */
static void unpack_block(const double *blk, double M[6][6][2]) {
  const double *L = blk + 6;
  for (int r = 0; r < 6; r++)
    for (int c = 0; c < 6; c++) {
      if (r == c) { M[r][c][0] = blk[r]; M[r][c][1] = 0.0; }
      else if (r > c) {
        int k = 15 - (6 - c) * (5 - c) / 2 + r - c - 1;
        M[r][c][0] = L[2 * k]; M[r][c][1] = L[2 * k + 1];
      } else {
        int k = 15 - (6 - r) * (5 - r) / 2 + c - r - 1;
        M[r][c][0] = L[2 * k]; M[r][c][1] = -L[2 * k + 1];
      }
    }
}
static void pack_block(double M[6][6][2], double *blk) {
  double *L = blk + 6;
  for (int r = 0; r < 6; r++) blk[r] = M[r][r][0];
  for (int c = 0; c < 6; c++)
    for (int r = c + 1; r < 6; r++) {
      int k = 15 - (6 - c) * (5 - c) / 2 + r - c - 1;
      L[2 * k] = M[r][c][0]; L[2 * k + 1] = M[r][c][1];
    }
}

void qo_clover_twisted_inverse_d(double *cinv, const double *clover, int V, double mu2) {
  for (long b = 0; b < 2L * V; b++) {
    double A[6][6][2], S[6][12][2];
    unpack_block(clover + b * 36, A);
    /* S = [A*A + mu2 | I] */
    for (int r = 0; r < 6; r++)
      for (int c = 0; c < 6; c++) {
        double re = 0, im = 0;
        for (int k = 0; k < 6; k++) {
          re += A[r][k][0] * A[k][c][0] - A[r][k][1] * A[k][c][1];
          im += A[r][k][0] * A[k][c][1] + A[r][k][1] * A[k][c][0];
        }
        S[r][c][0] = re + (r == c ? mu2 : 0.0);
        S[r][c][1] = im;
        S[r][c + 6][0] = (r == c); S[r][c + 6][1] = 0.0;
      }
    /* Gauss-Jordan with partial pivoting */
    for (int p = 0; p < 6; p++) {
      int best = p; double bm = 0;
      for (int r = p; r < 6; r++) {
        double m = S[r][p][0] * S[r][p][0] + S[r][p][1] * S[r][p][1];
        if (m > bm) { bm = m; best = r; }
      }
      if (best != p)
        for (int c = 0; c < 12; c++)
          for (int z = 0; z < 2; z++) { double t = S[p][c][z]; S[p][c][z] = S[best][c][z]; S[best][c][z] = t; }
      double pr = S[p][p][0], pi = S[p][p][1], den = pr * pr + pi * pi;
      double ir = pr / den, ii = -pi / den;
      for (int c = 0; c < 12; c++) {
        double re = S[p][c][0] * ir - S[p][c][1] * ii, im = S[p][c][0] * ii + S[p][c][1] * ir;
        S[p][c][0] = re; S[p][c][1] = im;
      }
      for (int r = 0; r < 6; r++) {
        if (r == p) continue;
        double fr = S[r][p][0], fi = S[r][p][1];
        for (int c = 0; c < 12; c++) {
          S[r][c][0] -= fr * S[p][c][0] - fi * S[p][c][1];
          S[r][c][1] -= fr * S[p][c][1] + fi * S[p][c][0];
        }
      }
    }
    double Inv[6][6][2];
    for (int r = 0; r < 6; r++)
      for (int c = 0; c < 6; c++) { Inv[r][c][0] = S[r][c + 6][0]; Inv[r][c][1] = S[r][c + 6][1]; }
    /* symmetrise to exactly Hermitian before packing */
    for (int r = 0; r < 6; r++) {
      Inv[r][r][1] = 0.0;
      for (int c = 0; c < r; c++) {
        double re = 0.5 * (Inv[r][c][0] + Inv[c][r][0]), im = 0.5 * (Inv[r][c][1] - Inv[c][r][1]);
        Inv[r][c][0] = re; Inv[r][c][1] = im; Inv[c][r][0] = re; Inv[c][r][1] = -im;
      }
    }
    pack_block(Inv, cinv + b * 36);
  }
}

/* ------------------------------------------------------------------------------------------
 * Grid-decomposed host operator: the reference's MULTI_GPU branch
 * (tests/wilson_dslash_reference.cpp:137-171; ghost lookup tests/dslash_util.h:214-394).
 * ghost_gauge[d]: the -d neighbour's last slice of U_d, [parity][face index][18];
 * fwd_ghost[d] / back_ghost[d]: full (unprojected) spinors of the +d / -d neighbour's first / last
 * slice, input parity, [face index][24].  Face index = lexicographic index over the other three
 * coordinates, halved.  partitioned[d] = 0 -> periodic wrap inside this rank.
 * ------------------------------------------------------------------------------------------ */
static int face_index(const int X[4], int d, int x1, int x2, int x3, int x4) {
  switch (d) {
    case 0: return ((x4 * X[2] + x3) * X[1] + x2) / 2;
    case 1: return ((x4 * X[2] + x3) * X[0] + x1) / 2;
    case 2: return ((x4 * X[1] + x2) * X[0] + x1) / 2;
    default: return ((x3 * X[1] + x2) * X[0] + x1) / 2;
  }
}

void qo_wil_dslash_halo_d(double *res, double *const gauge[4], double *const ghost_gauge[4], const double *in,
                          double *const fwd_ghost[4], double *const back_ghost[4], int oddBit, int dagger, const int X[4],
                          const int partitioned[4]) {
  const int Vh = X[0] * X[1] * X[2] * X[3] / 2;
  for (long k = 0; k < (long)Vh * 24; k++) res[k] = 0.0;
  for (int i = 0; i < Vh; i++) {
    const int Y = qo_full_lattice_index(X, i, oddBit);
    const int c[4] = {Y % X[0], (Y / X[0]) % X[1], (Y / (X[1] * X[0])) % X[2], Y / (X[2] * X[1] * X[0])};
    for (int dir = 0; dir < 8; dir++) {
      const int d = dir / 2, fwd = (dir % 2 == 0);
      const int f = face_index(X, d, c[0], c[1], c[2], c[3]);
      const double *U, *psi;
      if (fwd) {
        U = gauge[d] + ((long)oddBit * Vh + i) * 18;
        if (partitioned[d] && c[d] == X[d] - 1) psi = fwd_ghost[d] + (long)f * 24;
        else psi = in + (long)nbr_of_d(X, i, dir, oddBit) * 24;
      } else {
        if (partitioned[d] && c[d] == 0) {
          const int faceCB = Vh / X[d];
          U = ghost_gauge[d] + ((long)(1 - oddBit) * faceCB + f) * 18;
          psi = back_ghost[d] + (long)f * 24;
        } else {
          U = link_of_d(X, Vh, i, dir, oddBit, gauge);
          psi = in + (long)nbr_of_d(X, i, dir, oddBit) * 24;
        }
      }
      double proj[24], hop[24];
      const int projIdx = 2 * (dir / 2) + (dir + dagger) % 2;
      project_d(proj, projIdx, psi);
      for (int s = 0; s < 4; s++) {
        if (fwd) su3_mul_d(&hop[s * 6], U, &proj[s * 6]);
        else su3_tmul_d(&hop[s * 6], U, &proj[s * 6]);
      }
      double *o = res + (long)i * 24;
      for (int k = 0; k < 24; k++) o[k] = o[k] + hop[k];
    }
  }
}
