/* qo_qkxtm.c — TEST INFRASTRUCTURE (CPU oracle; never linked into or called by the product).
 *
 * Restatement of the source preparation that sits in front of the solve loop of the QKXTM drivers
 * (SURVEY 8f row 1): Gaussian (Wuppertal) smearing of a spin-colour vector with the APE-smeared links,
 *   lib/qudaQKXTM_Vector_Kepler.cpp:386-421 (iteration loop, ping-pong between two vectors),
 *   lib/code_pieces_Kepler/Gauss_core_Kepler.h:1-230 (one step: three spatial directions, forward link times
 *   forward neighbour plus daggered backward link times backward neighbour, then (psi + alpha*sum)/(1+6 alpha)),
 *   lib/code_pieces_Kepler/core_def_Kepler.h:498-530 (apply_U_on_S / apply_U_DAG_on_S: row-major 3x3 times colour).
 * Data layouts are the HOST layouts of the QKXTM classes: sites in lexicographic order x fastest
 * (LEXIC, iv = ((t*Lz+z)*Ly+y)*Lx+x), vector iv*24 + (spin*3 + colour)*2 + re/im
 * (lib/qudaQKXTM_Vector_Kepler.cpp:72-81), links gauge[dir][iv*18 + (c1*3 + c2)*2 + re/im]
 * (lib/qudaQKXTM_Gauge_Kepler.cpp:73-89).  Periodic in all directions (the kernel applies no boundary phase).
 *
 * PARITY UNPINNED as a standalone operation: the reference implementation is a CUDA kernel (needs nvcc) and its tests
 * hold no vectors for it; tests/test_oracle_qkxtm.py checks this restatement against an independent numpy
 * formulation and against exact properties (gauge covariance, unit-gauge diffusion weights). */
#include <stdlib.h>
#include <string.h>

static void mv(double *o, const double *U, const double *s) { /* o += U s, one colour vector */
  for (int a = 0; a < 3; a++) {
    double re = 0, im = 0;
    for (int b = 0; b < 3; b++) {
      const double ur = U[(a * 3 + b) * 2], ui = U[(a * 3 + b) * 2 + 1];
      re += ur * s[2 * b] - ui * s[2 * b + 1];
      im += ur * s[2 * b + 1] + ui * s[2 * b];
    }
    o[2 * a] += re; o[2 * a + 1] += im;
  }
}
static void mdv(double *o, const double *U, const double *s) { /* o += U^dagger s */
  for (int a = 0; a < 3; a++) {
    double re = 0, im = 0;
    for (int b = 0; b < 3; b++) {
      const double ur = U[(b * 3 + a) * 2], ui = -U[(b * 3 + a) * 2 + 1];
      re += ur * s[2 * b] - ui * s[2 * b + 1];
      im += ur * s[2 * b + 1] + ui * s[2 * b];
    }
    o[2 * a] += re; o[2 * a + 1] += im;
  }
}

/* one smearing step (Gauss_core_Kepler.h) */
static void gauss_step(double *out, const double *in, const double *const gauge[4], const int X[4], double alpha) {
  const long V = (long)X[0] * X[1] * X[2] * X[3];
  const double normalize = 1.0 / (1.0 + 6.0 * alpha);
#pragma omp parallel for
  for (long iv = 0; iv < V; iv++) {
    int c[4];
    long l = iv;
    for (int d = 0; d < 4; d++) { c[d] = (int)(l % X[d]); l /= X[d]; }
    double tmp[24];
    memset(tmp, 0, sizeof(tmp));
    long stride = 1;
    for (int mu = 0; mu < 3; mu++) {
      const long plus = iv + ((c[mu] + 1) % X[mu] - c[mu]) * stride;
      const long minus = iv + ((c[mu] - 1 + X[mu]) % X[mu] - c[mu]) * stride;
      for (int s = 0; s < 4; s++) {
        mv(tmp + 6 * s, gauge[mu] + iv * 18, in + plus * 24 + 6 * s);
        mdv(tmp + 6 * s, gauge[mu] + minus * 18, in + minus * 24 + 6 * s);
      }
      stride *= X[mu];
    }
    for (int k = 0; k < 24; k++) out[iv * 24 + k] = normalize * (in[iv * 24 + k] + alpha * tmp[k]);
  }
}

/* out = smear^nsmear(in); in is left untouched (QKXTM_Vector_Kepler::gaussianSmearing overwrites both) */
void qo_gauss_smear(double *out, const double *in, const double *const gauge[4], const int X[4], double alpha, int nsmear) {
  const long n = (long)X[0] * X[1] * X[2] * X[3] * 24;
  double *a = (double *)malloc(n * sizeof(double)), *b = (double *)malloc(n * sizeof(double));
  memcpy(a, in, n * sizeof(double));
  for (int i = 0; i < nsmear; i++) {
    gauss_step(b, a, gauge, X, alpha);
    double *t = a; a = b; b = t;
  }
  memcpy(out, a, n * sizeof(double));
  free(a); free(b);
}

/* QUDA host order (even sites then odd, checkerboard index i = lexicographic/2; tests/test_util.cpp:419-443) <-> QKXTM
 * lexicographic order, n reals per site */
void qo_eo_to_lex(double *lex, const double *eo, const int X[4], int n) {
  const long V = (long)X[0] * X[1] * X[2] * X[3], Vh = V / 2;
  for (long iv = 0; iv < V; iv++) {
    long l = iv / X[0];
    const int x = (int)(iv % X[0]), y = (int)(l % X[1]); l /= X[1];
    const int z = (int)(l % X[2]), t = (int)(l / X[2]);
    const int parity = (x + y + z + t) & 1;
    memcpy(lex + iv * n, eo + (parity * Vh + iv / 2) * n, n * sizeof(double));
  }
}
void qo_lex_to_eo(double *eo, const double *lex, const int X[4], int n) {
  const long V = (long)X[0] * X[1] * X[2] * X[3], Vh = V / 2;
  for (long iv = 0; iv < V; iv++) {
    long l = iv / X[0];
    const int x = (int)(iv % X[0]), y = (int)(l % X[1]); l /= X[1];
    const int z = (int)(l % X[2]), t = (int)(l / X[2]);
    const int parity = (x + y + z + t) & 1;
    memcpy(eo + (parity * Vh + iv / 2) * n, lex + iv * n, n * sizeof(double));
  }
}

/* ---- APE smearing of the spatial links and the plaquette ----
 * Restated from lib/gauge_ape.cu:44-156 (computeStaple: the two staples of every spatial mu != dir around the link
 * (x, dir); computeAPEStep: TestU = (1 - alpha) + alpha/4 * staple * U^dag, polar projection, U' = TestU * U; only dir < 3
 * is smeared and only spatial staples enter), include/su3_project.cuh:23-124 (checkUnitary / polarSu3: Newton iteration
 * X <- (X + X^-dag)/2 until |X_ij - conj(X^-1_ji)| <= tol elementwise, then the determinant phase is divided out),
 * lib/interface_quda.cpp:5565-5640 (performAPEnStep: nSteps steps from a copy of the resident field) and
 * lib/gauge_plaq.cu:38-100, :129-152 (plaquette: spatial and temporal averages of Re tr P / 3, total = their mean).
 * Links in the host QDP order of loadGaugeQuda (gauge[mu][(parity*Vh + i)*18 + ...], tests/test_util.cpp:851); the
 * arithmetic runs on a lexicographic copy.  The Newton loop is capped at 100 sweeps (the reference loops until the test
 * passes).  PARITY UNPINNED (CUDA kernels, no vectors in the reference's tests): checked in tests/test_oracle_qkxtm.py
 * against an array formulation with a closed-form polar projection, and through unitarity / gauge covariance. */
#include <math.h>

typedef struct { double re[9], im[9]; } qm3;
static void m3_load(qm3 *m, const double *p) { for (int k = 0; k < 9; k++) { m->re[k] = p[2 * k]; m->im[k] = p[2 * k + 1]; } }
static void m3_store(double *p, const qm3 *m) { for (int k = 0; k < 9; k++) { p[2 * k] = m->re[k]; p[2 * k + 1] = m->im[k]; } }
static void m3_mul(qm3 *o, const qm3 *a, const qm3 *b) {
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      double re = 0, im = 0;
      for (int k = 0; k < 3; k++) {
        re += a->re[3 * i + k] * b->re[3 * k + j] - a->im[3 * i + k] * b->im[3 * k + j];
        im += a->re[3 * i + k] * b->im[3 * k + j] + a->im[3 * i + k] * b->re[3 * k + j];
      }
      o->re[3 * i + j] = re; o->im[3 * i + j] = im;
    }
}
static void m3_dag(qm3 *o, const qm3 *a) {
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { o->re[3 * i + j] = a->re[3 * j + i]; o->im[3 * i + j] = -a->im[3 * j + i]; }
}
static void c_mul(double *re, double *im, double ar, double ai, double br, double bi) { *re = ar * br - ai * bi; *im = ar * bi + ai * br; }
static void m3_det(double *dr, double *di, const qm3 *a) {
  double r = 0, i = 0;
  for (int c = 0; c < 3; c++) {
    const int c1 = (c + 1) % 3, c2 = (c + 2) % 3;
    double pr, pi, qr, qi, mr, mi, tr, ti;
    c_mul(&pr, &pi, a->re[3 + c1], a->im[3 + c1], a->re[6 + c2], a->im[6 + c2]);
    c_mul(&qr, &qi, a->re[3 + c2], a->im[3 + c2], a->re[6 + c1], a->im[6 + c1]);
    mr = pr - qr; mi = pi - qi;
    c_mul(&tr, &ti, a->re[c], a->im[c], mr, mi);
    r += tr; i += ti;
  }
  *dr = r; *di = i;
}
static void m3_inv(qm3 *o, const qm3 *a) {   /* adjugate / determinant */
  double dr, di;
  m3_det(&dr, &di, a);
  const double n = dr * dr + di * di, ir = dr / n, ii = -di / n;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      const int r1 = (j + 1) % 3, r2 = (j + 2) % 3, c1 = (i + 1) % 3, c2 = (i + 2) % 3;   /* cofactor (j, i) */
      double pr, pi, qr, qi, tr, ti;
      c_mul(&pr, &pi, a->re[3 * r1 + c1], a->im[3 * r1 + c1], a->re[3 * r2 + c2], a->im[3 * r2 + c2]);
      c_mul(&qr, &qi, a->re[3 * r1 + c2], a->im[3 * r1 + c2], a->re[3 * r2 + c1], a->im[3 * r2 + c1]);
      c_mul(&tr, &ti, pr - qr, pi - qi, ir, ii);
      o->re[3 * i + j] = tr; o->im[3 * i + j] = ti;
    }
}
static void polar_su3(qm3 *m, double tol) {
  qm3 out = *m, inv, invd;
  m3_inv(&inv, &out);
  for (int sweep = 0; sweep < 100; sweep++) {
    m3_dag(&invd, &inv);
    for (int k = 0; k < 9; k++) { out.re[k] = 0.5 * (out.re[k] + invd.re[k]); out.im[k] = 0.5 * (out.im[k] + invd.im[k]); }
    m3_inv(&inv, &out);
    int bad = 0;
    for (int i = 0; i < 3 && !bad; i++)
      for (int j = 0; j < 3; j++)
        if (fabs(out.re[3 * i + j] - inv.re[3 * j + i]) > tol || fabs(out.im[3 * i + j] + inv.im[3 * j + i]) > tol) { bad = 1; break; }
    if (!bad) break;
  }
  double dr, di;
  m3_det(&dr, &di, &out);
  const double mod = pow(dr * dr + di * di, 1.0 / 6.0), angle = atan2(di, dr) / -3.0;
  const double cr = cos(angle) / mod, ci = sin(angle) / mod;
  for (int k = 0; k < 9; k++) { m->re[k] = out.re[k] * cr - out.im[k] * ci; m->im[k] = out.re[k] * ci + out.im[k] * cr; }
}

static long lex_shift(long iv, const int X[4], int mu, int s) {
  long stride = 1;
  for (int d = 0; d < mu; d++) stride *= X[d];
  const int c = (int)((iv / stride) % X[mu]);
  return iv + ((c + s + X[mu]) % X[mu] - c) * stride;
}

static void ape_step_lex(double *const out[4], const double *const in[4], const int X[4], double alpha) {
  const long V = (long)X[0] * X[1] * X[2] * X[3];
#pragma omp parallel for
  for (long iv = 0; iv < V; iv++) {
    for (int dir = 0; dir < 3; dir++) {
      qm3 S, U1, U2, U3, t, t2, d;
      memset(&S, 0, sizeof(S));
      const int nu = dir;
      for (int mu = 0; mu < 3; mu++) {
        if (mu == dir) continue;
        /* upper: U_mu(x) U_nu(x+mu) U_mu(x+nu)^dag */
        m3_load(&U1, in[mu] + iv * 18);
        m3_load(&U2, in[nu] + lex_shift(iv, X, mu, +1) * 18);
        m3_load(&U3, in[mu] + lex_shift(iv, X, nu, +1) * 18);
        m3_mul(&t, &U1, &U2); m3_dag(&d, &U3); m3_mul(&t2, &t, &d);
        for (int k = 0; k < 9; k++) { S.re[k] += t2.re[k]; S.im[k] += t2.im[k]; }
        /* lower: U_mu(x-mu)^dag U_nu(x-mu) U_mu(x-mu+nu) */
        const long xm = lex_shift(iv, X, mu, -1);
        m3_load(&U1, in[mu] + xm * 18);
        m3_load(&U2, in[nu] + xm * 18);
        m3_load(&U3, in[mu] + lex_shift(xm, X, nu, +1) * 18);
        m3_dag(&d, &U1); m3_mul(&t, &d, &U2); m3_mul(&t2, &t, &U3);
        for (int k = 0; k < 9; k++) { S.re[k] += t2.re[k]; S.im[k] += t2.im[k]; }
      }
      qm3 U, Ud, T;
      m3_load(&U, in[dir] + iv * 18);
      const double f = alpha / 4.0;
      for (int k = 0; k < 9; k++) { S.re[k] *= f; S.im[k] *= f; }
      m3_dag(&Ud, &U);
      m3_mul(&T, &S, &Ud);
      for (int k = 0; k < 3; k++) T.re[4 * k] += 1.0 - alpha;
      polar_su3(&T, 1e-15);
      m3_mul(&t, &T, &U);
      m3_store(out[dir] + iv * 18, &t);
    }
    memcpy(out[3] + iv * 18, in[3] + iv * 18, 18 * sizeof(double));
  }
}

/* gauge_out = APE^nsteps(gauge_in), both in QDP even-odd order */
void qo_ape_smear(double *const gauge_out[4], const double *const gauge_in[4], const int X[4], double alpha, int nsteps) {
  const long V = (long)X[0] * X[1] * X[2] * X[3];
  double *a[4], *b[4];
  for (int d = 0; d < 4; d++) {
    a[d] = (double *)malloc(V * 18 * sizeof(double));
    b[d] = (double *)malloc(V * 18 * sizeof(double));
    qo_eo_to_lex(a[d], gauge_in[d], X, 18);
  }
  for (int s = 0; s < nsteps; s++) {
    ape_step_lex(b, (const double *const *)a, X, alpha);
    for (int d = 0; d < 4; d++) { double *t = a[d]; a[d] = b[d]; b[d] = t; }
  }
  for (int d = 0; d < 4; d++) { qo_lex_to_eo(gauge_out[d], a[d], X, 18); free(a[d]); free(b[d]); }
}

/* plaq[0] = mean of spatial and temporal, plaq[1] = spatial, plaq[2] = temporal (lib/gauge_plaq.cu:149-153) */
void qo_plaquette(double plaq[3], const double *const gauge[4], const int X[4]) {
  const long V = (long)X[0] * X[1] * X[2] * X[3];
  double *a[4];
  for (int d = 0; d < 4; d++) { a[d] = (double *)malloc(V * 18 * sizeof(double)); qo_eo_to_lex(a[d], gauge[d], X, 18); }
  double sp = 0, tm = 0;
#pragma omp parallel for reduction(+ : sp, tm)
  for (long iv = 0; iv < V; iv++)
    for (int mu = 0; mu < 3; mu++)
      for (int nu = mu + 1; nu < 4; nu++) {
        qm3 U1, U2, U3, U4, t, t2, d;
        m3_load(&U1, a[mu] + iv * 18);
        m3_load(&U2, a[nu] + lex_shift(iv, X, mu, +1) * 18);
        m3_load(&U3, a[mu] + lex_shift(iv, X, nu, +1) * 18);
        m3_load(&U4, a[nu] + iv * 18);
        m3_mul(&t, &U1, &U2); m3_dag(&d, &U3); m3_mul(&t2, &t, &d); m3_dag(&d, &U4); m3_mul(&t, &t2, &d);
        const double tr = t.re[0] + t.re[4] + t.re[8];
        if (nu < 3) sp += tr; else tm += tr;
      }
  for (int d = 0; d < 4; d++) free(a[d]);
  plaq[1] = sp / (9.0 * V); plaq[2] = tm / (9.0 * V); plaq[0] = 0.5 * (plaq[1] + plaq[2]);
}
