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
