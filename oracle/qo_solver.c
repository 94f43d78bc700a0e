/*
 * oracle/qo_solver.c — TEST INFRASTRUCTURE / CPU BASELINE, NOT PRODUCT CODE.
 *
 * CPU restatement of the reference's restarted GCR in its plainest configuration (lib/inv_gcr_quda.cpp:235-516 with no
 * preconditioner, K = 1, one precision, no reliable updates; orthoDir pipeline 0 :92-97, backSubs :133-141, updateSolution
 * :143-157) on the oracle's host operator qo_tm_mat_d (= the reference's tm_mat, tests/wilson_dslash_reference.cpp:310-330), with
 * BLAS-1 loops in the style of the reference's CPU twins (lib/blas_cpu.cpp:10-358: flat loops over the reals of site-major
 * fields), parallelised with an outer `omp parallel for`.  It is the "MG-GCR seconds next to the host path" half of the CPU
 * baseline: bench.py times it on the same problem the GPU's plain GCR / MG-GCR solve, and
 * tests/test_oracle_solver.py checks its solution against the pinned operator.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>

#include "qo_fields.h"

typedef struct { double re, im; } qo_cplx;

static double nrm2(const double *x, long n) {
  double s = 0;
#pragma omp parallel for reduction(+ : s)
  for (long i = 0; i < n; i++) s += x[i] * x[i];
  return s;
}
static qo_cplx cdot(const double *x, const double *y, long n) {   /* sum conj(x) y */
  double re = 0, im = 0;
#pragma omp parallel for reduction(+ : re, im)
  for (long i = 0; i < n; i += 2) {
    re += x[i] * y[i] + x[i + 1] * y[i + 1];
    im += x[i] * y[i + 1] - x[i + 1] * y[i];
  }
  qo_cplx r = {re, im};
  return r;
}
static void caxpy(qo_cplx a, const double *x, double *y, long n) {   /* y += a x */
#pragma omp parallel for
  for (long i = 0; i < n; i += 2) {
    const double xr = x[i], xi = x[i + 1];
    y[i] += a.re * xr - a.im * xi;
    y[i + 1] += a.re * xi + a.im * xr;
  }
}
static void ax(double a, double *x, long n) {
#pragma omp parallel for
  for (long i = 0; i < n; i++) x[i] *= a;
}
static double xmyNorm(const double *x, double *y, long n) {   /* y = x - y ; |y|^2 */
  double s = 0;
#pragma omp parallel for reduction(+ : s)
  for (long i = 0; i < n; i++) { y[i] = x[i] - y[i]; s += y[i] * y[i]; }
  return s;
}
static double now(void) {
  struct timeval t;
  gettimeofday(&t, NULL);
  return t.tv_sec + 1e-6 * t.tv_usec;
}

/* x = M^-1 b (x = 0 on entry is NOT assumed: the residual is computed), relative tolerance on |r| / |b|.
 * Returns the number of operator applications inside the Krylov loop (= iterations); *secs the wall time of the solve,
 * *true_res = |b - M x| / |b| recomputed at the end. */
int qo_gcr_tm_d(double *x, double *const gauge[4], const double *b, const int X[4], double kappa, double mu, int flavor, double tol, int nkrylov,
                int maxiter, double *secs, double *true_res) {
  const long V = (long)X[0] * X[1] * X[2] * X[3], n = V * 24;
  double **p = (double **)malloc(nkrylov * sizeof(double *)), **Ap = (double **)malloc(nkrylov * sizeof(double *));
  for (int k = 0; k < nkrylov; k++) { p[k] = (double *)malloc(n * sizeof(double)); Ap[k] = (double *)malloc(n * sizeof(double)); }
  double *r = (double *)malloc(n * sizeof(double));
  qo_cplx *alpha = (qo_cplx *)calloc(nkrylov, sizeof(qo_cplx)), *delta = (qo_cplx *)calloc(nkrylov, sizeof(qo_cplx));
  qo_cplx *beta = (qo_cplx *)calloc((size_t)nkrylov * nkrylov, sizeof(qo_cplx));
  double *gamma = (double *)calloc(nkrylov, sizeof(double));
  const double t0 = now();
  const double b2 = nrm2(b, n), stop = tol * tol * b2;
  qo_tm_mat_d(r, gauge, x, kappa, mu, flavor, 0, X);
  double r2 = xmyNorm(b, r, n);
  int total = 0, k = 0;
  while (r2 > stop && total < maxiter) {
    memcpy(p[k], r, n * sizeof(double));                           /* K = 1: the search direction is the residual */
    qo_tm_mat_d(Ap[k], gauge, p[k], kappa, mu, flavor, 0, X);
    for (int i = 0; i < k; i++) {                                  /* orthoDir, pipeline 0 */
      const qo_cplx bik = cdot(Ap[i], Ap[k], n);
      beta[i * nkrylov + k] = bik;
      const qo_cplx m = {-bik.re, -bik.im};
      caxpy(m, Ap[i], Ap[k], n);
    }
    gamma[k] = sqrt(nrm2(Ap[k], n));
    ax(1.0 / gamma[k], Ap[k], n);
    alpha[k] = cdot(Ap[k], r, n);
    const qo_cplx ma = {-alpha[k].re, -alpha[k].im};
    caxpy(ma, Ap[k], r, n);                                        /* r -= alpha Ap */
    r2 = nrm2(r, n);
    k++; total++;
    if (k == nkrylov || !(r2 > stop) || total == maxiter) {        /* restart or done: x += sum_k delta_k p_k (backSubs) */
      for (int i = k - 1; i >= 0; i--) {
        qo_cplx d = alpha[i];
        for (int j = i + 1; j < k; j++) {
          const qo_cplx bb = beta[i * nkrylov + j];
          d.re -= bb.re * delta[j].re - bb.im * delta[j].im;
          d.im -= bb.re * delta[j].im + bb.im * delta[j].re;
        }
        delta[i].re = d.re / gamma[i]; delta[i].im = d.im / gamma[i];
      }
      for (int i = 0; i < k; i++) caxpy(delta[i], p[i], x, n);
      qo_tm_mat_d(r, gauge, x, kappa, mu, flavor, 0, X);           /* true residual at every restart */
      r2 = xmyNorm(b, r, n);
      k = 0;
    }
  }
  *secs = now() - t0;
  *true_res = sqrt(r2 / b2);
  for (int i = 0; i < nkrylov; i++) { free(p[i]); free(Ap[i]); }
  free(p); free(Ap); free(r); free(alpha); free(delta); free(beta); free(gamma);
  return total;
}
