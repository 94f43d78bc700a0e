/*
 * oracle/ref_driver.cpp — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A harness (this repo's own code) that calls the REFERENCE's host operators, compiled by
 * oracle/Makefile from the sources where they lie under /root/reference/tests/
 * (wilson_dslash_reference.cpp, clover_reference.cpp, blas_reference.cpp, test_util.cpp, misc.cpp)
 * plus lib/{comm_single,comm_common,util_quda,malloc}.cpp, into oracle/_ref/ref_driver.
 * It writes inputs and outputs as raw little-endian float64 files + a manifest; oracle/make_golden.py
 * packs them into tests/golden/*.npz.  Nothing from the reference is copied into this repo.
 *
 * usage: ref_driver golden <outdir> X Y Z T      (full element-wise vectors, small lattices)
 *        ref_driver checksum X Y Z T niter       (||out||^2 scalars + single-thread timing)
 */
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <sys/time.h>
#include <vector>

#include <quda.h>
#include <test_util.h>
#include <wilson_dslash_reference.h>
#include <blas_reference.h>

#include "qo_fields.h"  // only qo_clover_twisted_inverse_d: builds the (A^2+mu^2)^-1 INPUT field

extern int V, Vh;  // reference globals, tests/test_util.cpp:29-30

static std::string g_dir;
static FILE *g_manifest = nullptr;

static void dump(const std::string &name, const double *p, size_t n) {
  std::string path = g_dir + "/" + name + ".f64";
  FILE *f = fopen(path.c_str(), "wb");
  if (!f || fwrite(p, sizeof(double), n, f) != n) { fprintf(stderr, "write failed: %s\n", path.c_str()); exit(1); }
  fclose(f);
  fprintf(g_manifest, "%s %zu\n", name.c_str(), n);
}

static double now() {
  timeval t;
  gettimeofday(&t, nullptr);
  return t.tv_sec + 1e-6 * t.tv_usec;
}

static const char *mpc_name[4] = {"ee", "oo", "eeasym", "ooasym"};
static const QudaMatPCType mpc[4] = {QUDA_MATPC_EVEN_EVEN, QUDA_MATPC_ODD_ODD, QUDA_MATPC_EVEN_EVEN_ASYMMETRIC,
                                     QUDA_MATPC_ODD_ODD_ASYMMETRIC};

int main(int argc, char **argv) {
  if (argc < 2) return 1;
  const bool golden = !strcmp(argv[1], "golden");
  int a0 = golden ? 3 : 2;
  if (argc < a0 + 4) return 1;
  if (golden) g_dir = argv[2];
  int X[4];
  for (int d = 0; d < 4; d++) X[d] = atoi(argv[a0 + d]);
  int niter = (!golden && argc > a0 + 4) ? atoi(argv[a0 + 4]) : 0;

  QudaGaugeParam gp;
  memset(&gp, 0, sizeof(gp));
  for (int d = 0; d < 4; d++) gp.X[d] = X[d];
  gp.anisotropy = 1.0;
  gp.type = QUDA_WILSON_LINKS;
  gp.gauge_order = QUDA_QDP_GAUGE_ORDER;
  gp.t_boundary = QUDA_ANTI_PERIODIC_T;
  gp.cpu_prec = QUDA_DOUBLE_PRECISION;
  gp.gauge_fix = QUDA_GAUGE_FIXED_NO;

  setDims(gp.X);
  setSpinorSiteSize(24);
  const QudaPrecision prec = QUDA_DOUBLE_PRECISION;
  const size_t nsp = (size_t)V * 24, nh = (size_t)Vh * 24;

  srand(137);  // rank-0 seed of the reference harness: tests/test_util.cpp:81-92
  double *gauge[4];
  for (int d = 0; d < 4; d++) gauge[d] = (double *)malloc((size_t)V * 18 * sizeof(double));
  construct_gauge_field((void **)gauge, 1, prec, &gp);
  std::vector<double> spinor(nsp), in(nsp), out(nsp);
  for (size_t i = 0; i < nsp; i++) spinor[i] = rand() / (double)RAND_MAX;
  std::vector<double> clover((size_t)V * 72), cinv((size_t)V * 72);
  construct_clover_field(clover.data(), 0.1, 1.0, prec);

  const double kappa = golden ? 0.12 : 0.1, mu = golden ? 0.3 : 0.01;
  qo_clover_twisted_inverse_d(cinv.data(), clover.data(), V, 4 * kappa * kappa * mu * mu);

  if (!golden) {
    // checksum + timing mode (BASELINE.md section 3 protocol: resident fields, 1 thread)
    printf("{\"X\": [%d,%d,%d,%d], \"kappa\": %.17g, \"mu\": %.17g", X[0], X[1], X[2], X[3], kappa, mu);
    in = spinor;
    wil_dslash(out.data(), (void **)gauge, in.data(), 0, 0, prec, gp);
    printf(", \"wil_dslash_p0_d0\": %.17g", norm_2(out.data(), nh, prec));
    in = spinor;
    tm_dslash(out.data(), (void **)gauge, in.data(), kappa, mu, QUDA_TWIST_PLUS, 0, QUDA_MATPC_EVEN_EVEN, 0, prec, gp);
    printf(", \"tm_dslash_fp_ee_d0_p0\": %.17g", norm_2(out.data(), nh, prec));
    in = spinor;
    tm_matpc(out.data(), (void **)gauge, in.data(), kappa, mu, QUDA_TWIST_PLUS, QUDA_MATPC_EVEN_EVEN, 0, prec, gp);
    printf(", \"tm_matpc_fp_ee_d0\": %.17g", norm_2(out.data(), nh, prec));
    in = spinor;
    tmc_dslash(out.data(), (void **)gauge, in.data(), clover.data(), cinv.data(), kappa, mu, QUDA_TWIST_PLUS, 0,
               QUDA_MATPC_EVEN_EVEN, 0, prec, gp);
    printf(", \"tmc_dslash_fp_ee_d0_p0\": %.17g", norm_2(out.data(), nh, prec));
    if (niter > 0) {
      in = spinor;
      double t0 = now();
      for (int i = 0; i < niter; i++)
        tm_dslash(out.data(), (void **)gauge, in.data(), kappa, mu, QUDA_TWIST_PLUS, 0, QUDA_MATPC_EVEN_EVEN, 0, prec, gp);
      double dt = (now() - t0) / niter;
      printf(", \"tm_dslash_sec\": %.6g, \"tm_dslash_gflops\": %.6g", dt, 1368.0 * Vh / dt * 1e-9);
    }
    printf("}\n");
    return 0;
  }

  g_manifest = fopen((g_dir + "/manifest.txt").c_str(), "w");
  fprintf(g_manifest, "# X %d %d %d %d kappa %.17g mu %.17g\n", X[0], X[1], X[2], X[3], kappa, mu);
  for (int d = 0; d < 4; d++) dump("gauge" + std::to_string(d), gauge[d], (size_t)V * 18);
  dump("spinor", spinor.data(), nsp);
  dump("clover", clover.data(), (size_t)V * 72);
  dump("clover_inv", cinv.data(), (size_t)V * 72);

  char nm[128];
  // NB several reference operators overwrite-and-restore their input (not bit-exactly): fresh copy per case.
  for (int p = 0; p < 2; p++)
    for (int dg = 0; dg < 2; dg++) {
      in = spinor;
      wil_dslash(out.data(), (void **)gauge, in.data(), p, dg, prec, gp);
      snprintf(nm, sizeof nm, "wil_dslash_p%d_d%d", p, dg);
      dump(nm, out.data(), nh);
    }
  for (int p = 0; p < 2; p++) {
    in = spinor;
    apply_clover(out.data(), clover.data(), in.data(), p, prec);
    snprintf(nm, sizeof nm, "apply_clover_p%d", p);
    dump(nm, out.data(), nh);
  }
  for (int dg = 0; dg < 2; dg++) {
    in = spinor;
    wil_mat(out.data(), (void **)gauge, in.data(), kappa, dg, prec, gp);
    snprintf(nm, sizeof nm, "wil_mat_d%d", dg);
    dump(nm, out.data(), nsp);
    in = spinor;
    wil_matpc(out.data(), (void **)gauge, in.data(), kappa, QUDA_MATPC_EVEN_EVEN, dg, prec, gp);
    snprintf(nm, sizeof nm, "wil_matpc_ee_d%d", dg);
    dump(nm, out.data(), nh);
  }
  for (int m = 0; m < 4; m++)
    for (int dg = 0; dg < 2; dg++) {
      const int p0 = (m == 0 || m == 2) ? 0 : 1;
      for (int k = 0; k < 2; k++) {
        // natural parity with flavour +1, the other parity with flavour -1
        const int p = k == 0 ? p0 : 1 - p0;
        const QudaTwistFlavorType fl = k == 0 ? QUDA_TWIST_PLUS : QUDA_TWIST_MINUS;
        const char *fn = k == 0 ? "fp" : "fm";
        in = spinor;
        tm_dslash(out.data(), (void **)gauge, in.data(), kappa, mu, fl, p, mpc[m], dg, prec, gp);
        snprintf(nm, sizeof nm, "tm_dslash_%s_%s_d%d_p%d", fn, mpc_name[m], dg, p);
        dump(nm, out.data(), nh);
        in = spinor;
        tmc_dslash(out.data(), (void **)gauge, in.data(), clover.data(), cinv.data(), kappa, mu, fl, p, mpc[m], dg, prec, gp);
        snprintf(nm, sizeof nm, "tmc_dslash_%s_%s_d%d_p%d", fn, mpc_name[m], dg, p);
        dump(nm, out.data(), nh);
      }
      const QudaTwistFlavorType fl = (m + dg) % 2 ? QUDA_TWIST_MINUS : QUDA_TWIST_PLUS;
      const char *fn = (m + dg) % 2 ? "fm" : "fp";
      // the PC operators act on the parity-p0 half of the full source
      in = spinor;
      tm_matpc(out.data(), (void **)gauge, in.data() + p0 * nh, kappa, mu, fl, mpc[m], dg, prec, gp);
      snprintf(nm, sizeof nm, "tm_matpc_%s_%s_d%d", fn, mpc_name[m], dg);
      dump(nm, out.data(), nh);
      in = spinor;
      tmc_matpc(out.data(), (void **)gauge, in.data() + p0 * nh, clover.data(), cinv.data(), kappa, mu, fl, mpc[m], dg, prec, gp);
      snprintf(nm, sizeof nm, "tmc_matpc_%s_%s_d%d", fn, mpc_name[m], dg);
      dump(nm, out.data(), nh);
    }
  for (int dg = 0; dg < 2; dg++)
    for (int k = 0; k < 2; k++) {
      const QudaTwistFlavorType fl = k == 0 ? QUDA_TWIST_PLUS : QUDA_TWIST_MINUS;
      const char *fn = k == 0 ? "fp" : "fm";
      in = spinor;
      tm_mat(out.data(), (void **)gauge, in.data(), kappa, mu, fl, dg, prec, gp);
      snprintf(nm, sizeof nm, "tm_mat_%s_d%d", fn, dg);
      dump(nm, out.data(), nsp);
      in = spinor;
      tmc_mat(out.data(), (void **)gauge, clover.data(), in.data(), kappa, mu, fl, dg, prec, gp);
      snprintf(nm, sizeof nm, "tmc_mat_%s_d%d", fn, dg);
      dump(nm, out.data(), nsp);
    }
  fclose(g_manifest);
  return 0;
}
