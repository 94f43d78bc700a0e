// cxx_consumer.cpp — a C++ caller of libquda.so that uses the library's C++ surface the way the reference's in-library callers
// do (the QKXTM calcMG_* functions, lib/interface_quda.cpp:6018-6560, and invertQuda itself, :2276-2540): setDiracParam +
// Dirac::create, ColorSpinorParam / cudaColorSpinorField / cpuColorSpinorField with operator= for the host<->device copies,
// Dirac::prepare / reconstruct, the DiracM functor, Solver::create(SolverParam, m, mSloppy, mPre) and blas::norm2 / xpay.
// Headers come from include/ under the reference's names (dirac_quda.h, invert_quda.h, color_spinor_field.h, blas_quda.h).
// Built and run by tests/test_dropin_gpu.py (g++, no hipcc needed); exit status 0 when the even-odd GCR solution it computes
// satisfies |b - M x| / |b| < 1e-9 with M applied through the same operator object, and equals invertQuda's solution.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include <quda.h>
#include <blas_quda.h>
#include <color_spinor_field.h>
#include <dirac_quda.h>
#include <invert_quda.h>
#include <interface_internal.h>

using namespace quda;

int main(int argc, char **argv) {
  const int L = argc > 1 ? atoi(argv[1]) : 8;
  const int X[4] = {L, L, L, L};
  const size_t V = (size_t)L * L * L * L;
  std::vector<double> links[4];
  void *gauge[4];
  for (int d = 0; d < 4; d++) {   // unit links with a small deterministic anti-hermitian kick (exact unitarity is not needed here)
    links[d].assign(V * 18, 0.0);
    for (size_t i = 0; i < V; i++)
      for (int c = 0; c < 3; c++) { links[d][i * 18 + c * 8] = cos(0.1 * ((i + d + c) % 7)); links[d][i * 18 + c * 8 + 1] = sin(0.1 * ((i + d + c) % 7)); }
    gauge[d] = links[d].data();
  }
  std::vector<double> hb(V * 24), hx(V * 24, 0.0), hx2(V * 24, 0.0);
  for (size_t i = 0; i < V * 24; i++) hb[i] = cos(0.11 * (double)i) + 0.25;

  setVerbosityQuda(QUDA_SILENT, "", stdout);
  initQuda(0);
  QudaGaugeParam gp = newQudaGaugeParam();
  for (int d = 0; d < 4; d++) gp.X[d] = X[d];
  gp.anisotropy = 1.0; gp.type = QUDA_WILSON_LINKS; gp.gauge_order = QUDA_QDP_GAUGE_ORDER; gp.t_boundary = QUDA_ANTI_PERIODIC_T;
  gp.cpu_prec = QUDA_DOUBLE_PRECISION; gp.cuda_prec = QUDA_DOUBLE_PRECISION; gp.reconstruct = QUDA_RECONSTRUCT_NO;
  gp.cuda_prec_sloppy = QUDA_SINGLE_PRECISION; gp.reconstruct_sloppy = QUDA_RECONSTRUCT_NO;
  gp.cuda_prec_precondition = QUDA_SINGLE_PRECISION; gp.reconstruct_precondition = QUDA_RECONSTRUCT_NO;
  gp.gauge_fix = QUDA_GAUGE_FIXED_NO; gp.ga_pad = 0;
  loadGaugeQuda((void *)gauge, &gp);

  QudaInvertParam ip = newQudaInvertParam();
  ip.dslash_type = QUDA_TWISTED_MASS_DSLASH; ip.kappa = 0.11; ip.mu = 0.05; ip.epsilon = 0; ip.mass = 0.5 / ip.kappa - 4.0;
  ip.twist_flavor = QUDA_TWIST_MINUS; ip.matpc_type = QUDA_MATPC_EVEN_EVEN; ip.dagger = QUDA_DAG_NO;
  ip.solution_type = QUDA_MAT_SOLUTION; ip.solve_type = QUDA_DIRECT_PC_SOLVE; ip.mass_normalization = QUDA_KAPPA_NORMALIZATION;
  ip.cpu_prec = QUDA_DOUBLE_PRECISION; ip.cuda_prec = QUDA_DOUBLE_PRECISION; ip.cuda_prec_sloppy = QUDA_SINGLE_PRECISION;
  ip.cuda_prec_precondition = QUDA_SINGLE_PRECISION;
  ip.gamma_basis = QUDA_UKQCD_GAMMA_BASIS; ip.dirac_order = QUDA_DIRAC_ORDER;
  ip.clover_cpu_prec = QUDA_DOUBLE_PRECISION; ip.clover_cuda_prec = QUDA_DOUBLE_PRECISION; ip.clover_cuda_prec_sloppy = QUDA_SINGLE_PRECISION;
  ip.clover_cuda_prec_precondition = QUDA_SINGLE_PRECISION; ip.clover_order = QUDA_PACKED_CLOVER_ORDER;
  ip.input_location = QUDA_CPU_FIELD_LOCATION; ip.output_location = QUDA_CPU_FIELD_LOCATION;
  ip.tune = QUDA_TUNE_NO; ip.sp_pad = 0; ip.cl_pad = 0; ip.verbosity = QUDA_SILENT;
  ip.inv_type = QUDA_GCR_INVERTER; ip.inv_type_precondition = QUDA_INVALID_INVERTER; ip.tol = 1e-10; ip.maxiter = 2000;
  ip.reliable_delta = 1e-4; ip.gcrNkrylov = 20;
  ip.use_init_guess = QUDA_USE_INIT_GUESS_NO; ip.preserve_source = QUDA_PRESERVE_SOURCE_YES; ip.residual_type = QUDA_L2_RELATIVE_RESIDUAL;

  // --- the in-library pattern: operators, fields, prepare, solve, reconstruct ---
  DiracParam dp, dpSloppy, dpPre;
  setDiracParam(dp, &ip, true);
  setDiracSloppyParam(dpSloppy, &ip, true);
  setDiracPreParam(dpPre, &ip, true);
  Dirac *d = Dirac::create(dp), *dSloppy = Dirac::create(dpSloppy), *dPre = Dirac::create(dpPre);

  ColorSpinorParam cpuParam(hb.data(), ip, X, false);
  cpuColorSpinorField h_b(cpuParam);
  cpuParam.v = hx.data();
  cpuColorSpinorField h_x(cpuParam);
  ColorSpinorParam cudaParam = deviceSpinorParam(ip.cuda_prec, QUDA_FULL_SITE_SUBSET, ip.twist_flavor);
  cudaParam.create = QUDA_ZERO_FIELD_CREATE;
  cudaColorSpinorField b(cudaParam), x(cudaParam);
  b = h_b;
  const double nb = blas::norm2(b);

  ColorSpinorField *in = nullptr, *out = nullptr;
  d->prepare(in, out, x, b, ip.solution_type);
  {
    DiracM m(*d), mSloppy(*dSloppy), mPre(*dPre);
    SolverParam sp(ip);
    Solver *solve = Solver::create(sp, m, mSloppy, mPre);
    (*solve)(*out, *in);
    sp.updateInvertParam(ip);
    delete solve;
  }
  d->reconstruct(x, b, ip.solution_type);
  h_x = x;
  const int iters = ip.iter;

  // residual with the FULL operator object built the same way
  DiracParam dfull;
  setDiracParam(dfull, &ip, false);
  Dirac *dF = Dirac::create(dfull);
  cudaColorSpinorField r(cudaParam);
  dF->M(r, x);
  const double r2 = blas::xmyNorm(b, r);   // r = b - r
  const double res = sqrt(r2 / nb);

  // the same solve through the C entry point
  invertQuda(hx2.data(), hb.data(), &ip);
  double diff = 0, ref = 0;
  for (size_t i = 0; i < V * 24; i++) { diff += (hx[i] - hx2[i]) * (hx[i] - hx2[i]); ref += hx2[i] * hx2[i]; }
  printf("cxx_consumer: %d^4 even-odd GCR through the C++ surface: %d iterations, |b - M x|/|b| = %.3e, vs invertQuda %.3e (%d iterations)\n", L, iters, res,
         sqrt(diff / ref), ip.iter);

  delete d; delete dSloppy; delete dPre; delete dF;
  freeGaugeQuda();
  endQuda();
  return (res < 1e-9 && diff <= 1e-18 * ref) ? 0 : 1;
}
