// qkxtm_driver.cpp — a driver in the shape of the reference's qkxtm/CalcMG_2pt3pt_EvenOdd.cpp:649-747 and
// qkxtm/CalcMG_Loops_w_oneD_TSM_EvenOdd.cpp: it includes <qudaQKXTM_Kepler.h> from include/, builds one multigrid hierarchy per
// twist flavour into inv_param.preconditionerUP / preconditionerDN and calls the reference's entry points BY THEIR OWN NAMES
// (calcMG_threepTwop_EvenOdd, calcMG_loop_wOneD_TSM_EvenOdd, calcMG_loop_wOneD_TSM_wExact).  Where the reference contracts, the
// sink registered below appends every solution (and its source) to a file; tests/test_qkxtm_gpu.py re-checks each of them with
// the oracle's tm_mat.
//
//   qkxtm_driver gauge.bin Lx Ly Lz Lt out.bin     gauge.bin: 4 x V x 18 doubles, QDP even-odd order (as loadGaugeQuda takes them)
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <quda.h>
#include <quda_amd_ext.h>
#include <qudaQKXTM_Kepler.h>

static FILE *g_out = nullptr;
static void sink(void *, const char *kind, int index, int flavor, const double *h_source, const double *h_solution, size_t nreal) {
  char tag[16] = {0};
  strncpy(tag, kind, sizeof(tag) - 1);
  const int hdr[4] = {index, flavor, h_source ? 1 : 0, (int)nreal};
  fwrite(tag, 1, sizeof(tag), g_out);
  fwrite(hdr, sizeof(int), 4, g_out);
  if (h_source) fwrite(h_source, sizeof(double), nreal, g_out);
  fwrite(h_solution, sizeof(double), nreal, g_out);
}

int main(int argc, char **argv) {
  if (argc < 7) { fprintf(stderr, "usage: %s gauge.bin Lx Ly Lz Lt out.bin\n", argv[0]); return 2; }
  const int X[4] = {atoi(argv[2]), atoi(argv[3]), atoi(argv[4]), atoi(argv[5])};
  const size_t V = (size_t)X[0] * X[1] * X[2] * X[3];
  const double kappa = 0.124, mu = 0.005;
  std::vector<double> links[4];
  void *gauge[4];
  FILE *f = fopen(argv[1], "rb");
  if (!f) return 2;
  for (int d = 0; d < 4; d++) { links[d].resize(V * 18); if (fread(links[d].data(), sizeof(double), V * 18, f) != V * 18) return 2; gauge[d] = links[d].data(); }
  fclose(f);
  g_out = fopen(argv[6], "wb");
  if (!g_out) return 2;

  setVerbosityQuda(QUDA_SILENT, "", stdout);
  initQuda(0);
  QudaGaugeParam gp = newQudaGaugeParam();
  for (int d = 0; d < 4; d++) gp.X[d] = X[d];
  gp.anisotropy = 1.0; gp.type = QUDA_WILSON_LINKS; gp.gauge_order = QUDA_QDP_GAUGE_ORDER; gp.t_boundary = QUDA_PERIODIC_T;
  gp.cpu_prec = QUDA_DOUBLE_PRECISION; gp.cuda_prec = QUDA_DOUBLE_PRECISION; gp.reconstruct = QUDA_RECONSTRUCT_NO;
  gp.cuda_prec_sloppy = QUDA_SINGLE_PRECISION; gp.reconstruct_sloppy = QUDA_RECONSTRUCT_NO;
  gp.cuda_prec_precondition = QUDA_SINGLE_PRECISION; gp.reconstruct_precondition = QUDA_RECONSTRUCT_NO;
  gp.gauge_fix = QUDA_GAUGE_FIXED_NO; gp.ga_pad = 0;
  loadGaugeQuda((void *)gauge, &gp);

  // smeared links for the source smearing: produced in place (performAPEnStep) and handed over in the QKXTM lexicographic layout
  const int nsmearAPE = 2;
  const double alphaAPE = 0.5;
  performAPEnStep(nsmearAPE, alphaAPE);
  std::vector<double> ape[4];
  void *gauge_APE[4];
  for (int d = 0; d < 4; d++) { ape[d].resize(V * 18); gauge_APE[d] = ape[d].data(); }
  qudaAmdSaveSmearedGauge(gauge_APE, 1);

  QudaInvertParam ip = newQudaInvertParam();
  ip.dslash_type = QUDA_TWISTED_MASS_DSLASH; ip.kappa = kappa; ip.mu = mu; ip.epsilon = 0; ip.mass = 0.5 / kappa - 4.0;
  ip.twist_flavor = QUDA_TWIST_PLUS; ip.matpc_type = QUDA_MATPC_EVEN_EVEN; ip.dagger = QUDA_DAG_NO;
  ip.solution_type = QUDA_MAT_SOLUTION; ip.solve_type = QUDA_DIRECT_PC_SOLVE; ip.mass_normalization = QUDA_KAPPA_NORMALIZATION;
  ip.cpu_prec = QUDA_DOUBLE_PRECISION; ip.cuda_prec = QUDA_DOUBLE_PRECISION; ip.cuda_prec_sloppy = QUDA_SINGLE_PRECISION;
  ip.cuda_prec_precondition = QUDA_SINGLE_PRECISION;
  ip.gamma_basis = QUDA_UKQCD_GAMMA_BASIS; ip.dirac_order = QUDA_DIRAC_ORDER;
  ip.clover_cpu_prec = QUDA_DOUBLE_PRECISION; ip.clover_cuda_prec = QUDA_DOUBLE_PRECISION; ip.clover_cuda_prec_sloppy = QUDA_SINGLE_PRECISION;
  ip.clover_cuda_prec_precondition = QUDA_SINGLE_PRECISION; ip.clover_order = QUDA_PACKED_CLOVER_ORDER;
  ip.input_location = QUDA_CPU_FIELD_LOCATION; ip.output_location = QUDA_CPU_FIELD_LOCATION;
  ip.tune = QUDA_TUNE_NO; ip.sp_pad = 0; ip.cl_pad = 0; ip.verbosity = QUDA_SILENT;
  ip.inv_type = QUDA_GCR_INVERTER; ip.tol = 1e-10; ip.maxiter = 2000; ip.reliable_delta = 1e-4; ip.gcrNkrylov = 20;
  ip.use_init_guess = QUDA_USE_INIT_GUESS_NO; ip.preserve_source = QUDA_PRESERVE_SOURCE_YES; ip.residual_type = QUDA_L2_RELATIVE_RESIDUAL;

  // one hierarchy per twist flavour (reference CalcMG_2pt3pt_EvenOdd.cpp:700-730: mu > 0 -> preconditionerUP, mu < 0 -> DN)
  void *mg[2];
  QudaInvertParam mg_ip[2];
  QudaMultigridParam mp[2];
  for (int fl = 0; fl < 2; fl++) {
    mg_ip[fl] = ip;
    mg_ip[fl].solve_type = QUDA_DIRECT_SOLVE;
    mg_ip[fl].twist_flavor = fl == 0 ? QUDA_TWIST_PLUS : QUDA_TWIST_MINUS;
    mp[fl] = newQudaMultigridParam();
    mp[fl].invert_param = &mg_ip[fl];
    mp[fl].n_level = 2;
    for (int l = 0; l < 2; l++) {
      for (int d = 0; d < 4; d++) mp[fl].geo_block_size[l][d] = 4;
      for (int d = 4; d < QUDA_MAX_DIM; d++) mp[fl].geo_block_size[l][d] = 1;
      mp[fl].spin_block_size[l] = l == 0 ? 2 : 1;
      mp[fl].n_vec[l] = 8; mp[fl].nu_pre[l] = 2; mp[fl].nu_post[l] = 2;
      mp[fl].cycle_type[l] = QUDA_MG_CYCLE_RECURSIVE; mp[fl].smoother[l] = QUDA_MR_INVERTER; mp[fl].smoother_tol[l] = 0.25;
      mp[fl].global_reduction[l] = QUDA_BOOLEAN_YES; mp[fl].smoother_solve_type[l] = QUDA_DIRECT_PC_SOLVE;
      mp[fl].coarse_grid_solution_type[l] = QUDA_MATPC_SOLUTION; mp[fl].omega[l] = 0.85; mp[fl].location[l] = QUDA_CUDA_FIELD_LOCATION;
    }
    mp[fl].setup_maxiter = 100; mp[fl].setup_tol = 1e-4;
    mp[fl].compute_null_vector = QUDA_COMPUTE_NULL_VECTOR_YES; mp[fl].generate_all_levels = QUDA_BOOLEAN_YES; mp[fl].run_verify = QUDA_BOOLEAN_NO;
    mg[fl] = newMultigridQuda(&mp[fl]);
  }
  ip.inv_type_precondition = QUDA_MG_INVERTER;
  ip.preconditionerUP = mg[0]; ip.preconditionerDN = mg[1];
  ip.tol_precondition = 1e-1; ip.maxiter_precondition = 1; ip.precondition_cycle = 1; ip.omega = 1.0;

  qudaAmdSetSolutionSink(sink, nullptr);

  // ---- two- and three-point driver: two source positions ----
  static quda::qudaQKXTMinfo_Kepler info;   // ~20 KB, passed by value as in the reference
  memset(&info, 0, sizeof(info));
  info.nsmearAPE = nsmearAPE; info.alphaAPE = alphaAPE; info.nsmearGauss = 3; info.alphaGauss = 0.8;
  for (int d = 0; d < 4; d++) info.lL[d] = X[d];
  info.Nsources = 2;
  const int pos[2][4] = {{1, 2, 3, 5}, {0, 3, 1, 2}};
  for (int s = 0; s < 2; s++) for (int d = 0; d < 4; d++) info.sourcePosition[s][d] = pos[s][d] % X[d];
  info.Precision = QUDA_DOUBLE_PRECISION; info.isEven = true; info.kappa = kappa; info.mu = mu; info.inv_tol = ip.tol;
  info.source_type = quda::RANDOM;
  char twop[] = "unused_twop", threep[] = "unused_threep";
  calcMG_threepTwop_EvenOdd(gauge_APE, gauge, &gp, &ip, info, twop, threep, quda::PROTON);
  printf("calcMG_threepTwop_EvenOdd: %d outer iterations in 48 solves, %.3f s\n", ip.iter, ip.secs);

  // ---- loop driver with the truncated solver method: 3 low-precision solves + 2 (full, low) pairs; then plain, 2 sources ----
  static quda::qudaQKXTM_loopInfo loop;
  memset(&loop, 0, sizeof(loop));
  loop.Nstoch = 2; loop.seed = 4711; loop.useTSM = true; loop.TSM_NHP = 2; loop.TSM_NLP = 3; loop.TSM_tol = 1e-3; loop.TSM_maxiter = 0;
  loop.kappa = kappa; loop.mu = mu; loop.inv_tol = ip.tol;
  ip.twist_flavor = QUDA_TWIST_PLUS; ip.preconditioner = mg[0];
  calcMG_loop_wOneD_TSM_EvenOdd(gauge, &ip, &gp, loop, info);
  printf("calcMG_loop_wOneD_TSM_EvenOdd: %d outer iterations, %.3f s\n", ip.iter, ip.secs);
  static quda::qudaQKXTM_arpackInfo arpack;
  memset(&arpack, 0, sizeof(arpack));
  loop.useTSM = false;
  ip.twist_flavor = QUDA_TWIST_MINUS; ip.preconditioner = mg[1];
  QudaInvertParam evp = ip;
  calcMG_loop_wOneD_TSM_wExact(gauge, &evp, &ip, &gp, arpack, loop, info);
  printf("calcMG_loop_wOneD_TSM_wExact (nEv = 0): %d outer iterations, %.3f s\n", ip.iter, ip.secs);

  fclose(g_out);
  destroyMultigridQuda(mg[0]); destroyMultigridQuda(mg[1]);
  freeGaugeQuda();
  endQuda();
  return 0;
}
