/*
 * c_driver.c — a plain-C caller of libquda.so running the call sequence of INTEGRATION.md section 1 (the sequence of the
 * reference's drivers, qkxtm/CalcMG_2pt3pt_EvenOdd.cpp:649-747 and tests/multigrid_invert_test.cpp:477-513):
 *   initQuda -> loadGaugeQuda -> dslashQuda / MatQuda -> newMultigridQuda -> invertQuda (MG-preconditioned GCR) ->
 *   destroyMultigridQuda -> freeGaugeQuda -> endQuda
 * Built and run by tests/test_dropin_gpu.py with `gcc -I include c_driver.c -lquda`; only <quda.h> is included.
 * The residual |b - M x| / |b| is recomputed through MatQuda and must be below 1e-9 (exit status 0 / 1); the Python test
 * additionally checks the solution this program writes against the oracle's tm_mat.
 *
 *   c_driver L T out.bin        (lattice L^3 x T, twisted mass kappa = 0.12 mu = 0.02 on a deterministic smooth SU(3) field)
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <quda.h>

/* U = exp(i eps H) to second order, re-unitarised by Gram-Schmidt: a smooth, deterministic SU(3)-like field */
static void make_link(double *u, unsigned seed, double eps) {
  double h[3][3][2];
  unsigned s = seed * 2654435761u + 12345u;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
      for (int c = 0; c < 2; c++) { s = s * 1664525u + 1013904223u; h[i][j][c] = ((s >> 8) / 16777216.0 - 0.5); }
  double m[3][3][2];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {   /* 1 + i eps (h + h^dagger)/2 */
      const double hr = 0.5 * (h[i][j][0] + h[j][i][0]), hi = 0.5 * (h[i][j][1] - h[j][i][1]);
      m[i][j][0] = (i == j ? 1.0 : 0.0) - eps * hi;
      m[i][j][1] = eps * hr;
    }
  /* Gram-Schmidt rows 0, 1; row 2 = conj(row0 x row1) */
  double n = 0;
  for (int j = 0; j < 3; j++) n += m[0][j][0] * m[0][j][0] + m[0][j][1] * m[0][j][1];
  n = 1 / sqrt(n);
  for (int j = 0; j < 3; j++) { m[0][j][0] *= n; m[0][j][1] *= n; }
  double pr = 0, pi = 0;   /* <row0, row1> */
  for (int j = 0; j < 3; j++) { pr += m[0][j][0] * m[1][j][0] + m[0][j][1] * m[1][j][1]; pi += m[0][j][0] * m[1][j][1] - m[0][j][1] * m[1][j][0]; }
  for (int j = 0; j < 3; j++) { m[1][j][0] -= pr * m[0][j][0] - pi * m[0][j][1]; m[1][j][1] -= pr * m[0][j][1] + pi * m[0][j][0]; }
  n = 0;
  for (int j = 0; j < 3; j++) n += m[1][j][0] * m[1][j][0] + m[1][j][1] * m[1][j][1];
  n = 1 / sqrt(n);
  for (int j = 0; j < 3; j++) { m[1][j][0] *= n; m[1][j][1] *= n; }
  for (int j = 0; j < 3; j++) {
    const int a = (j + 1) % 3, b = (j + 2) % 3;
    const double cr = (m[0][a][0] * m[1][b][0] - m[0][a][1] * m[1][b][1]) - (m[0][b][0] * m[1][a][0] - m[0][b][1] * m[1][a][1]);
    const double ci = (m[0][a][0] * m[1][b][1] + m[0][a][1] * m[1][b][0]) - (m[0][b][0] * m[1][a][1] + m[0][b][1] * m[1][a][0]);
    m[2][j][0] = cr; m[2][j][1] = -ci;
  }
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) { u[(i * 3 + j) * 2] = m[i][j][0]; u[(i * 3 + j) * 2 + 1] = m[i][j][1]; }
}

int main(int argc, char **argv) {
  const int L = argc > 1 ? atoi(argv[1]) : 8, T = argc > 2 ? atoi(argv[2]) : 8;
  const char *outfile = argc > 3 ? argv[3] : NULL;
  const int X[4] = {L, L, L, T};
  const size_t V = (size_t)L * L * L * T;
  const double kappa = 0.12, mu = 0.02;

  double *gauge[4];
  for (int d = 0; d < 4; d++) {
    gauge[d] = (double *)malloc(V * 18 * sizeof(double));
    for (size_t i = 0; i < V; i++) make_link(gauge[d] + 18 * i, (unsigned)(d * V + i), 0.3);
  }
  double *b = (double *)calloc(V * 24, sizeof(double)), *x = (double *)calloc(V * 24, sizeof(double)), *r = (double *)calloc(V * 24, sizeof(double));
  for (size_t i = 0; i < V * 24; i++) b[i] = sin(0.37 * (double)i) + 0.5;

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

  QudaInvertParam ip = newQudaInvertParam();
  ip.dslash_type = QUDA_TWISTED_MASS_DSLASH; ip.kappa = kappa; ip.mu = mu; ip.epsilon = 0; ip.mass = 0.5 / kappa - 4.0;
  ip.twist_flavor = QUDA_TWIST_PLUS; ip.matpc_type = QUDA_MATPC_EVEN_EVEN; ip.dagger = QUDA_DAG_NO;
  ip.solution_type = QUDA_MAT_SOLUTION; ip.solve_type = QUDA_DIRECT_SOLVE; ip.mass_normalization = QUDA_KAPPA_NORMALIZATION;
  ip.cpu_prec = QUDA_DOUBLE_PRECISION; ip.cuda_prec = QUDA_DOUBLE_PRECISION; ip.cuda_prec_sloppy = QUDA_SINGLE_PRECISION;
  ip.cuda_prec_precondition = QUDA_SINGLE_PRECISION;
  ip.gamma_basis = QUDA_DEGRAND_ROSSI_GAMMA_BASIS; ip.dirac_order = QUDA_DIRAC_ORDER;
  ip.clover_cpu_prec = QUDA_DOUBLE_PRECISION; ip.clover_cuda_prec = QUDA_DOUBLE_PRECISION; ip.clover_cuda_prec_sloppy = QUDA_SINGLE_PRECISION;
  ip.clover_cuda_prec_precondition = QUDA_SINGLE_PRECISION; ip.clover_order = QUDA_PACKED_CLOVER_ORDER;
  ip.input_location = QUDA_CPU_FIELD_LOCATION; ip.output_location = QUDA_CPU_FIELD_LOCATION;
  ip.tune = QUDA_TUNE_NO; ip.sp_pad = 0; ip.cl_pad = 0; ip.verbosity = QUDA_SILENT;
  ip.inv_type = QUDA_GCR_INVERTER; ip.tol = 1e-10; ip.maxiter = 1000; ip.reliable_delta = 1e-4; ip.gcrNkrylov = 20;
  ip.use_init_guess = QUDA_USE_INIT_GUESS_NO; ip.preserve_source = QUDA_PRESERVE_SOURCE_YES; ip.residual_type = QUDA_L2_RELATIVE_RESIDUAL;

  /* one stencil application and one full-operator application through the C entry points */
  dslashQuda(r, b, &ip, QUDA_EVEN_PARITY);
  MatQuda(r, b, &ip);

  /* two-level hierarchy, filled as the reference harness does (tests/multigrid_invert_test.cpp:195-290) */
  QudaInvertParam mg_ip = ip;
  QudaMultigridParam mp = newQudaMultigridParam();
  mp.invert_param = &mg_ip;
  mp.n_level = 2;
  for (int l = 0; l < 2; l++) {
    for (int d = 0; d < 4; d++) mp.geo_block_size[l][d] = 4;
    for (int d = 4; d < QUDA_MAX_DIM; d++) mp.geo_block_size[l][d] = 1;
    mp.spin_block_size[l] = l == 0 ? 2 : 1;
    mp.n_vec[l] = 24; mp.nu_pre[l] = 2; mp.nu_post[l] = 2;
    mp.cycle_type[l] = QUDA_MG_CYCLE_RECURSIVE; mp.smoother[l] = QUDA_MR_INVERTER; mp.smoother_tol[l] = 0.25;
    mp.global_reduction[l] = QUDA_BOOLEAN_YES; mp.smoother_solve_type[l] = QUDA_DIRECT_PC_SOLVE;
    mp.coarse_grid_solution_type[l] = QUDA_MAT_SOLUTION; mp.omega[l] = 0.85; mp.location[l] = QUDA_CUDA_FIELD_LOCATION;
  }
  mp.setup_maxiter = 500; mp.setup_tol = 5e-6;
  mp.compute_null_vector = QUDA_COMPUTE_NULL_VECTOR_YES; mp.generate_all_levels = QUDA_BOOLEAN_YES; mp.run_verify = QUDA_BOOLEAN_NO;
  void *mg = newMultigridQuda(&mp);

  ip.inv_type_precondition = QUDA_MG_INVERTER; ip.preconditioner = mg;
  ip.tol_precondition = 1e-1; ip.maxiter_precondition = 1; ip.precondition_cycle = 1; ip.omega = 1.0;
  invertQuda(x, b, &ip);

  MatQuda(r, x, &ip);
  double n2 = 0, b2 = 0;
  for (size_t i = 0; i < V * 24; i++) { n2 += (b[i] - r[i]) * (b[i] - r[i]); b2 += b[i] * b[i]; }
  const double res = sqrt(n2 / b2);
  printf("c_driver: %dx%dx%dx%d MG-GCR iter %d secs %.4f true_res %.3e recomputed %.3e\n", L, L, L, T, ip.iter, ip.secs, ip.true_res, res);
  if (outfile) {
    FILE *f = fopen(outfile, "wb");
    if (!f) return 2;
    for (int d = 0; d < 4; d++) fwrite(gauge[d], sizeof(double), V * 18, f);
    fwrite(b, sizeof(double), V * 24, f);
    fwrite(x, sizeof(double), V * 24, f);
    fclose(f);
  }
  destroyMultigridQuda(mg);
  freeGaugeQuda();
  endQuda();
  return res < 1e-9 ? 0 : 1;
}
