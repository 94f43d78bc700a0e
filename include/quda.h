/*
 * quda.h — C ABI of the MI355X-native twisted-mass Dslash / multigrid library (libquda.so).
 *
 * Drop-in boundary: every type, enumerator VALUE and entry point below is binary- and source-compatible
 * with the reference's public interface for this path:
 *     reference include/quda.h          (param structs :25-409, entry points :442-747, :1036-1038)
 *     reference include/enum_quda.h     (enumerator values)
 *     reference include/quda_constants.h
 * so that a driver written against the reference (tests/dslash_test.cpp, tests/multigrid_invert_test.cpp,
 * qkxtm/CalcMG_2pt3pt_EvenOdd.cpp) recompiles and links against this library unchanged.  Entry points of
 * the reference that belong to other fermion actions / HMC / gauge tools are deliberately absent
 * (SURVEY.md section 2, rows 17-19).  All functions are extern "C", take plain pointers and PODs.
 *
 * Error convention (reference include/util_quda.h:51-61): no return codes; a failure prints
 * "ERROR: ... (file:line in func())" and terminates the process with exit status 1.
 */
#ifndef _QUDA_H
#define _QUDA_H

#include <enum_quda.h>
#include <stdio.h> /* FILE */
#include <quda_constants.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- gauge-field description handed to loadGaugeQuda: reference include/quda.h:25-80 ---- */
typedef struct QudaGaugeParam_s {
  QudaFieldLocation location;
  int X[4];                      /* local extents, not checkerboarded */
  double anisotropy, tadpole_coeff, scale;
  QudaLinkType type;
  QudaGaugeFieldOrder gauge_order;
  QudaTboundary t_boundary;
  QudaPrecision cpu_prec, cuda_prec;
  QudaReconstructType reconstruct;
  QudaPrecision cuda_prec_sloppy;
  QudaReconstructType reconstruct_sloppy;
  QudaPrecision cuda_prec_precondition;
  QudaReconstructType reconstruct_precondition;
  QudaGaugeFixed gauge_fix;
  int ga_pad, site_ga_pad, staple_pad, llfat_ga_pad, mom_ga_pad;
  double gaugeGiB;               /* out: device storage used */
  int preserve_gauge;
  QudaStaggeredPhase staggered_phase_type;
  int staggered_phase_applied;
  double i_mu;
  int overlap, overwrite_mom;
  int use_resident_gauge, use_resident_mom, make_resident_gauge, make_resident_mom;
  int return_result_gauge, return_result_mom;
} QudaGaugeParam;

/* ---- operator / solver description: reference include/quda.h:86-298 ---- */
typedef struct QudaInvertParam_s {
  QudaFieldLocation input_location, output_location;
  QudaDslashType dslash_type;
  QudaInverterType inv_type;
  double mass, kappa, m5;
  int Ls;
  double b_5[QUDA_MAX_DWF_LS], c_5[QUDA_MAX_DWF_LS];
  double mu, epsilon;
  QudaTwistFlavorType twist_flavor;
  double tol, tol_restart, tol_hq;
  double true_res, true_res_hq;  /* out */
  int maxiter;
  double reliable_delta;
  int use_sloppy_partial_accumulator, max_res_increase, max_res_increase_total, heavy_quark_check, pipeline;
  int num_offset, num_src, overlap;
  double offset[QUDA_MAX_MULTI_SHIFT], tol_offset[QUDA_MAX_MULTI_SHIFT], tol_hq_offset[QUDA_MAX_MULTI_SHIFT];
  double true_res_offset[QUDA_MAX_MULTI_SHIFT], iter_res_offset[QUDA_MAX_MULTI_SHIFT], true_res_hq_offset[QUDA_MAX_MULTI_SHIFT];
  QudaSolutionType solution_type;
  QudaSolveType solve_type;
  QudaMatPCType matpc_type;
  QudaDagType dagger;
  QudaMassNormalization mass_normalization;
  QudaSolverNormalization solver_normalization;
  QudaPreserveSource preserve_source;
  QudaPrecision cpu_prec, cuda_prec, cuda_prec_sloppy, cuda_prec_precondition;
  QudaDiracFieldOrder dirac_order;
  QudaGammaBasis gamma_basis;
  QudaFieldLocation clover_location;
  QudaPrecision clover_cpu_prec, clover_cuda_prec, clover_cuda_prec_sloppy, clover_cuda_prec_precondition;
  QudaCloverFieldOrder clover_order;
  QudaUseInitGuess use_init_guess;
  double clover_coeff;
  int compute_clover_trlog;
  double trlogA[2];
  int compute_clover, compute_clover_inverse, return_clover, return_clover_inverse;
  QudaVerbosity verbosity;
  int sp_pad, cl_pad;
  int iter;                      /* out */
  double spinorGiB, cloverGiB, gflops, secs; /* out */
  QudaTune tune;
  int Nsteps, gcrNkrylov;
  QudaInverterType inv_type_precondition;
  void *preconditioner;          /* instance returned by newMultigridQuda */
  void *preconditionerUP, *preconditionerDN; /* QKXTM: per-flavour hierarchies, reference :226-228 */
  QudaDslashType dslash_type_precondition;
  QudaVerbosity verbosity_precondition;
  double tol_precondition;
  int maxiter_precondition;
  double omega;
  int precondition_cycle;
  QudaSchwarzType schwarz_type;
  QudaResidualType residual_type;
  QudaPrecision cuda_prec_ritz;
  int nev, max_search_dim, rhs_idx, deflation_grid, use_reduced_vector_set;
  double eigenval_tol;
  int use_cg_updates;
  double cg_iterref_tol;
  int eigcg_max_restarts, max_restart_num;
  double inc_tol;
  int make_resident_solution, use_resident_solution;
} QudaInvertParam;

/* reference include/quda.h:301-325 (kept for layout compatibility of drivers that embed it) */
typedef struct QudaEigParam_s {
  QudaInvertParam *invert_param;
  QudaSolutionType RitzMat_lanczos, RitzMat_Convcheck;
  QudaEigType eig_type;
  double *MatPoly_param;
  int NPoly;
  double Stp_residual;
  int nk, np, f_size;
  double eigen_shift;
} QudaEigParam;

/* ---- multigrid hierarchy description: reference include/quda.h:327-409 ---- */
typedef struct QudaMultigridParam_s {
  QudaInvertParam *invert_param;
  int n_level;
  int geo_block_size[QUDA_MAX_MG_LEVEL][QUDA_MAX_DIM];
  int spin_block_size[QUDA_MAX_MG_LEVEL];
  int n_vec[QUDA_MAX_MG_LEVEL];
  QudaInverterType smoother[QUDA_MAX_MG_LEVEL];
  QudaSolutionType coarse_grid_solution_type[QUDA_MAX_MG_LEVEL];
  QudaSolveType smoother_solve_type[QUDA_MAX_MG_LEVEL];
  QudaMultigridCycleType cycle_type[QUDA_MAX_MG_LEVEL];
  int nu_pre[QUDA_MAX_MG_LEVEL], nu_post[QUDA_MAX_MG_LEVEL];
  double smoother_tol[QUDA_MAX_MG_LEVEL];
  int setup_maxiter;             /* null-vector BiCGstab iterations (QKXTM addition, :366-369) */
  double setup_tol;
  double omega[QUDA_MAX_MG_LEVEL];
  QudaBoolean global_reduction[QUDA_MAX_MG_LEVEL];
  QudaFieldLocation location[QUDA_MAX_MG_LEVEL];
  QudaComputeNullVector compute_null_vector;
  QudaBoolean generate_all_levels, run_verify;
  char vec_infile[256], vec_outfile[256];
  double gflops, secs;           /* out */
  double delta_muPR, delta_kappaPR, delta_cswPR; /* QKXTM: setup-operator rescaling, :401-407 */
  double delta_muCG, delta_kappaCG, delta_cswCG;
} QudaMultigridParam;

/* ---- entry points ---- */
typedef int (*QudaCommsMap)(const int *coords, void *fdata);

void setVerbosityQuda(QudaVerbosity verbosity, const char prefix[], FILE *outfile); /* ref quda.h:442 */
void initCommsGridQuda(int nDim, const int *dims, QudaCommsMap func, void *fdata);   /* ref quda.h:483 */
void initQudaDevice(int device);                                                     /* ref quda.h:495 */
void initQudaMemory(void);                                                           /* ref quda.h:503 */
void initQuda(int device);                                                           /* ref quda.h:514 */
void endQuda(void);                                                                  /* ref quda.h:519 */

QudaGaugeParam newQudaGaugeParam(void);             /* ref quda.h:528 — fields preset to "invalid" sentinels */
QudaInvertParam newQudaInvertParam(void);           /* ref quda.h:537 */
QudaMultigridParam newQudaMultigridParam(void);     /* ref quda.h:546 */
void printQudaGaugeParam(QudaGaugeParam *param);    /* ref quda.h:561 */
void printQudaInvertParam(QudaInvertParam *param);  /* ref quda.h:567 */
void printQudaMultigridParam(QudaMultigridParam *param); /* ref quda.h:573 */

void loadGaugeQuda(void *h_gauge, QudaGaugeParam *param);  /* ref quda.h:586; lib/interface_quda.cpp:521 */
void freeGaugeQuda(void);                                  /* ref quda.h:591 */
void saveGaugeQuda(void *h_gauge, QudaGaugeParam *param);  /* ref quda.h:598; lib/interface_quda.cpp:694: resident links back in host QDP order */
void plaqQuda(double plaq[3]);                             /* ref quda.h:964; lib/interface_quda.cpp:5510: total, spatial, temporal plaquette of the resident links */
void performAPEnStep(unsigned int nSteps, double alpha);   /* ref quda.h:971; lib/interface_quda.cpp:5565: APE-smear the resident spatial links into the library's smeared field */
void loadCloverQuda(void *h_clover, void *h_clovinv, QudaInvertParam *inv_param); /* ref quda.h:607; interface_quda.cpp:730 */
void freeCloverQuda(void);                                 /* ref quda.h:613 */

void invertQuda(void *h_x, void *h_b, QudaInvertParam *param);       /* ref quda.h:636; interface_quda.cpp:2276 */
/* param->num_src sources through ONE lockstep solve (GCR, optionally MG-preconditioned: the cycle below the fine level runs on block fields
 * through the multi-right-hand-side MFMA coarse operator); ref quda.h:647, whose own implementation "cannot work" (interface_quda.cpp:2546-2549) */
void invertMultiSrcQuda(void **_hp_x, void **_hp_b, QudaInvertParam *param);
void *newMultigridQuda(QudaMultigridParam *param);                   /* ref quda.h:666; interface_quda.cpp:2257 */
void destroyMultigridQuda(void *mg_instance);                        /* ref quda.h:671 */

void dslashQuda(void *h_out, void *h_in, QudaInvertParam *inv_param, QudaParity parity); /* ref quda.h:692; interface_quda.cpp:1496 */
void cloverQuda(void *h_out, void *h_in, QudaInvertParam *inv_param, QudaParity *parity, int inverse); /* ref quda.h:728 */
void MatQuda(void *h_out, void *h_in, QudaInvertParam *inv_param);        /* ref quda.h:738; interface_quda.cpp:1716 */
void MatDagMatQuda(void *h_out, void *h_in, QudaInvertParam *inv_param);  /* ref quda.h:747 */

void openMagma(void);   /* ref quda.h:1036 — no-ops here: the dense coarse-clover inverse is done on device */
void closeMagma(void);  /* ref quda.h:1038 */

#ifdef __cplusplus
}
#endif
#endif /* _QUDA_H */
