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
#ifndef QUDA_AMD_QUDA_H
#define QUDA_AMD_QUDA_H

#include <limits.h>
#include <stdio.h>

/* ---- constants: reference include/quda_constants.h:1-44 ---- */
#define QUDA_VERSION_MAJOR 0
#define QUDA_VERSION_MINOR 9
#define QUDA_VERSION_SUBMINOR 0
#define QUDA_VERSION ((QUDA_VERSION_MAJOR << 16) | (QUDA_VERSION_MINOR << 8) | QUDA_VERSION_SUBMINOR)
#define QUDA_MAX_DIM 6
#define QUDA_MAX_GEOMETRY 8
#define QUDA_MAX_MULTI_SHIFT 32
#define QUDA_MAX_DWF_LS 128
#define QUDA_MAX_MG_LEVEL 4
#define QUDA_MAX_MULTI_REDUCE 16

#define QUDA_INVALID_ENUM INT_MIN

#ifdef __cplusplus
extern "C" {
#endif

/* ---- enumerations (values are ABI): reference include/enum_quda.h ---- */
typedef enum QudaLinkType_s {
  QUDA_SU3_LINKS, QUDA_GENERAL_LINKS, QUDA_THREE_LINKS, QUDA_MOMENTUM, QUDA_COARSE_LINKS,
  QUDA_WILSON_LINKS = QUDA_SU3_LINKS, QUDA_ASQTAD_FAT_LINKS = QUDA_GENERAL_LINKS,
  QUDA_ASQTAD_LONG_LINKS = QUDA_THREE_LINKS, QUDA_ASQTAD_MOM_LINKS = QUDA_MOMENTUM,
  QUDA_ASQTAD_GENERAL_LINKS = QUDA_GENERAL_LINKS, QUDA_INVALID_LINKS = QUDA_INVALID_ENUM
} QudaLinkType;

typedef enum QudaGaugeFieldOrder_s {
  QUDA_FLOAT_GAUGE_ORDER = 1, QUDA_FLOAT2_GAUGE_ORDER = 2, QUDA_FLOAT4_GAUGE_ORDER = 4,
  QUDA_QDP_GAUGE_ORDER, QUDA_QDPJIT_GAUGE_ORDER, QUDA_CPS_WILSON_GAUGE_ORDER, QUDA_MILC_GAUGE_ORDER,
  QUDA_BQCD_GAUGE_ORDER, QUDA_TIFR_GAUGE_ORDER, QUDA_INVALID_GAUGE_ORDER = QUDA_INVALID_ENUM
} QudaGaugeFieldOrder;

typedef enum QudaTboundary_s { QUDA_ANTI_PERIODIC_T = -1, QUDA_PERIODIC_T = 1, QUDA_INVALID_T_BOUNDARY = QUDA_INVALID_ENUM } QudaTboundary;

typedef enum QudaPrecision_s {
  QUDA_HALF_PRECISION = 2, QUDA_SINGLE_PRECISION = 4, QUDA_DOUBLE_PRECISION = 8, QUDA_INVALID_PRECISION = QUDA_INVALID_ENUM
} QudaPrecision;

typedef enum QudaReconstructType_s {
  QUDA_RECONSTRUCT_NO = 18, QUDA_RECONSTRUCT_12 = 12, QUDA_RECONSTRUCT_8 = 8, QUDA_RECONSTRUCT_9 = 9,
  QUDA_RECONSTRUCT_13 = 13, QUDA_RECONSTRUCT_10 = 10, QUDA_RECONSTRUCT_INVALID = QUDA_INVALID_ENUM
} QudaReconstructType;

typedef enum QudaGaugeFixed_s { QUDA_GAUGE_FIXED_NO, QUDA_GAUGE_FIXED_YES, QUDA_GAUGE_FIXED_INVALID = QUDA_INVALID_ENUM } QudaGaugeFixed;

typedef enum QudaDslashType_s {
  QUDA_WILSON_DSLASH, QUDA_CLOVER_WILSON_DSLASH, QUDA_DOMAIN_WALL_DSLASH, QUDA_DOMAIN_WALL_4D_DSLASH,
  QUDA_MOBIUS_DWF_DSLASH, QUDA_STAGGERED_DSLASH, QUDA_ASQTAD_DSLASH, QUDA_TWISTED_MASS_DSLASH,
  QUDA_TWISTED_CLOVER_DSLASH, QUDA_INVALID_DSLASH = QUDA_INVALID_ENUM
} QudaDslashType;

typedef enum QudaInverterType_s {
  QUDA_CG_INVERTER, QUDA_BICGSTAB_INVERTER, QUDA_GCR_INVERTER, QUDA_MR_INVERTER, QUDA_MPBICGSTAB_INVERTER,
  QUDA_SD_INVERTER, QUDA_XSD_INVERTER, QUDA_PCG_INVERTER, QUDA_MPCG_INVERTER, QUDA_EIGCG_INVERTER,
  QUDA_INC_EIGCG_INVERTER, QUDA_GMRESDR_INVERTER, QUDA_GMRESDR_PROJ_INVERTER, QUDA_GMRESDR_SH_INVERTER,
  QUDA_FGMRESDR_INVERTER, QUDA_MG_INVERTER, QUDA_INVALID_INVERTER = QUDA_INVALID_ENUM
} QudaInverterType;

typedef enum QudaEigType_s { QUDA_LANCZOS, QUDA_IMP_RST_LANCZOS, QUDA_INVALID_TYPE = QUDA_INVALID_ENUM } QudaEigType;

typedef enum QudaSolutionType_s {
  QUDA_MAT_SOLUTION, QUDA_MATDAG_MAT_SOLUTION, QUDA_MATPC_SOLUTION, QUDA_MATPC_DAG_SOLUTION,
  QUDA_MATPCDAG_MATPC_SOLUTION, QUDA_MATPCDAG_MATPC_SHIFT_SOLUTION, QUDA_INVALID_SOLUTION = QUDA_INVALID_ENUM
} QudaSolutionType;

typedef enum QudaSolveType_s {
  QUDA_DIRECT_SOLVE, QUDA_NORMOP_SOLVE, QUDA_DIRECT_PC_SOLVE, QUDA_NORMOP_PC_SOLVE, QUDA_NORMERR_SOLVE,
  QUDA_NORMERR_PC_SOLVE, QUDA_NORMEQ_SOLVE = QUDA_NORMOP_SOLVE, QUDA_NORMEQ_PC_SOLVE = QUDA_NORMOP_PC_SOLVE,
  QUDA_INVALID_SOLVE = QUDA_INVALID_ENUM
} QudaSolveType;

typedef enum QudaMultigridCycleType_s {
  QUDA_MG_CYCLE_VCYCLE, QUDA_MG_CYCLE_FCYCLE, QUDA_MG_CYCLE_WCYCLE, QUDA_MG_CYCLE_RECURSIVE,
  QUDA_MG_CYCLE_INVALID = QUDA_INVALID_ENUM
} QudaMultigridCycleType;

typedef enum QudaSchwarzType_s { QUDA_ADDITIVE_SCHWARZ, QUDA_MULTIPLICATIVE_SCHWARZ, QUDA_INVALID_SCHWARZ = QUDA_INVALID_ENUM } QudaSchwarzType;

typedef enum QudaResidualType_s {
  QUDA_L2_RELATIVE_RESIDUAL = 1, QUDA_L2_ABSOLUTE_RESIDUAL = 2, QUDA_HEAVY_QUARK_RESIDUAL = 4,
  QUDA_INVALID_RESIDUAL = QUDA_INVALID_ENUM
} QudaResidualType;

typedef enum QudaMatPCType_s {
  QUDA_MATPC_EVEN_EVEN, QUDA_MATPC_ODD_ODD, QUDA_MATPC_EVEN_EVEN_ASYMMETRIC, QUDA_MATPC_ODD_ODD_ASYMMETRIC,
  QUDA_MATPC_INVALID = QUDA_INVALID_ENUM
} QudaMatPCType;

typedef enum QudaDagType_s { QUDA_DAG_NO, QUDA_DAG_YES, QUDA_DAG_INVALID = QUDA_INVALID_ENUM } QudaDagType;

typedef enum QudaMassNormalization_s {
  QUDA_KAPPA_NORMALIZATION, QUDA_MASS_NORMALIZATION, QUDA_ASYMMETRIC_MASS_NORMALIZATION,
  QUDA_INVALID_NORMALIZATION = QUDA_INVALID_ENUM
} QudaMassNormalization;

typedef enum QudaSolverNormalization_s { QUDA_DEFAULT_NORMALIZATION, QUDA_SOURCE_NORMALIZATION } QudaSolverNormalization;
typedef enum QudaPreserveSource_s { QUDA_PRESERVE_SOURCE_NO, QUDA_PRESERVE_SOURCE_YES, QUDA_PRESERVE_SOURCE_INVALID = QUDA_INVALID_ENUM } QudaPreserveSource;

typedef enum QudaDiracFieldOrder_s {
  QUDA_INTERNAL_DIRAC_ORDER, QUDA_DIRAC_ORDER, QUDA_QDP_DIRAC_ORDER, QUDA_QDPJIT_DIRAC_ORDER,
  QUDA_CPS_WILSON_DIRAC_ORDER, QUDA_LEX_DIRAC_ORDER, QUDA_INVALID_DIRAC_ORDER = QUDA_INVALID_ENUM
} QudaDiracFieldOrder;

typedef enum QudaCloverFieldOrder_s {
  QUDA_FLOAT_CLOVER_ORDER = 1, QUDA_FLOAT2_CLOVER_ORDER = 2, QUDA_FLOAT4_CLOVER_ORDER = 4, QUDA_PACKED_CLOVER_ORDER,
  QUDA_QDPJIT_CLOVER_ORDER, QUDA_BQCD_CLOVER_ORDER, QUDA_INVALID_CLOVER_ORDER = QUDA_INVALID_ENUM
} QudaCloverFieldOrder;

typedef enum QudaVerbosity_s { QUDA_SILENT, QUDA_SUMMARIZE, QUDA_VERBOSE, QUDA_DEBUG_VERBOSE, QUDA_INVALID_VERBOSITY = QUDA_INVALID_ENUM } QudaVerbosity;
typedef enum QudaTune_s { QUDA_TUNE_NO, QUDA_TUNE_YES, QUDA_TUNE_INVALID = QUDA_INVALID_ENUM } QudaTune;
typedef enum QudaPreserveDirac_s { QUDA_PRESERVE_DIRAC_NO, QUDA_PRESERVE_DIRAC_YES, QUDA_PRESERVE_DIRAC_INVALID = QUDA_INVALID_ENUM } QudaPreserveDirac;
typedef enum QudaParity_s { QUDA_EVEN_PARITY = 0, QUDA_ODD_PARITY, QUDA_INVALID_PARITY = QUDA_INVALID_ENUM } QudaParity;

typedef enum QudaDiracType_s {
  QUDA_WILSON_DIRAC, QUDA_WILSONPC_DIRAC, QUDA_CLOVER_DIRAC, QUDA_CLOVERPC_DIRAC, QUDA_DOMAIN_WALL_DIRAC,
  QUDA_DOMAIN_WALLPC_DIRAC, QUDA_DOMAIN_WALL_4DPC_DIRAC, QUDA_MOBIUS_DOMAIN_WALL_DIRAC,
  QUDA_MOBIUS_DOMAIN_WALLPC_DIRAC, QUDA_STAGGERED_DIRAC, QUDA_STAGGEREDPC_DIRAC, QUDA_ASQTAD_DIRAC,
  QUDA_ASQTADPC_DIRAC, QUDA_TWISTED_MASS_DIRAC, QUDA_TWISTED_MASSPC_DIRAC, QUDA_TWISTED_CLOVER_DIRAC,
  QUDA_TWISTED_CLOVERPC_DIRAC, QUDA_COARSE_DIRAC, QUDA_COARSEPC_DIRAC, QUDA_INVALID_DIRAC = QUDA_INVALID_ENUM
} QudaDiracType;

typedef enum QudaFieldLocation_s { QUDA_CPU_FIELD_LOCATION = 1, QUDA_CUDA_FIELD_LOCATION = 2, QUDA_INVALID_FIELD_LOCATION = QUDA_INVALID_ENUM } QudaFieldLocation;
typedef enum QudaSiteSubset_s { QUDA_PARITY_SITE_SUBSET = 1, QUDA_FULL_SITE_SUBSET = 2, QUDA_INVALID_SITE_SUBSET = QUDA_INVALID_ENUM } QudaSiteSubset;
typedef enum QudaSiteOrder_s { QUDA_LEXICOGRAPHIC_SITE_ORDER, QUDA_EVEN_ODD_SITE_ORDER, QUDA_ODD_EVEN_SITE_ORDER, QUDA_INVALID_SITE_ORDER = QUDA_INVALID_ENUM } QudaSiteOrder;

typedef enum QudaFieldOrder_s {
  QUDA_FLOAT_FIELD_ORDER = 1, QUDA_FLOAT2_FIELD_ORDER = 2, QUDA_FLOAT4_FIELD_ORDER = 4,
  QUDA_SPACE_SPIN_COLOR_FIELD_ORDER, QUDA_SPACE_COLOR_SPIN_FIELD_ORDER, QUDA_QDPJIT_FIELD_ORDER,
  QUDA_QOP_DOMAIN_WALL_FIELD_ORDER, QUDA_INVALID_FIELD_ORDER = QUDA_INVALID_ENUM
} QudaFieldOrder;

typedef enum QudaFieldCreate_s {
  QUDA_NULL_FIELD_CREATE, QUDA_ZERO_FIELD_CREATE, QUDA_COPY_FIELD_CREATE, QUDA_REFERENCE_FIELD_CREATE,
  QUDA_INVALID_FIELD_CREATE = QUDA_INVALID_ENUM
} QudaFieldCreate;

typedef enum QudaGammaBasis_s { QUDA_DEGRAND_ROSSI_GAMMA_BASIS, QUDA_UKQCD_GAMMA_BASIS, QUDA_CHIRAL_GAMMA_BASIS, QUDA_INVALID_GAMMA_BASIS = QUDA_INVALID_ENUM } QudaGammaBasis;
typedef enum QudaSourceType_s { QUDA_POINT_SOURCE, QUDA_RANDOM_SOURCE, QUDA_CONSTANT_SOURCE, QUDA_SINUSOIDAL_SOURCE, QUDA_INVALID_SOURCE = QUDA_INVALID_ENUM } QudaSourceType;

typedef enum QudaTwistFlavorType_s {
  QUDA_TWIST_MINUS = -1, QUDA_TWIST_PLUS = +1, QUDA_TWIST_NONDEG_DOUBLET = +2, QUDA_TWIST_DEG_DOUBLET = -2,
  QUDA_TWIST_NO = 0, QUDA_TWIST_INVALID = QUDA_INVALID_ENUM
} QudaTwistFlavorType;

typedef enum QudaTwistDslashType_s {
  QUDA_DEG_TWIST_INV_DSLASH, QUDA_DEG_DSLASH_TWIST_INV, QUDA_DEG_DSLASH_TWIST_XPAY, QUDA_NONDEG_DSLASH,
  QUDA_DSLASH_INVALID = QUDA_INVALID_ENUM
} QudaTwistDslashType;

typedef enum QudaTwistCloverDslashType_s {
  QUDA_DEG_CLOVER_TWIST_INV_DSLASH, QUDA_DEG_DSLASH_CLOVER_TWIST_INV, QUDA_DEG_DSLASH_CLOVER_TWIST_XPAY,
  QUDA_TC_DSLASH_INVALID = QUDA_INVALID_ENUM
} QudaTwistCloverDslashType;

typedef enum QudaTwistGamma5Type_s { QUDA_TWIST_GAMMA5_DIRECT, QUDA_TWIST_GAMMA5_INVERSE, QUDA_TWIST_GAMMA5_INVALID = QUDA_INVALID_ENUM } QudaTwistGamma5Type;
typedef enum QudaUseInitGuess_s { QUDA_USE_INIT_GUESS_NO, QUDA_USE_INIT_GUESS_YES, QUDA_USE_INIT_GUESS_INVALID = QUDA_INVALID_ENUM } QudaUseInitGuess;
typedef enum QudaComputeNullVector_s { QUDA_COMPUTE_NULL_VECTOR_NO, QUDA_COMPUTE_NULL_VECTOR_YES, QUDA_COMPUTE_NULL_VECTOR_INVALID = QUDA_INVALID_ENUM } QudaComputeNullVector;
typedef enum QudaBoolean_s { QUDA_BOOLEAN_NO = 0, QUDA_BOOLEAN_YES = 1, QUDA_BOOLEAN_INVALID = QUDA_INVALID_ENUM } QudaBoolean;
typedef enum QudaDirection_s { QUDA_BACKWARDS = -1, QUDA_FORWARDS = +1, QUDA_BOTH_DIRS = 2 } QudaDirection;
typedef enum QudaFieldGeometry_s { QUDA_SCALAR_GEOMETRY = 1, QUDA_VECTOR_GEOMETRY = 4, QUDA_TENSOR_GEOMETRY = 6, QUDA_COARSE_GEOMETRY = 8, QUDA_INVALID_GEOMETRY = QUDA_INVALID_ENUM } QudaFieldGeometry;
typedef enum QudaGhostExchange_s { QUDA_GHOST_EXCHANGE_NO, QUDA_GHOST_EXCHANGE_PAD, QUDA_GHOST_EXCHANGE_EXTENDED, QUDA_GHOST_EXCHANGE_INVALID = QUDA_INVALID_ENUM } QudaGhostExchange;
typedef enum QudaStaggeredPhase_s { QUDA_MILC_STAGGERED_PHASE = 0, QUDA_CPS_STAGGERED_PHASE = 1, QUDA_TIFR_STAGGERED_PHASE = 2, QUDA_INVALID_STAGGERED_PHASE = QUDA_INVALID_ENUM } QudaStaggeredPhase;

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
#endif /* QUDA_AMD_QUDA_H */
