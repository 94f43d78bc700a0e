/*
 * oracle/ — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the reference's host operators for the twisted-mass /
 * twisted-clover Dslash path.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load this library.  The product
 * (quda-qkxtm-multigrid_amd/) never links, imports or calls anything here.
 *
 * Parity is PINNED: every operator below is checked bit-for-bit (fp64) against
 * golden vectors produced by the reference's own host code
 * (tests/wilson_dslash_reference.cpp, tests/clover_reference.cpp) built from
 * /root/reference by oracle/Makefile into oracle/_ref/ (see make_golden.py,
 * tests/golden/).
 *
 * Conventions (all verified against the reference sources cited per function):
 *   - site order: even-odd checkerboard, cb index i = X_lex/2
 *   - host spinor: site-major 24 reals (spin, colour, re/im), DeGrand-Rossi basis
 *   - host gauge: gauge[mu] = V x 18 reals, even half then odd half, row-major 3x3
 *   - host clover: ((parity*Vh+i)*2+chi)*36 reals = 6 diag + 15 complex lower-tri
 */
#ifndef QO_FIELDS_H
#define QO_FIELDS_H

#ifdef __cplusplus
extern "C" {
#endif

/* matpc types and twist kinds follow the reference's enum VALUES
 * (include/enum_quda.h:QudaMatPCType, QudaTwistGamma5Type). */
enum { QO_MATPC_EVEN_EVEN = 0, QO_MATPC_ODD_ODD = 1, QO_MATPC_EVEN_EVEN_ASYM = 2, QO_MATPC_ODD_ODD_ASYM = 3 };
enum { QO_TWIST_DIRECT = 0, QO_TWIST_INVERSE = 1 };

/* geometry: reference tests/test_util.cpp:406-471 */
int qo_full_lattice_index(const int X[4], int i, int oddBit);
int qo_neighbor_index(const int X[4], int i, int oddBit, int dx4, int dx3, int dx2, int dx1);

/* thread count for the "all host cores" CPU baseline (1 = the reference's own
 * single-threaded loop nest; >1 = outer parallel-for over sites). */
void qo_set_threads(int n);
int qo_get_threads(void);

/* ---- double precision ---- */
void qo_wil_dslash_d(double *res, double *const gauge[4], const double *in, int oddBit, int dagger, const int X[4]);
void qo_twist_gamma5_d(double *out, const double *in, int dagger, double kappa, double mu, int flavor, int nsites, int twist);
void qo_tm_dslash_d(double *res, double *const gauge[4], double *in, double kappa, double mu, int flavor,
                    int oddBit, int matpc, int dagger, const int X[4]);
void qo_wil_mat_d(double *out, double *const gauge[4], const double *in, double kappa, int dagger, const int X[4]);
void qo_wil_matpc_d(double *out, double *const gauge[4], const double *in, double kappa, int matpc, int dagger, const int X[4]);
void qo_tm_mat_d(double *out, double *const gauge[4], const double *in, double kappa, double mu, int flavor, int dagger, const int X[4]);
void qo_tm_matpc_d(double *out, double *const gauge[4], double *in, double kappa, double mu, int flavor,
                   int matpc, int dagger, const int X[4]);
void qo_apply_clover_d(double *out, const double *clover, const double *in, int parity, const int X[4]);
void qo_twist_clover_gamma5_d(double *out, const double *in, const double *clover, const double *cinv, int dagger,
                              double kappa, double mu, int flavor, int parity, int twist, const int X[4]);
void qo_tmc_dslash_d(double *out, double *const gauge[4], const double *in, const double *clover, const double *cinv,
                     double kappa, double mu, int flavor, int parity, int matpc, int dagger, const int X[4]);
void qo_tmc_mat_d(double *out, double *const gauge[4], const double *clover, const double *in, double kappa, double mu,
                  int flavor, int dagger, const int X[4]);
void qo_tmc_matpc_d(double *out, double *const gauge[4], const double *in, const double *clover, const double *cinv,
                    double kappa, double mu, int flavor, int matpc, int dagger, const int X[4]);

/* grid-decomposed variant (reference MULTI_GPU branch), see qo_dslash.c */
void qo_wil_dslash_halo_d(double *res, double *const gauge[4], double *const ghost_gauge[4], const double *in,
                          double *const fwd_ghost[4], double *const back_ghost[4], int oddBit, int dagger, const int X[4],
                          const int partitioned[4]);

/* ---- single precision (same loop nests, float arithmetic) ---- */
void qo_wil_dslash_f(float *res, float *const gauge[4], const float *in, int oddBit, int dagger, const int X[4]);
void qo_twist_gamma5_f(float *out, const float *in, int dagger, float kappa, float mu, int flavor, int nsites, int twist);
void qo_tm_dslash_f(float *res, float *const gauge[4], float *in, double kappa, double mu, int flavor,
                    int oddBit, int matpc, int dagger, const int X[4]);
void qo_tm_mat_f(float *out, float *const gauge[4], const float *in, double kappa, double mu, int flavor, int dagger, const int X[4]);
void qo_tm_matpc_f(float *out, float *const gauge[4], float *in, double kappa, double mu, int flavor,
                   int matpc, int dagger, const int X[4]);
void qo_apply_clover_f(float *out, const float *clover, const float *in, int parity, const int X[4]);

/* ---- synthetic inputs (harness side of the reference tests) ---- */
/* random SU(3) links with glibc rand(): tests/test_util.cpp:879-956, scaling/BC :683-706 */
void qo_construct_gauge_field_d(double *const gauge[4], const int X[4], double anisotropy, int antiperiodic_t);
/* uniform(-norm,norm) + diag on the 12 diagonals: tests/test_util.cpp:1100-1120 */
void qo_construct_clover_field_d(double *clover, int V, double norm, double diag);
/* rand()/RAND_MAX per real (what oracle/ref_driver.cpp feeds the reference with) */
void qo_construct_spinor_field_d(double *spinor, int nreal);
void qo_srand(unsigned seed);

/* (A^2 + mu2)^-1 per 6x6 chiral block in the packed host order.  This is the "inverse"
 * field tmc_* expect (tests/clover_reference.cpp:217-224; device twin lib/clover_invert.cu:56-85).
 * Harness input builder (the reference test fetches it from the GPU, dslash_test.cpp:343-349). */
void qo_clover_twisted_inverse_d(double *cinv, const double *clover, int V, double mu2);

/* blas helpers (tests/blas_reference.cpp) */
double qo_norm2_d(const double *v, long n);
void qo_xpay_d(const double *x, double a, double *y, long n);

#ifdef __cplusplus
}
#endif
#endif
