/* oracle/qo_mg.h — TEST INFRASTRUCTURE (see qo_fields.h).  Multigrid pieces of the CPU restatement; layouts and
 * reference citations in qo_mg.c. */
#ifndef QO_MG_H
#define QO_MG_H
#ifdef __cplusplus
extern "C" {
#endif
void qo_mg_fine_to_coarse(int *map, const int X[4], const int geo_bs[4]);
void qo_mg_block_orthogonalize(double *V, const int X[4], const int geo_bs[4], int Ns, int Nc, int Nvec, int spin_bs);
void qo_mg_restrict(double *out, const double *in, const double *V, const int X[4], const int geo_bs[4], int Ns, int Nc, int Nvec, int spin_bs);
void qo_mg_prolongate(double *out, const double *in, const double *V, const int X[4], const int geo_bs[4], int Ns, int Nc, int Nvec, int spin_bs);
void qo_mg_coarse_op_fine(double *Y, double *X_out, const double *V, double *const gauge[4], const double *clover, double kappa, double mu,
                          const int X[4], const int geo_bs[4], int Nvec);
void qo_mg_coarse_op_coarse(double *Y, double *X_out, const double *V, const double *Yf, const double *Xf, double kappa, const int X[4],
                            const int geo_bs[4], int NcF, int Nvec);
void qo_mg_coarse_apply(double *out, const double *in, const double *Y, const double *X, double kappa, const int Xc[4], int Nvec);
void qo_clover_compute_d(double *clover, double *const gauge[4], double coeff, const int X[4]);
#ifdef __cplusplus
}
#endif
#endif
