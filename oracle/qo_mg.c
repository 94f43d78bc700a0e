/*
 * oracle/qo_mg.c — TEST INFRASTRUCTURE (see qo_fields.h): CPU restatement of the reference's multigrid building
 * blocks, the host code paths that live in .cu files and therefore cannot be built here.  Each function
 * cites the reference lines it follows and keeps their loop nest / operation order; arithmetic is double precision
 * (the reference instantiates the same templates in float).
 *
 * Parity status: these pieces have no golden vectors in the reference tree (nvcc-only sources).  They are pinned
 * indirectly — the coarse operator built here from the already bit-pinned fine links must satisfy the reference's
 * own MG::verify identity  R D P = D_c  against qo_tm_mat / qo_tmc_mat (tests/test_oracle_mg.py), and block
 * Gram-Schmidt must give P^dag P = 1.
 *
 * Layouts (reference CPU orders, complex = 2 doubles):
 *   vector   v[(parity*Vh + x_cb) * Ns*Nc + s*Nc + c]                      (QUDA_SPACE_SPIN_COLOR_FIELD_ORDER)
 *   V        V[((parity*Vh + x_cb) * Ns*Nc + s*Nc + c) * Nvec + v]         (colorspinor FieldOrderCB with nVec)
 *   Y        Y[((d*Vc + site) * n + row) * n + col], d 0-3 backward, 4-7 forward, n = 2*Nvec (QDP gauge order)
 *   X        X[(site * n + row) * n + col]
 */
#include "qo_mg.h"

#include <complex.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef double complex cplx;

/* include/index_helper.cuh getCoords: checkerboard index + parity -> coordinates */
static void cb_coords(int x[4], int x_cb, int parity, const int X[4]) {
  const int Xh = X[0] / 2;
  int l = x_cb;
  const int xh = l % Xh; l /= Xh;
  x[1] = l % X[1]; l /= X[1];
  x[2] = l % X[2]; x[3] = l / X[2];
  x[0] = 2 * xh + ((x[1] + x[2] + x[3] + parity) & 1);
}
static int cb_index(const int x[4], const int X[4]) { return ((((x[3] * X[2] + x[2]) * X[1] + x[1]) * X[0] + x[0]) >> 1); }
static int site_parity(const int x[4]) { return (x[0] + x[1] + x[2] + x[3]) & 1; }
static long vol(const int X[4]) { return (long)X[0] * X[1] * X[2] * X[3]; }

/* lib/transfer.cpp:220-240 createGeoMap: fine parity-ordered site -> coarse parity-ordered site */
void qo_mg_fine_to_coarse(int *map, const int X[4], const int geo_bs[4]) {
  int Xc[4];
  for (int d = 0; d < 4; d++) Xc[d] = X[d] / geo_bs[d];
  const long Vh = vol(X) / 2, Vhc = vol(Xc) / 2;
  for (int parity = 0; parity < 2; parity++)
    for (long i = 0; i < Vh; i++) {
      int x[4], xc[4];
      cb_coords(x, (int)i, parity, X);
      for (int d = 0; d < 4; d++) xc[d] = x[d] / geo_bs[d];
      map[parity * Vh + i] = (int)(site_parity(xc) * Vhc + cb_index(xc, Xc));
    }
}

/* lib/transfer_util.cu:168-247 (blockOrderV) + :328-363 (blockGramSchmidt): per (aggregate, chirality) modified
 * Gram-Schmidt over the Nvec columns of V, element order inside a block = (site in block, block spin, colour) */
void qo_mg_block_orthogonalize(double *V_, const int X[4], const int geo_bs[4], int Ns, int Nc, int Nvec, int spin_bs) {
  cplx *V = (cplx *)V_;
  int Xc[4];
  for (int d = 0; d < 4; d++) Xc[d] = X[d] / geo_bs[d];
  const long Vf = vol(X), Vh = Vf / 2, Vc = vol(Xc);
  const int geoBlock = geo_bs[0] * geo_bs[1] * geo_bs[2] * geo_bs[3];
  const int nChi = Ns / spin_bs;
  const int blockSize = geoBlock * Nc * spin_bs;
  int *map = (int *)malloc(Vf * sizeof(int));
  qo_mg_fine_to_coarse(map, X, geo_bs);
  /* block-ordered copy: v[((A*nChi + chi)*Nvec + k)*blockSize + e] */
  cplx *blk = (cplx *)malloc((size_t)Vc * nChi * Nvec * blockSize * sizeof(cplx));
  for (int pass = 0; pass < 2; pass++) {
    if (pass == 1) {
      for (long b = 0; b < Vc * nChi; b++) {
        cplx *v = blk + (size_t)b * Nvec * blockSize;
        for (int jc = 0; jc < Nvec; jc++) {
          for (int ic = 0; ic < jc; ic++) {
            cplx dot = 0.0;
            for (int i = 0; i < blockSize; i++) dot += conj(v[ic * blockSize + i]) * v[jc * blockSize + i];
            for (int i = 0; i < blockSize; i++) v[jc * blockSize + i] -= dot * v[ic * blockSize + i];
          }
          double nrm2 = 0.0;
          for (int i = 0; i < blockSize; i++) nrm2 += creal(v[jc * blockSize + i]) * creal(v[jc * blockSize + i]) + cimag(v[jc * blockSize + i]) * cimag(v[jc * blockSize + i]);
          const double scale = nrm2 > 0.0 ? 1.0 / sqrt(nrm2) : 0.0;
          for (int i = 0; i < blockSize; i++) v[jc * blockSize + i] *= scale;
        }
      }
    }
    for (int parity = 0; parity < 2; parity++)
      for (long x_cb = 0; x_cb < Vh; x_cb++) {
        const long i = parity * Vh + x_cb;
        int x[4];
        cb_coords(x, (int)x_cb, parity, X);
        int blockOffset = 0;
        for (int d = 3; d >= 0; d--) blockOffset = blockOffset * geo_bs[d] + x[d] % geo_bs[d];
        for (int k = 0; k < Nvec; k++)
          for (int s = 0; s < Ns; s++)
            for (int c = 0; c < Nc; c++) {
              const int chi = s / spin_bs, bs = s % spin_bs;
              const size_t idx = (((size_t)map[i] * nChi + chi) * Nvec + k) * blockSize + (size_t)blockOffset * spin_bs * Nc + bs * Nc + c;
              cplx *f = &V[((size_t)i * Ns * Nc + s * Nc + c) * Nvec + k];
              if (pass == 0) blk[idx] = *f; else *f = blk[idx];
            }
      }
  }
  free(blk);
  free(map);
}

/* lib/restrictor.cu:51-125: out(coarse site, s/spin_bs, v) = sum over the aggregate, the spins of the chirality and
 * colour of conj(V) * in */
void qo_mg_restrict(double *out_, const double *in_, const double *V_, const int X[4], const int geo_bs[4], int Ns, int Nc, int Nvec, int spin_bs) {
  cplx *out = (cplx *)out_;
  const cplx *in = (const cplx *)in_, *V = (const cplx *)V_;
  int Xc[4];
  for (int d = 0; d < 4; d++) Xc[d] = X[d] / geo_bs[d];
  const long Vf = vol(X), Vc = vol(Xc);
  const int nChi = Ns / spin_bs;
  int *map = (int *)malloc(Vf * sizeof(int));
  qo_mg_fine_to_coarse(map, X, geo_bs);
  memset(out, 0, (size_t)Vc * nChi * Nvec * sizeof(cplx));
  for (long x = 0; x < Vf; x++)
    for (int v = 0; v < Nvec; v++)
      for (int s = 0; s < Ns; s++) {
        cplx acc = 0.0;
        for (int c = 0; c < Nc; c++) acc += conj(V[((size_t)x * Ns * Nc + s * Nc + c) * Nvec + v]) * in[(size_t)x * Ns * Nc + s * Nc + c];
        out[((size_t)map[x] * nChi + s / spin_bs) * Nvec + v] += acc;
      }
  free(map);
}

/* lib/prolongator.cu:42-116: out(x, s, c) = sum_v V(x, s, c, v) * in(coarse(x), s/spin_bs, v) */
void qo_mg_prolongate(double *out_, const double *in_, const double *V_, const int X[4], const int geo_bs[4], int Ns, int Nc, int Nvec, int spin_bs) {
  cplx *out = (cplx *)out_;
  const cplx *in = (const cplx *)in_, *V = (const cplx *)V_;
  const long Vf = vol(X);
  const int nChi = Ns / spin_bs;
  int *map = (int *)malloc(Vf * sizeof(int));
  qo_mg_fine_to_coarse(map, X, geo_bs);
  for (long x = 0; x < Vf; x++)
    for (int s = 0; s < Ns; s++)
      for (int c = 0; c < Nc; c++) {
        cplx acc = 0.0;
        for (int v = 0; v < Nvec; v++) acc += V[((size_t)x * Ns * Nc + s * Nc + c) * Nvec + v] * in[((size_t)map[x] * nChi + s / spin_bs) * Nvec + v];
        out[(size_t)x * Ns * Nc + s * Nc + c] = acc;
      }
  free(map);
}

/* include/gamma.cuh:31-113, DeGrand-Rossi: row s of gamma_dim has one non-zero element `elem` in column `col` */
static cplx gamma_row(int dim, int s, int *col) {
  static const int coupling[4][4] = {{3, 2, 1, 0}, {3, 2, 1, 0}, {2, 3, 0, 1}, {2, 3, 0, 1}};
  *col = coupling[dim][s];
  switch (dim) {
    case 0: return s < 2 ? I : -I;
    case 1: return (s == 0 || s == 3) ? -1.0 : 1.0;
    case 2: return (s == 0 || s == 3) ? I : -I;
    default: return 1.0;
  }
}

/* element (i, j) of the Hermitian 6x6 chiral block of a packed clover site (tests/clover_reference.cpp:25-53) */
static cplx clover_elem(const double *blk, int i, int j) {
  const int N = 6;
  if (i == j) return blk[i];
  const double *L = blk + N;
  if (j < i) { const int k = N * (N - 1) / 2 - (N - j) * (N - j - 1) / 2 + i - j - 1; return L[2 * k] + I * L[2 * k + 1]; }
  const int k = N * (N - 1) / 2 - (N - i) * (N - i - 1) / 2 + j - i - 1;
  return L[2 * k] - I * L[2 * k + 1];
}

/* shared tail of calculateY for a uni-directional operator (lib/coarse_op.cuh:623-645 reverse, :670-712 local) */
static void reverse_and_local(cplx *Y, cplx *Xm, long Vc, int Nvec, double kappa) {
  const int n = 2 * Nvec;
  for (int d = 0; d < 4; d++)
    for (long A = 0; A < Vc; A++)
      for (int r = 0; r < n; r++)
        for (int c = 0; c < n; c++) {
          const double sign = (r / Nvec == c / Nvec) ? 1.0 : -1.0;
          Y[(((size_t)(d + 4) * Vc + A) * n + r) * n + c] = sign * Y[(((size_t)d * Vc + A) * n + r) * n + c];
        }
  cplx *Xl = (cplx *)malloc((size_t)n * n * sizeof(cplx));
  for (long A = 0; A < Vc; A++) {
    cplx *Xa = Xm + (size_t)A * n * n;
    memcpy(Xl, Xa, (size_t)n * n * sizeof(cplx));
    for (int r = 0; r < n; r++)
      for (int c = 0; c < n; c++) {
        const double sign = (r / Nvec == c / Nvec) ? 1.0 : -1.0;
        /* X = -kappa (sign * X + X^dagger): the reference transposes spin then colour (:678-703) = full Hermitian conjugate */
        Xa[r * n + c] = -kappa * (sign * Xl[r * n + c] + conj(Xl[c * n + r]));
      }
  }
  free(Xl);
}

/* lib/coarse_op.cuh:1310-1498 calculateY for a fine (Ns = 4, Nc = 3) Wilson / twisted-mass / twisted-clover operator,
 * uni-directional branch (the full, non-preconditioned operator the reference's MG coarsens): computeUV :59-125,
 * multiplyVUV/computeVUV :487-600, computeYreverse :623-645, computeCoarseLocal :670-712, computeCoarseClover
 * :732-793 or AddCoarseDiagonal :813-826, AddCoarseTmDiagonal :843-870.  gauge: QDP order as the oracle's Dslash
 * (boundary condition already in the links); clover: packed host order or NULL; mu = 2 kappa mu flavour
 * (lib/dirac_twisted_mass.cpp:207-211). */
void qo_mg_coarse_op_fine(double *Y_, double *X_, const double *V_, double *const gauge[4], const double *clover, double kappa, double mu,
                          const int X[4], const int geo_bs[4], int Nvec) {
  const int Ns = 4, Nc = 3, spin_bs = 2;
  cplx *Y = (cplx *)Y_, *Xm = (cplx *)X_;
  const cplx *V = (const cplx *)V_;
  int Xc[4];
  for (int d = 0; d < 4; d++) Xc[d] = X[d] / geo_bs[d];
  const long Vf = vol(X), Vh = Vf / 2, Vc = vol(Xc), Vhc = Vc / 2;
  const int n = 2 * Nvec;
  memset(Y, 0, (size_t)8 * Vc * n * n * sizeof(cplx));
  memset(Xm, 0, (size_t)Vc * n * n * sizeof(cplx));
  cplx *UV = (cplx *)malloc((size_t)Ns * Nc * Nvec * sizeof(cplx));
  cplx *vuv = (cplx *)malloc((size_t)n * n * sizeof(cplx));
  for (int dim = 0; dim < 4; dim++)
    for (int parity = 0; parity < 2; parity++)
      for (long x_cb = 0; x_cb < Vh; x_cb++) {
        int x[4], y[4], xc[4];
        cb_coords(x, (int)x_cb, parity, X);
        memcpy(y, x, sizeof(y));
        y[dim] = (x[dim] + 1) % X[dim];
        const size_t xs = (size_t)parity * Vh + x_cb, ys = (size_t)(1 - parity) * Vh + cb_index(y, X);
        const cplx *U = (const cplx *)gauge[dim] + xs * 9;
        /* UV(s, ic, v) = sum_jc U(ic, jc) V(x + mu; s, jc, v) */
        for (int s = 0; s < Ns; s++)
          for (int ic = 0; ic < Nc; ic++)
            for (int v = 0; v < Nvec; v++) {
              cplx acc = 0.0;
              for (int jc = 0; jc < Nc; jc++) acc += U[ic * 3 + jc] * V[((ys * Ns + s) * Nc + jc) * Nvec + v];
              UV[(s * Nc + ic) * Nvec + v] = acc;
            }
        /* vuv = V^dagger (1 + gamma_dim) UV  (dir == QUDA_BACKWARDS: positive projector) */
        for (int i = 0; i < n * n; i++) vuv[i] = 0.0;
        for (int s = 0; s < Ns; s++) {
          int s_col;
          const cplx coupling = gamma_row(dim, s, &s_col);
          const int sr = s / spin_bs, sc = s_col / spin_bs;
          for (int ic_c = 0; ic_c < Nvec; ic_c++)
            for (int jc_c = 0; jc_c < Nvec; jc_c++)
              for (int ic = 0; ic < Nc; ic++) {
                const cplx vc = conj(V[((xs * Ns + s) * Nc + ic) * Nvec + ic_c]);
                vuv[(sr * Nvec + ic_c) * n + sr * Nvec + jc_c] += vc * UV[(s * Nc + ic) * Nvec + jc_c];
                vuv[(sr * Nvec + ic_c) * n + sc * Nvec + jc_c] += coupling * vc * UV[(s_col * Nc + ic) * Nvec + jc_c];
              }
        }
        for (int d = 0; d < 4; d++) xc[d] = x[d] / geo_bs[d];
        const int isDiagonal = ((x[dim] + 1) % X[dim]) / geo_bs[dim] == xc[dim];
        const size_t A = (size_t)site_parity(xc) * Vhc + cb_index(xc, Xc);
        cplx *M = isDiagonal ? Xm + A * n * n : Y + ((size_t)dim * Vc + A) * n * n;
        for (int i = 0; i < n * n; i++) M[i] += vuv[i];
      }
  free(UV);
  free(vuv);
  reverse_and_local(Y, Xm, Vc, Nvec, kappa);
  if (clover) {
    /* X(s_c, s_c) += V^dagger C V inside each chirality */
    cplx *CV = (cplx *)malloc((size_t)6 * Nvec * sizeof(cplx));
    for (int parity = 0; parity < 2; parity++)
      for (long x_cb = 0; x_cb < Vh; x_cb++) {
        int x[4], xc[4];
        cb_coords(x, (int)x_cb, parity, X);
        for (int d = 0; d < 4; d++) xc[d] = x[d] / geo_bs[d];
        const size_t xs = (size_t)parity * Vh + x_cb, A = (size_t)site_parity(xc) * Vhc + cb_index(xc, Xc);
        for (int chi = 0; chi < 2; chi++) {
          const double *blk = clover + (xs * 2 + chi) * 36;
          for (int i = 0; i < 6; i++)
            for (int v = 0; v < Nvec; v++) {
              cplx acc = 0.0;
              for (int j = 0; j < 6; j++) acc += clover_elem(blk, i, j) * V[((xs * Ns + 2 * chi) * Nc + j) * Nvec + v];
              CV[i * Nvec + v] = acc;
            }
          for (int ic_c = 0; ic_c < Nvec; ic_c++)
            for (int jc_c = 0; jc_c < Nvec; jc_c++) {
              cplx acc = 0.0;
              for (int i = 0; i < 6; i++) acc += conj(V[((xs * Ns + 2 * chi) * Nc + i) * Nvec + ic_c]) * CV[i * Nvec + jc_c];
              Xm[A * n * n + (size_t)(chi * Nvec + ic_c) * n + chi * Nvec + jc_c] += acc;
            }
        }
      }
    free(CV);
  } else {
    for (long A = 0; A < Vc; A++)
      for (int r = 0; r < n; r++) Xm[(size_t)A * n * n + (size_t)r * n + r] += 1.0;
  }
  if (mu != 0.0)
    for (long A = 0; A < Vc; A++)
      for (int c = 0; c < Nvec; c++) {
        Xm[(size_t)A * n * n + (size_t)c * n + c] += mu * I;
        Xm[(size_t)A * n * n + (size_t)(Nvec + c) * n + Nvec + c] -= mu * I;
      }
}

/* calculateY with from_coarse = true, dirac = QUDA_COARSE_DIRAC (lib/dirac_coarse.cpp:228-232 -> CoarseCoarseOp):
 * the fine operator is itself a coarse operator (Ns = 2, NcF colours, links Yf / Xf in the layout above).  UV uses the
 * backward links Yf[dim] (:88-93), VUV is dense in spin (:552-564), X gets V^dagger Xf V (:776-790) and no diagonal. */
void qo_mg_coarse_op_coarse(double *Y_, double *X_, const double *V_, const double *Yf_, const double *Xf_, double kappa, const int X[4],
                            const int geo_bs[4], int NcF, int Nvec) {
  const int Ns = 2;
  cplx *Y = (cplx *)Y_, *Xm = (cplx *)X_;
  const cplx *V = (const cplx *)V_, *Yf = (const cplx *)Yf_, *Xf = (const cplx *)Xf_;
  int Xc[4];
  for (int d = 0; d < 4; d++) Xc[d] = X[d] / geo_bs[d];
  const long Vf = vol(X), Vh = Vf / 2, Vc = vol(Xc), Vhc = Vc / 2;
  const int n = 2 * Nvec, nf = Ns * NcF;
  memset(Y, 0, (size_t)8 * Vc * n * n * sizeof(cplx));
  memset(Xm, 0, (size_t)Vc * n * n * sizeof(cplx));
  /* UV[(s_col*Ns + s), ic, v] */
  cplx *UV = (cplx *)malloc((size_t)Ns * Ns * NcF * Nvec * sizeof(cplx));
  cplx *vuv = (cplx *)malloc((size_t)n * n * sizeof(cplx));
  for (int dim = 0; dim < 4; dim++)
    for (int parity = 0; parity < 2; parity++)
      for (long x_cb = 0; x_cb < Vh; x_cb++) {
        int x[4], y[4], xc[4];
        cb_coords(x, (int)x_cb, parity, X);
        memcpy(y, x, sizeof(y));
        y[dim] = (x[dim] + 1) % X[dim];
        const size_t xs = (size_t)parity * Vh + x_cb, ys = (size_t)(1 - parity) * Vh + cb_index(y, X);
        const cplx *U = Yf + ((size_t)dim * Vf + xs) * nf * nf;
        for (int s = 0; s < Ns; s++)
          for (int s_col = 0; s_col < Ns; s_col++)
            for (int ic = 0; ic < NcF; ic++)
              for (int v = 0; v < Nvec; v++) {
                cplx acc = 0.0;
                for (int jc = 0; jc < NcF; jc++) acc += U[(size_t)(s * NcF + ic) * nf + s_col * NcF + jc] * V[((ys * Ns + s_col) * NcF + jc) * Nvec + v];
                UV[((size_t)(s_col * Ns + s) * NcF + ic) * Nvec + v] = acc;
              }
        for (int i = 0; i < n * n; i++) vuv[i] = 0.0;
        for (int s_col = 0; s_col < Ns; s_col++)
          for (int s = 0; s < Ns; s++)
            for (int ic_c = 0; ic_c < Nvec; ic_c++)
              for (int jc_c = 0; jc_c < Nvec; jc_c++) {
                cplx acc = 0.0;
                for (int ic = 0; ic < NcF; ic++) acc += conj(V[((xs * Ns + s) * NcF + ic) * Nvec + ic_c]) * UV[((size_t)(s_col * Ns + s) * NcF + ic) * Nvec + jc_c];
                vuv[(size_t)(s * Nvec + ic_c) * n + s_col * Nvec + jc_c] += acc;
              }
        for (int d = 0; d < 4; d++) xc[d] = x[d] / geo_bs[d];
        const int isDiagonal = ((x[dim] + 1) % X[dim]) / geo_bs[dim] == xc[dim];
        const size_t A = (size_t)site_parity(xc) * Vhc + cb_index(xc, Xc);
        cplx *M = isDiagonal ? Xm + A * n * n : Y + ((size_t)dim * Vc + A) * n * n;
        for (int i = 0; i < n * n; i++) M[i] += vuv[i];
      }
  free(UV);
  free(vuv);
  reverse_and_local(Y, Xm, Vc, Nvec, kappa);
  /* coarse clover: X(s, s_col) += V^dagger(s) Xf(s, s_col) V(s_col) */
  cplx *CV = (cplx *)malloc((size_t)nf * Nvec * sizeof(cplx));
  for (long xs = 0; xs < Vf; xs++) {
    int x[4], xc[4];
    const int parity = xs >= Vh;
    cb_coords(x, (int)(xs - parity * Vh), parity, X);
    for (int d = 0; d < 4; d++) xc[d] = x[d] / geo_bs[d];
    const size_t A = (size_t)site_parity(xc) * Vhc + cb_index(xc, Xc);
    const cplx *C = Xf + (size_t)xs * nf * nf;
    for (int s = 0; s < Ns; s++)
      for (int s_col = 0; s_col < Ns; s_col++) {
        for (int ic = 0; ic < NcF; ic++)
          for (int v = 0; v < Nvec; v++) {
            cplx acc = 0.0;
            for (int jc = 0; jc < NcF; jc++) acc += C[(size_t)(s * NcF + ic) * nf + s_col * NcF + jc] * V[(((size_t)xs * Ns + s_col) * NcF + jc) * Nvec + v];
            CV[ic * Nvec + v] = acc;
          }
        for (int ic_c = 0; ic_c < Nvec; ic_c++)
          for (int jc_c = 0; jc_c < Nvec; jc_c++) {
            cplx acc = 0.0;
            for (int ic = 0; ic < NcF; ic++) acc += conj(V[(((size_t)xs * Ns + s) * NcF + ic) * Nvec + ic_c]) * CV[ic * Nvec + jc_c];
            Xm[A * n * n + (size_t)(s * Nvec + ic_c) * n + s_col * Nvec + jc_c] += acc;
          }
      }
  }
  free(CV);
}

/* lib/dslash_coarse.cu:50-203 (applyDslash), :216-234 (applyClover), :265-290 (CPU loop):
 * out(x) = X(x) in(x) - kappa sum_d [ Y_{d+4}(x) in(x + d) + Y_d(x - d)^dagger in(x - d) ] */
void qo_mg_coarse_apply(double *out_, const double *in_, const double *Y_, const double *X_, double kappa, const int Xc[4], int Nvec) {
  cplx *out = (cplx *)out_;
  const cplx *in = (const cplx *)in_, *Y = (const cplx *)Y_, *Xm = (const cplx *)X_;
  const long Vc = vol(Xc), Vh = Vc / 2;
  const int n = 2 * Nvec;
  for (int parity = 0; parity < 2; parity++)
    for (long x_cb = 0; x_cb < Vh; x_cb++) {
      int x[4], y[4];
      cb_coords(x, (int)x_cb, parity, Xc);
      const size_t xs = (size_t)parity * Vh + x_cb;
      for (int row = 0; row < n; row++) {
        cplx acc = 0.0;
        for (int d = 0; d < 4; d++) {
          memcpy(y, x, sizeof(y));
          y[d] = (x[d] + 1) % Xc[d];
          const size_t fs = (size_t)(1 - parity) * Vh + cb_index(y, Xc);
          for (int col = 0; col < n; col++) acc += Y[(((size_t)(d + 4) * Vc + xs) * n + row) * n + col] * in[fs * n + col];
        }
        for (int d = 0; d < 4; d++) {
          memcpy(y, x, sizeof(y));
          y[d] = (x[d] - 1 + Xc[d]) % Xc[d];
          const size_t bs = (size_t)(1 - parity) * Vh + cb_index(y, Xc);
          for (int col = 0; col < n; col++) acc += conj(Y[(((size_t)d * Vc + bs) * n + col) * n + row]) * in[bs * n + col];
        }
        acc *= -kappa;
        for (int col = 0; col < n; col++) acc += Xm[(xs * n + row) * n + col] * in[xs * n + col];
        out[xs * n + row] = acc;
      }
    }
}

/* ---------------------------------------------------------------------------------------------------------------------
 * Clover term from the gauge field (what loadCloverQuda(NULL, NULL, ...) computes on the device): restatement of
 * computeFmunuCore (lib/field_strength_tensor.cu:30-192: F_mu_nu = (Q - Q^dagger)/8 from the four plaquette leaves, pairs
 * mu > nu at index mu(mu-1)/2 + nu) and cloverComputeCore (lib/clover_quda.cu:41-139: the two chiral blocks
 * [[1 - B1, B2^dag], [B2, 1 + B1]] with B1 = i c (F0 -/+ F5), B2 = c (F1 +/- F4 - i (F2 -/+ F3))).  Output: host packed order
 * (6 diagonal reals + 15 lower-triangle complex per chiral block, tests/clover_reference.cpp:25-53) holding the TRUE
 * clover matrix — the reference's native device order stores half of it (lib/clover_quda.cu:128-133).
 * No golden vectors exist for this (nvcc-only sources): "parity unpinned"; tests check unit gauge -> identity,
 * Hermiticity and the independent sigma_mu_nu construction (tests/test_oracle_mg.py).
 * ------------------------------------------------------------------------------------------------------------------- */
static void m3_mul(cplx *c, const cplx *a, const cplx *b) {
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) { cplx s = 0.0; for (int k = 0; k < 3; k++) s += a[i * 3 + k] * b[k * 3 + j]; c[i * 3 + j] = s; }
}
static void m3_dag(cplx *c, const cplx *a) {
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) c[i * 3 + j] = conj(a[j * 3 + i]);
}
/* link U_mu at the site with (unwrapped) coordinates x */
static const cplx *link_at(double *const gauge[4], int mu, const int xin[4], const int X[4]) {
  int x[4];
  for (int d = 0; d < 4; d++) x[d] = ((xin[d] % X[d]) + X[d]) % X[d];
  const long Vh = vol(X) / 2;
  return (const cplx *)gauge[mu] + ((size_t)site_parity(x) * Vh + cb_index(x, X)) * 9;
}
static void plaq4(cplx *out, const cplx *a, int da, const cplx *b, int db, const cplx *c, int dc, const cplx *d, int dd) {
  cplx t[4][9], p[9], q[9];
  const cplx *m[4] = {a, b, c, d};
  const int dg[4] = {da, db, dc, dd};
  for (int k = 0; k < 4; k++) { if (dg[k]) m3_dag(t[k], m[k]); else memcpy(t[k], m[k], sizeof(t[k])); }
  m3_mul(p, t[0], t[1]); m3_mul(q, p, t[2]); m3_mul(out, q, t[3]);
}

void qo_clover_compute_d(double *clover, double *const gauge[4], double coeff, const int X[4]) {
  const long Vh = vol(X) / 2;
  static const int idtab[15] = {0, 1, 3, 6, 10, 2, 4, 7, 11, 5, 8, 12, 9, 13, 14};
  for (int parity = 0; parity < 2; parity++)
    for (long idx = 0; idx < Vh; idx++) {
      int x[4];
      cb_coords(x, (int)idx, parity, X);
      cplx F[6][9];
      for (int mu = 0; mu < 4; mu++)
        for (int nu = 0; nu < mu; nu++) {
          cplx Q[9], L[9];
          int y[4], z[4], w[4];
          /* U(x,mu) U(x+mu,nu) U^dag(x+nu,mu) U^dag(x,nu) */
          memcpy(y, x, sizeof(y)); y[mu]++;
          memcpy(z, x, sizeof(z)); z[nu]++;
          plaq4(Q, link_at(gauge, mu, x, X), 0, link_at(gauge, nu, y, X), 0, link_at(gauge, mu, z, X), 1, link_at(gauge, nu, x, X), 1);
          /* U(x,nu) U^dag(x+nu-mu,mu) U^dag(x-mu,nu) U(x-mu,mu) */
          memcpy(y, x, sizeof(y)); y[nu]++; y[mu]--;
          memcpy(z, x, sizeof(z)); z[mu]--;
          plaq4(L, link_at(gauge, nu, x, X), 0, link_at(gauge, mu, y, X), 1, link_at(gauge, nu, z, X), 1, link_at(gauge, mu, z, X), 0);
          for (int k = 0; k < 9; k++) Q[k] += L[k];
          /* U^dag(x-nu,nu) U(x-nu,mu) U(x+mu-nu,nu) U^dag(x,mu) */
          memcpy(y, x, sizeof(y)); y[nu]--;
          memcpy(z, x, sizeof(z)); z[mu]++; z[nu]--;
          plaq4(L, link_at(gauge, nu, y, X), 1, link_at(gauge, mu, y, X), 0, link_at(gauge, nu, z, X), 0, link_at(gauge, mu, x, X), 1);
          for (int k = 0; k < 9; k++) Q[k] += L[k];
          /* U^dag(x-mu,mu) U^dag(x-mu-nu,nu) U(x-mu-nu,mu) U(x-nu,nu) */
          memcpy(y, x, sizeof(y)); y[mu]--;
          memcpy(z, x, sizeof(z)); z[mu]--; z[nu]--;
          memcpy(w, x, sizeof(w)); w[nu]--;
          plaq4(L, link_at(gauge, mu, y, X), 1, link_at(gauge, nu, z, X), 1, link_at(gauge, mu, z, X), 0, link_at(gauge, nu, w, X), 0);
          for (int k = 0; k < 9; k++) Q[k] += L[k];
          cplx *Fm = F[mu * (mu - 1) / 2 + nu];
          for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Fm[i * 3 + j] = 0.125 * (Q[i * 3 + j] - conj(Q[j * 3 + i]));
        }
      for (int ch = 0; ch < 2; ch++) {
        cplx b1[9], b2[9], tri[15];
        double diag[6];
        for (int k = 0; k < 9; k++) {
          b1[k] = (I * coeff) * (ch == 0 ? F[0][k] - F[5][k] : F[0][k] + F[5][k]);
          b2[k] = coeff * (ch == 0 ? F[1][k] + F[4][k] - I * (F[2][k] - F[3][k]) : F[1][k] - F[4][k] - I * (F[2][k] + F[3][k]));
        }
        for (int i = 0; i < 3; i++) { diag[i] = 1.0 - creal(b1[i * 3 + i]); diag[i + 3] = 1.0 + creal(b1[i * 3 + i]); }
        tri[0] = -b1[1 * 3 + 0];
        tri[1] = -b1[2 * 3 + 0]; tri[2] = -b1[2 * 3 + 1];
        tri[3] = b2[0]; tri[4] = b2[1]; tri[5] = b2[2];
        tri[6] = b2[3]; tri[7] = b2[4]; tri[8] = b2[5]; tri[9] = b1[1 * 3 + 0];
        tri[10] = b2[6]; tri[11] = b2[7]; tri[12] = b2[8]; tri[13] = b1[2 * 3 + 0]; tri[14] = b1[2 * 3 + 1];
        double *A = clover + (((size_t)parity * Vh + idx) * 2 + ch) * 36;
        for (int i = 0; i < 6; i++) A[i] = diag[i];
        for (int i = 0; i < 15; i++) { A[6 + 2 * i] = creal(tri[idtab[i]]); A[6 + 2 * i + 1] = cimag(tri[idtab[i]]); }
      }
    }
}
