// block.h — multi-right-hand-side ("block") fields, BLAS and solver for the multigrid setup.
//
// The reference carries several right-hand sides through its coarse operator as a fifth field dimension
// (lib/dslash_coarse.cu:278, :294-333, src_idx :306-308; benchmarked by tests/multigrid_benchmark_test.cu:253-256) and
// keeps them in composite ColorSpinorFields (include/color_spinor_field.h:84-90 is_composite / composite_dim).  Here the
// consumer is the null-vector generation of the coarse levels (reference MG::generateNullVectors, lib/multigrid.cpp:693-779:
// Nvec independent BiCGstab solves of M x = 0): the Nvec solves run in lockstep on ONE block field, so every application of
// the coarse operator reads its dense link matrices once for all of them and runs on the matrix cores (block.hip).
//
// Layout of a block field: SITE-major, right-hand side fastest —
//     v[((parity * Vh + x_cb) * ncomp + j) * nrhs + i]   (float2, j = spin * Ncolor + colour, i = right-hand side)
// so the ncomp x nrhs panel of a site is one contiguous chunk (9 KB for 48 x 24): the operator stages a neighbour's panel with
// full-line 16-byte loads, and a BLAS thread always meets the same pair of right-hand sides.
// 12-component fields (the fine level's spin-colour vectors, read by fine_block_kernel) are PAIR-MAJOR inside the panel —
//     v[(((x * 6 + j / 2) * nrhs + i) * 2 + j % 2]
// so that the 16-byte word a lane loads is two components of ITS right-hand side (the stencil thread of (site, right-hand side) gets its 12
// components in six loads that still cover whole 128-byte lines across the 8 right-hand sides, with no exchange between lanes: the
// rhs-fastest order needed a DPP transposition per word, a third of the kernel's vector instructions).
#pragma once

#include <vector>

#include "blas.h"
#include "coarse.h"

namespace quda {

constexpr int kMaxBlockRhs = 32;

struct BlockField {
  float2 *v = nullptr;
  int nSites = 0, Vh = 0, ncomp = 0, nrhs = 0;
  // grid-decomposed lattices: nGhost further panels behind the nSites local ones, the faces of the neighbour ranks (BlockGhost below);
  // BLAS and pack / unpack never touch them
  int nGhost = 0;
  bool pairMajor = false;   // ncomp == 12: see above
  size_t bytes = 0;
  BlockField() {}
  BlockField(int nSites, int ncomp, int nrhs, int nGhost = 0);
  BlockField(const BlockField &) = delete;
  BlockField &operator=(const BlockField &) = delete;
  ~BlockField();
  size_t elems() const { return (size_t)nSites * ncomp * nrhs; }
};

// gather / scatter between nrhs ordinary (plane-major, full, fp32) fields and one block field
// parity >= 0: the block holds only that parity half of the fields (nSites = VolumeCB)
void blockPack(BlockField &dst, const std::vector<ColorSpinorField *> &src, int parity = -1);
void blockUnpack(const std::vector<ColorSpinorField *> &dst, const BlockField &src, int parity = -1);
// n <= nrhs single-parity fp32 fine fields (nSpin 4) <-> the columns of one 12-component block field of VolumeCB sites; f[i] == nullptr: a
// column of zeros (pack) / left alone (unpack); accumulate: dst += the fields
void blockPackParity(BlockField &dst, const ColorSpinorField *const *f, int n, bool accumulate = false);
void blockUnpackParity(ColorSpinorField *const *f, int n, const BlockField &src);

// Ghost zone of a block field on a grid-decomposed lattice (reference: the ghost of a multi-source coarse field is a full coarse
// spinor per face site and source, lib/dslash_coarse.cu:68-137; here the same for the multi-right-hand-side fine stencil): per
// partitioned dimension d two zones of faceSites[d] panels, [d][0] the x_d = L - 1 face of the -d neighbour (read by backward hops
// from x_d = 0), [d][1] the x_d = 0 face of the +d neighbour (read by forward hops from x_d = L - 1), at panel index
// nSites + offset[d][k] + face index.  Face index: full fields (coarse levels) the lexicographic index of the other three coordinates;
// single-parity fields (fine level) that index halved, as the reference numbers its face sites.
struct BlockGhost {
  int X[4] = {0, 0, 0, 0};     // local lattice (full extents)
  bool parityField = false;    // the field holds one parity of it (nSites = V / 2)
  int mask = 0;                // bit d: dimension d is partitioned
  int faceSites[4] = {0, 0, 0, 0};
  int offset[4][2] = {};
  int nGhost = 0;
};
BlockGhost blockGhost(const int X[4], bool parityField);
// fills the ghost zone of f: one pack launch, one grouped exchange (commExchange: RCCL send / recv, or a copy where the process is its
// own neighbour) on the compute stream.  parity: the parity f holds (parityField), else ignored.
void blockExchangeGhost(BlockField &f, const BlockGhost &gh, int parity);
void blockExchangeGhostRaw(const float2 *field, float2 *ghostZone, int ncomp, int nrhs, const BlockGhost &gh, int parity);

// out = M in for every right-hand side: X in + sum_d H_d in(x + dhat(d)) with the links read ONCE per site for all of them,
// on v_mfma_f32_16x16x4_f32 (exact fp32).  Needs n = 2 Nc a multiple of 16, nrhs a multiple of 8 (<= 32) and fp32 links;
// blockCoarseSupported() says whether a given operator / batch qualifies.  On a grid-decomposed lattice `in` must carry the ghost zone
// of blockGhost(G.Xc, false) (it is filled here); `in` is therefore not const.
bool blockCoarseSupported(const CoarseGauge &G, int nrhs);
// parity >= 0: only the output sites of that parity are computed (and written) — half the link traffic where the caller wants one parity anyway
void applyCoarseBlock(BlockField &out, BlockField &in, const CoarseGauge &G, int parity = -1);

// [site][9] table of the panel indices of a site's 8 neighbours (order of the link matrices: 2 mu forward, 2 mu + 1 backward) and of
// the site itself: parity * Vh + x_cb, or — across a partitioned face — the ghost panel nSites + offset + face index; device memory,
// built once per coarse lattice and partition mask
const int *coarseNeighbourTable(const int Xc[4]);

// per-right-hand-side BLAS on block fields (coefficients indexed by right-hand side)
namespace blockblas {
void zero(BlockField &x);
void copy(BlockField &dst, const BlockField &src);
void norm2(double *out, const BlockField &x);                                         // out[i] = |x_i|^2
void cDot(Complex *out, const BlockField &x, const BlockField &y);                    // out[i] = (x_i, y_i)
void caxpy(const Complex *a, const BlockField &x, BlockField &y);                     // y_i += a_i x_i
// (t_i, r_i) and |t_i|^2: the omega of BiCGstab
void cDotNormA(Complex *dot, double *norm, const BlockField &t, const BlockField &r);
// x_i += a_i p_i + w_i r_i ; r_i -= w_i t_i ; rho_i = (r0_i, r_i), r2_i = |r_i|^2   (one pass, as blas::caxpbypzYmbw + cDotProductNormB)
void bicgstabUpdate(Complex *rho, double *r2, const Complex *a, const BlockField &p, const Complex *w, BlockField &r, BlockField &x, const BlockField &t,
                    const BlockField &r0);
// p_i = r_i + a_i v_i + b_i p_i   (as blas::cxpaypbz)
void cxpaypbz(const BlockField &r, const Complex *a, const BlockField &v, const Complex *b, BlockField &p);
// (t_i, s_i), |t_i|^2, (r0_i, s_i), (r0_i, t_i) in one pass: omega and, by linearity, rho' = (r0, s - omega t) before r is formed
void bicgstabDots(Complex *ts, double *tt, Complex *r0s, Complex *r0t, const BlockField &t, const BlockField &s, const BlockField &r0);
// x_i += a_i p_i + w_i s_i ; r_i = s_i - w_i t_i (in place of s) ; p_i = r_i + b_i (p_i - w_i v_i) ; r2_i = |r_i|^2   (one pass: 5 reads, 3 writes)
void bicgstabFused(double *r2, const Complex *a, const Complex *w, const Complex *b, BlockField &p, BlockField &r, BlockField &x, const BlockField &t, const BlockField &v);
// minimal-residual step, coefficient from device sums [Re (Ar, r) | Im (Ar, r) | |Ar|^2][nrhs]: x = [x +] alpha rin, r = rin - alpha Ar, alpha = omega (Ar, r) / |Ar|^2
void mrUpdateDev(BlockField &x, BlockField &r, const BlockField &rin, const BlockField &Ar, const double *d_sums, double omega, bool fresh, bool needResidual = true);   // needResidual = false: the last step of a post-smoother, only x is updated
// two minimal-residual steps in one sweep from w1 = A r, w2 = A w1 and the epilogue sums of the two launches (block.hip mr2_update_kernel)
void mr2UpdateDev(BlockField &x, BlockField &r, const BlockField &rin, const BlockField &w1, const BlockField &w2, const double *d_s3, const double *d_s7, double omega, bool fresh,
                  bool needResidual = true);
void xmy(const BlockField &x, BlockField &y);                                         // y = x - y
void negate(BlockField &x);                                                           // x = -x
}  // namespace blockblas

// Nvec BiCGstab solves of M x_i = 0 in lockstep (null-vector mode of the reference's BiCGstab, lib/inv_bicgstab_quda.cpp:96-127:
// b = 0, x_i the initial guesses, b2_i := |M x_i|^2, shadow residual r0 = r), each stopping at |r_i|^2 <= tol^2 b2_i or maxiter.
// Returns the number of iterations of the slowest right-hand side; iters[i] per right-hand side if not null.
typedef void (*BlockMatVec)(BlockField &out, BlockField &in, void *ctx);   // `in` not const: its ghost zone is filled by the operator
// optional: the operator application that also returns inner products of a BiCGstab half step (an operator that has `out` and `in` in
// registers anyway): mode 1: sums = (r0, out) [re | im][nrhs]; mode 2: (out, in) [re | im], |out|^2, (r0, in) [re | im], (r0, out) [re | im]
typedef void (*BlockMatVecDots)(BlockField &out, BlockField &in, void *ctx, const BlockField &r0, int mode, double *sums);
int blockBiCGstabNull(BlockField &x, BlockMatVec mat, void *ctx, double tol, int maxiter, int *iters, BlockMatVecDots matDots = nullptr);

}  // namespace quda
