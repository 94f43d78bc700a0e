// block_solver.cpp — several right-hand sides through MG-preconditioned GCR in lockstep (invertMultiSrcQuda).
//
// The reference declares the entry point (include/quda.h:647, QudaInvertParam::num_src) and carries a multi-source fifth dimension through
// its coarse-grid kernel (lib/dslash_coarse.cu:278, :294-333), but its invertMultiSrcQuda "is just a copy of invertQuda and cannot work"
// (lib/interface_quda.cpp:2546-2549).  The consumer is there all the same: calcMG_threepTwop_EvenOdd solves twelve spin-colour sources per
// flavour with one operator and one hierarchy (lib/interface_quda.cpp:6018-6531).  Here the sources go through the solver TOGETHER:
//   * outer solver: restarted flexible GCR per source with a shared Krylov index (reference lib/inv_gcr_quda.cpp:235-516 per source; a source
//     that has converged simply stops being updated), mixed precision and true-residual restarts as the single-source GCR of solver.cpp;
//   * preconditioner: ONE multigrid cycle for all sources (MG::cycleBlock): smoothing, restriction and prolongation on the fine level per
//     source with the kernels of the single-source cycle, and everything below the fine level on BLOCK FIELDS (block.h: site-major, right-
//     hand side fastest) — the coarse operators read their dense link matrices once for all sources and run on the matrix cores
//     (coarse_block_kernel, v_mfma_f32_16x16x4_f32).
// The even-odd preconditioned coarse operator on block fields needs no kernel of its own: the full block operator applied to a field whose
// other parity is zero IS the parity-restricted hop (sites of one parity only have neighbours of the other), at 2.25 x the necessary flops —
// which the matrix cores have to spare — and the local terms Xinv come out of the same application (slot 8 of the preconditioned links).
#include <cmath>
#include <cstdio>
#include <cstring>
#include <sys/time.h>

#include "block.h"
#include "interface_internal.h"
#include "multigrid.h"

void setTuning(QudaTune tune);   // include/util_quda.h

namespace quda {

void massRescale(ColorSpinorField &b, QudaInvertParam &param);   // solve_interface.cpp

static long long g_msStats[4] = {0, 0, 0, 0};   // qudaAmdMultiSrcStats

static double nowSec() {
  timeval t;
  gettimeofday(&t, nullptr);
  return t.tv_sec + 1e-6 * t.tv_usec;
}

// ================================================================================================
// the cycle below the fine level on block fields
// ================================================================================================
struct BlockLevel {
  const CoarseGauge *Y = nullptr, *H = nullptr;   // links (slot 8 = X) and preconditioned links Xinv Y (slot 8 = Xinv)
  const Transfer *T = nullptr;                    // to the next coarser level (nullptr on the coarsest)
  int p = 0;                                      // parity of the even-odd preconditioned system
  int nuPre = 0, nuPost = 0;
  double omega = 1.0;
  BlockField *b = nullptr, *x = nullptr, *bt = nullptr, *r = nullptr, *Ar = nullptr, *t = nullptr, *w1 = nullptr, *w2 = nullptr;
  std::vector<ColorSpinorField *> fine, coarse;   // per right-hand side staging fields of the transfer to the next level
  // K-cycle (QUDA_MG_CYCLE_RECURSIVE on the level above): this level's system is solved by a flexible GCR around its own cycle (reference
  // lib/multigrid.cpp:230-260: coarse solver = GCR(10), at most 11 iterations, to the smoother tolerance of this level) — source kb, work fields
  bool kSolve = false;
  int kKrylov = 10, kMaxiter = 11;
  double kTol = 0.25;
  BlockField *kb = nullptr, *ky = nullptr, *kr = nullptr;
  std::vector<BlockField *> KP, KAP;
};

class BlockCoarseCycle {
 public:
  int nl = 0, nb = 0;   // levels, right-hand sides of the block (a multiple of 8)
  int nReal = 0;        // columns that carry a source (the rest pad the block: their staging fields are zero from their creation and stay so)
  std::vector<BlockLevel> L;
  // coarsest-grid GCR
  int nKrylov = 20, maxiter = 1000;
  double tol = 0.25;
  std::vector<BlockField *> P, AP;
  BlockField *y = nullptr;
  std::vector<MG *> parents;   // MG object of every level of the sub-hierarchy (creation only)
  long applies = 0, gcrIters = 0, kIters = 0;

  ~BlockCoarseCycle() {
    for (BlockLevel &l : L) {
      for (BlockField *f : {l.b, l.x, l.bt, l.r, l.Ar, l.t, l.w1, l.w2, l.kb, l.ky, l.kr}) delete f;
      for (BlockField *f : l.KP) delete f;
      for (BlockField *f : l.KAP) delete f;
      for (ColorSpinorField *f : l.fine) delete f;
      for (ColorSpinorField *f : l.coarse) delete f;
    }
    for (BlockField *f : P) delete f;
    for (BlockField *f : AP) delete f;
    delete y;
  }

  static void zeroParity(BlockField &f, int par) {
    const size_t half = (size_t)f.Vh * f.ncomp * f.nrhs;
    HIP_CHECK(hipMemsetAsync(f.v + (size_t)par * half, 0, half * sizeof(float2), computeStream()));
  }
  static void copyParity(BlockField &dst, const BlockField &src, int par) {
    const size_t half = (size_t)src.Vh * src.ncomp * src.nrhs;
    HIP_CHECK(hipMemcpyAsync(dst.v + (size_t)par * half, src.v + (size_t)par * half, half * sizeof(float2), hipMemcpyDeviceToDevice, computeStream()));
  }
  void apply(BlockField &out, BlockField &in, const CoarseGauge &G, int parity = -1) { applyCoarseBlock(out, in, G, parity); applies++; }
  // y = x - y
  void xmy(const BlockField &x, BlockField &yv) { blockblas::xmy(x, yv); }

  // out_p = in_p - Yhat_pq Yhat_qp in_p   (in: other parity zero; out: other parity zero)   reference DiracCoarsePC::M, lib/dirac_coarse.cpp:332-350
  // (the two hops are computed on the output parity only: l.t is written on parity q alone, its other half stays zero from its creation)
  void matpc(BlockLevel &l, BlockField &out, BlockField &in) {
    apply(*l.t, in, *l.H, 1 - l.p);
    apply(out, *l.t, *l.H, l.p);
    zeroParity(out, 1 - l.p);
    xmy(in, out);
  }
  // Schur prepare (reference DiracCoarsePC::prepare, :296-330): bt_p = Xinv (b_p - D_pq Xinv b_q)
  void prepare(BlockLevel &l) {
    const int p = l.p, q = 1 - p;
    blockblas::copy(*l.w1, *l.b);
    zeroParity(*l.w1, p);
    apply(*l.w2, *l.w1, *l.H, q);       // parity q: Xinv b_q
    zeroParity(*l.w2, p);
    apply(*l.w1, *l.w2, *l.Y, p);       // parity p: D_pq (Xinv b_q)
    xmy(*l.b, *l.w1);                   // b - ...
    zeroParity(*l.w1, q);
    apply(*l.bt, *l.w1, *l.H, p);       // parity p: Xinv ( . )
    zeroParity(*l.bt, q);
  }
  // x_q = Xinv (b_q - D_qp x_p)   (reference DiracCoarsePC::reconstruct, :352-372); x_q is overwritten
  void reconstruct(BlockLevel &l) {
    const int p = l.p, q = 1 - p;
    zeroParity(*l.x, q);
    apply(*l.w1, *l.x, *l.Y, q);        // parity q: D_qp x_p
    xmy(*l.b, *l.w1);
    zeroParity(*l.w1, p);
    apply(*l.w2, *l.w1, *l.H, q);       // parity q: Xinv ( . )
    copyParity(*l.x, *l.w2, q);
  }
  // MR on the preconditioned system, per right-hand side alpha (reference lib/inv_mr_quda.cpp:40-200)
  void mr(BlockLevel &l, int nu, bool guess) {
    std::vector<Complex> dot(nb), al(nb), mal(nb);
    std::vector<double> nrm(nb);
    if (guess) {
      blockblas::copy(*l.w1, *l.x);
      zeroParity(*l.w1, 1 - l.p);
      matpc(l, *l.r, *l.w1);
      xmy(*l.bt, *l.r);                 // r = bt - Mhat x_p
    } else {
      zeroParity(*l.x, l.p);
      blockblas::copy(*l.r, *l.bt);
    }
    for (int it = 0; it < nu; it++) {
      matpc(l, *l.Ar, *l.r);
      blockblas::cDotNormA(dot.data(), nrm.data(), *l.Ar, *l.r);
      for (int i = 0; i < nb; i++) { al[i] = nrm[i] > 0.0 ? l.omega * dot[i] / nrm[i] : Complex(0.0); mal[i] = -al[i]; }
      blockblas::caxpy(al.data(), *l.r, *l.x);     // x_p += alpha r   (r is zero on the other parity)
      blockblas::caxpy(mal.data(), *l.Ar, *l.r);   // r -= alpha Ar
    }
  }
  // restarted flexible GCR(nK) per right-hand side in lockstep (reference lib/inv_gcr_quda.cpp:235-516): y = A^-1 b to |r| <= tol |b|, p_k = K r
  // (prec == nullptr: p_k = r).  r is work space; op / prec are (out, in)
  template <typename Op, typename Prec>
  int gcrLockstep(BlockField &b, BlockField &yv, BlockField &r, std::vector<BlockField *> &Pv, std::vector<BlockField *> &APv, int nK, int maxit, double tl, Op op, Prec prec) {
    std::vector<double> b2(nb), r2(nb), stop(nb), nrm(nb);
    std::vector<Complex> dot(nb), c(nb);
    blockblas::norm2(b2.data(), b);
    blockblas::copy(r, b);
    blockblas::zero(yv);
    bool any = false;
    for (int i = 0; i < nb; i++) { stop[i] = tl * tl * b2[i]; r2[i] = b2[i]; any = any || b2[i] > 0.0; }
    // coefficients per right-hand side
    std::vector<std::vector<Complex>> alpha(nK, std::vector<Complex>(nb));
    std::vector<std::vector<double>> gamma(nK, std::vector<double>(nb));
    std::vector<std::vector<std::vector<Complex>>> beta(nK, std::vector<std::vector<Complex>>(nK, std::vector<Complex>(nb)));
    auto open = [&]() { for (int i = 0; i < nb; i++) if (r2[i] > stop[i]) return true; return false; };
    int k = 0, total = 0;
    while (any && open() && total < maxit) {
      prec(*Pv[k], r);
      op(*APv[k], *Pv[k]);
      for (int j = 0; j < k; j++) {
        blockblas::cDot(dot.data(), *APv[j], *APv[k]);
        for (int i = 0; i < nb; i++) { beta[j][k][i] = dot[i]; c[i] = -dot[i]; }
        blockblas::caxpy(c.data(), *APv[j], *APv[k]);
      }
      blockblas::cDotNormA(dot.data(), nrm.data(), *APv[k], r);
      for (int i = 0; i < nb; i++) {
        gamma[k][i] = sqrt(nrm[i]);
        alpha[k][i] = gamma[k][i] > 0.0 ? dot[i] / gamma[k][i] : Complex(0.0);
        c[i] = gamma[k][i] > 0.0 ? Complex(1.0 / gamma[k][i] - 1.0) : Complex(0.0);   // AP_k *= 1 / gamma  (y += c x with x = y)
      }
      blockblas::caxpy(c.data(), *APv[k], *APv[k]);
      for (int i = 0; i < nb; i++) c[i] = -alpha[k][i];
      blockblas::caxpy(c.data(), *APv[k], r);
      blockblas::norm2(r2.data(), r);
      k++; total++;
      if (k == nK || total == maxit || !open()) {
        // solution update by back substitution per right-hand side (reference :125-157), then the true residual
        std::vector<std::vector<Complex>> delta(k, std::vector<Complex>(nb));
        for (int i = 0; i < nb; i++)
          for (int a = k - 1; a >= 0; a--) {
            Complex d = alpha[a][i];
            for (int j = a + 1; j < k; j++) d -= beta[a][j][i] * delta[j][i];
            delta[a][i] = gamma[a][i] > 0.0 ? d / gamma[a][i] : Complex(0.0);
          }
        for (int a = 0; a < k; a++) blockblas::caxpy(delta[a].data(), *Pv[a], yv);
        if (total < maxit) {   // true residual of the restart (it decides whether the iteration goes on)
          op(r, yv);
          xmy(b, r);
          blockblas::norm2(r2.data(), r);
        }
        k = 0;
      }
    }
    return total;
  }
  // coarsest grid: GCR(nKrylov) on Mhat x_p = bt
  void gcr(BlockLevel &l) {
    gcrIters += gcrLockstep(*l.bt, *y, *l.r, P, AP, nKrylov, maxiter, tol, [&](BlockField &out, BlockField &in) { matpc(l, out, in); },
                            [&](BlockField &out, BlockField &in) { blockblas::copy(out, in); });
    // x_p = y (other parity of x is rebuilt by reconstruct)
    copyParity(*l.x, *y, l.p);
  }
  // K-cycle: level lev's full system M x = kb by GCR around its own cycle; the solution lands in l.x like a cycle's
  void kgcr(int lev) {
    BlockLevel &l = L[lev];
    kIters += gcrLockstep(*l.kb, *l.ky, *l.kr, l.KP, l.KAP, l.kKrylov, l.kMaxiter, l.kTol, [&](BlockField &out, BlockField &in) { apply(out, in, *l.Y); },
                          [&](BlockField &out, BlockField &in) { blockblas::copy(*l.b, in); cycle(lev); blockblas::copy(out, *l.x); });
    blockblas::copy(*l.x, *l.ky);
  }
  // the solve of level lev as the level above wants it: its cycle, or (K-cycle above) the GCR around it; source in source(lev)
  BlockField &source(int lev) { return L[lev].kSolve ? *L[lev].kb : *L[lev].b; }
  void solve(int lev) { if (L[lev].kSolve) kgcr(lev); else cycle(lev); }

  // x = cycle(b) on level 0 of the sub-hierarchy; fields of L[0].b / L[0].x are filled / read by the caller
  void cycle(int lev = 0) {
    BlockLevel &l = L[lev];
    if (lev == nl - 1) {
      prepare(l);
      gcr(l);
      reconstruct(l);
      return;
    }
    prepare(l);
    mr(l, l.nuPre, false);
    BlockField *full = l.r;
    if (l.nuPre > 0) {
      // the full residual behind the even-odd pre-smoother is X_pp r~ on the solved parity and zero on the other (MG::imageOfLast): the operator on
      // MR's residual (zero on the other parity), output parity p only — one half application instead of reconstruct + full operator (four halves)
      apply(*l.w1, *l.r, *l.Y, l.p);
      zeroParity(*l.w1, 1 - l.p);
      full = l.w1;
    } else {
      reconstruct(l);
      apply(*l.r, *l.x, *l.Y);            // r = b - M x on every site
      xmy(*l.b, *l.r);
    }
    // restrict / prolongate per right-hand side with the single-vector kernels of the level (aggregates of the coarse levels are tiny)
    BlockLevel &c = L[lev + 1];
    blockUnpack(l.fine, *full);
    for (int i = 0; i < nReal; i++) l.T->R(*l.coarse[i], *l.fine[i]);
    blockPack(source(lev + 1), l.coarse);
    solve(lev + 1);
    blockUnpack(l.coarse, *c.x);
    for (int i = 0; i < nReal; i++) l.T->P(*l.fine[i], *l.coarse[i]);
    blockPack(*l.w1, l.fine);
    std::vector<Complex> one(nb, Complex(1.0, 0.0));
    blockblas::caxpy(one.data(), *l.w1, *l.x);
    mr(l, l.nuPost, true);
    reconstruct(l);
  }
};

static BlockCoarseCycle *blockCoarseCreate(MG &top, int nb, const MGParam *parent) {
  const MGParam &tp = top.params();
  if (tp.level < 1) return nullptr;
  BlockCoarseCycle *bc = new BlockCoarseCycle;
  bc->nl = tp.Nlevel - tp.level;
  bc->nb = nb;
  bc->L.resize(bc->nl);
  MG *m = &top;
  for (int l = 0; l < bc->nl; l++, m = m->getCoarse()) {
    if (!m) { delete bc; return nullptr; }
    const MGParam &p = m->params();
    const bool coarsest = p.level == p.Nlevel - 1;
    const DiracCoarse *dc = dynamic_cast<const DiracCoarse *>(p.matResidual.Expose());
    const DiracCoarsePC *ds = dynamic_cast<const DiracCoarsePC *>(p.matSmooth.Expose());
    if (!dc || !ds || !m->smootherIsPC() || !blockCoarseSupported(dc->Links(), nb)) { delete bc; return nullptr; }
    BlockLevel &L = bc->L[l];
    {
      // the level above runs a K-cycle and this level is not the coarsest: GCR around this level's cycle (parameters of MG's coarse solver, multigrid.cpp)
      const MGParam *pp = l == 0 ? parent : &bc->parents[l - 1]->params();
      L.kSolve = !coarsest && pp && pp->cycle_type != QUDA_MG_CYCLE_VCYCLE;
      L.kTol = p.mg_global.smoother_tol[p.level];
    }
    bc->parents.push_back(m);
    L.Y = &dc->Links(); L.H = &ds->HatLinks();
    const QudaMatPCType pc = ds->getMatPCType();
    L.p = (pc == QUDA_MATPC_ODD_ODD) ? 1 : 0;
    L.nuPre = p.nu_pre; L.nuPost = p.nu_post; L.omega = p.omega;
    const int nGhost = blockGhost(L.Y->Xc, false).nGhost;
    for (BlockField **f : {&L.b, &L.x, &L.bt, &L.r, &L.Ar, &L.t, &L.w1, &L.w2}) { *f = new BlockField(L.Y->nSites, L.Y->n, nb, nGhost); blockblas::zero(**f); }
    if (L.kSolve) {
      for (BlockField **f : {&L.kb, &L.ky, &L.kr}) { *f = new BlockField(L.Y->nSites, L.Y->n, nb, nGhost); blockblas::zero(**f); }
      for (int k = 0; k < L.kKrylov; k++) {
        L.KP.push_back(new BlockField(L.Y->nSites, L.Y->n, nb, nGhost)); L.KAP.push_back(new BlockField(L.Y->nSites, L.Y->n, nb, nGhost));
        blockblas::zero(*L.KP.back()); blockblas::zero(*L.KAP.back());
      }
    }
    if (!coarsest) {
      L.T = m->getTransfer();
      if (!L.T) { delete bc; return nullptr; }
      for (int i = 0; i < nb; i++) { L.fine.push_back(L.T->createFineField()); L.coarse.push_back(L.T->createCoarseField()); }
    } else {
      const SolverParam *sp = m->preSmootherParam();
      bc->nKrylov = sp->Nkrylov; bc->maxiter = sp->maxiter; bc->tol = sp->tol;
      for (int k = 0; k < bc->nKrylov; k++) {
        bc->P.push_back(new BlockField(L.Y->nSites, L.Y->n, nb, nGhost)); bc->AP.push_back(new BlockField(L.Y->nSites, L.Y->n, nb, nGhost));
        blockblas::zero(*bc->P.back()); blockblas::zero(*bc->AP.back());
      }
      bc->y = new BlockField(L.Y->nSites, L.Y->n, nb, nGhost);
      blockblas::zero(*bc->y);
    }
  }
  return bc;
}

// ================================================================================================
// the smoother of the fine level for all sources: MR on the even-odd preconditioned operator, on block fields
// ================================================================================================
// The per-source smoother streams the gauge links once per source and step (384 - 576 B per site against 192 B of spinor), and every step is a
// chain of five launches per source.  Here groups of 8 (or 4) sources share one pass over the links (fine_block_kernel: dslash.h
// applyFineBlockParity), the two sums of the MR step come out of the second launch's epilogue where `A r` and `r` already sit in registers
// (mode 3), and the coefficient never leaves the device (fineBlockDotsFinishDev -> blockblas::mrUpdateDev) — the same arithmetic as MR of
// solver.cpp with device-side scalars (fp32 fields, fp64 sums, rank-local), reference lib/inv_mr_quda.cpp:40-200.
struct FineGroup {
  int first = 0, n = 0, nrhs = 0;   // sources [first, first + n) are the first n columns of blocks with nrhs columns
  BlockField *B = nullptr, *X = nullptr, *R = nullptr, *AR = nullptr, *T = nullptr, *W2 = nullptr;
  double *d_sums = nullptr, *d_sums7 = nullptr;
};
class BlockFineSmoother {
 public:
  const GaugeField *U = nullptr;
  double kappa = 0, a = 0, binv = 1, omega = 1, mu = 0;
  QudaDiracType type = QUDA_INVALID_DIRAC;
  QudaMatPCType matpcType = QUDA_MATPC_INVALID;
  int par = 0;                         // parity of the preconditioned system
  int nuPre = 0, nuPost = 0;
  bool globalSums = false;             // the MR sums cross ranks
  bool twoStep = true;                 // QUDA_AMD_MULTISRC_MR_PAIRS=0: one update per step
  float *tmat[2] = {nullptr, nullptr};   // twisted clover: dense (A + i a g5)^-1 per parity
  size_t tmatBytes = 0;
  std::vector<FineGroup> groups;

  ~BlockFineSmoother() {
    for (FineGroup &g : groups) {
      for (BlockField *f : {g.B, g.X, g.R, g.AR, g.T, g.W2}) delete f;
      if (g.d_sums) poolDeviceFree(g.d_sums, 0);
      if (g.d_sums7) poolDeviceFree(g.d_sums7, 0);
    }
    for (int p = 0; p < 2; p++) if (tmat[p]) poolDeviceFree(tmat[p], tmatBytes);
  }
  static float2 *ghostOf(BlockField &f) { return f.nGhost ? f.v + f.elems() : nullptr; }
  // out = in - kappa^2 A^-1 D_pq A^-1 D_qp in     (reference DiracTwistedMassPC::M / DiracTwistedCloverPC::M, symmetric preconditioning);
  // dots: (out, in) and |out|^2 per right-hand side into g.d_sums
  // dots: 0 none; 3: (out, in), |out|^2 -> g.d_sums; 2: the seven sums with a = `dotA` -> g.d_sums7 (rank-local device sums; sums across ranks go the mode-3 way only)
  void matpc(FineGroup &g, BlockField &out, BlockField &in, int dots, const GaugeField *links = nullptr, const BlockField *dotA = nullptr) {
    const FineBlockDots d = {dots == 2 ? dotA->v : nullptr, dots == 2 ? 2 : 3};
    const int p = par, q = 1 - par;
    const GaugeField &W = links ? *links : *U;
    if (tmat[0]) {
      applyFineBlockParity(g.T->v, nullptr, in.v, in.nrhs, W, q, 0.0, 0.0, 1.0, 0.0, tmat[q], 1, ghostOf(in));
      applyFineBlockParity(out.v, in.v, g.T->v, in.nrhs, W, p, 1.0, 0.0, -kappa * kappa, 0.0, tmat[p], 1, ghostOf(*g.T), dots ? &d : nullptr);
    } else {
      applyFineBlockParity(g.T->v, nullptr, in.v, in.nrhs, W, q, 0.0, 0.0, binv, -a, nullptr, 0, ghostOf(in));
      applyFineBlockParity(out.v, in.v, g.T->v, in.nrhs, W, p, 1.0, 0.0, -kappa * kappa * binv, -a, nullptr, 0, ghostOf(*g.T), dots ? &d : nullptr);
    }
    if (!dots) return;
    if (dots == 2) { fineBlockDotsFinishDev(g.d_sums7, in.nrhs, 2); return; }
    if (!globalSums) { fineBlockDotsFinishDev(g.d_sums, in.nrhs, 3); return; }
    // sums over all ranks (smoother with global_reduction on a grid-decomposed lattice): through the host, one round trip per step and group
    double sums[3 * 8];
    fineBlockDotsFinish(sums, in.nrhs, 3);
    HIP_CHECK(hipMemcpyAsync(g.d_sums, sums, 3 * in.nrhs * sizeof(double), hipMemcpyHostToDevice, computeStream()));
    HIP_CHECK(hipStreamSynchronize(computeStream()));
  }
  // nu MR steps on (X, R); fresh: X is not defined yet and the residual is `first` (the source itself), not R
  void mr(FineGroup &g, int nu, BlockField *first, bool needResidual) {
    int k = 0;
    // pairs of steps as one update (blockblas::mr2UpdateDev: w1 = A r, w2 = A w1, both coefficients from the epilogue sums) while two or more are left
    while (twoStep && !globalSums && k + 1 < nu) {
      BlockField &rin = (k == 0 && first) ? *first : *g.R;
      matpc(g, *g.AR, rin, 3);
      matpc(g, *g.W2, *g.AR, 2, nullptr, &rin);
      blockblas::mr2UpdateDev(*g.X, *g.R, rin, *g.AR, *g.W2, g.d_sums, g.d_sums7, omega, k == 0 && first, needResidual || k + 2 < nu);
      k += 2;
    }
    for (; k < nu; k++) {
      BlockField &rin = (k == 0 && first) ? *first : *g.R;
      matpc(g, *g.AR, rin, 3);
      blockblas::mrUpdateDev(*g.X, *g.R, rin, *g.AR, g.d_sums, omega, k == 0 && first, needResidual || k + 1 < nu);
    }
  }
  // R = B - M X
  void residual(FineGroup &g) {
    matpc(g, *g.AR, *g.X, 0);
    std::vector<Complex> zero(g.nrhs, Complex(0.0, 0.0)), mone(g.nrhs, Complex(-1.0, 0.0));
    blockblas::cxpaypbz(*g.B, zero.data(), *g.B, mone.data(), *g.AR);   // AR = B - AR
    std::swap(g.R, g.AR);
  }
};

static bool blockFineSmootherOn() {
  static int on = -1;
  if (on < 0) { const char *e = getenv("QUDA_AMD_MULTISRC_BLOCK_SMOOTHER"); on = (e && !atoi(e)) ? 0 : 1; }
  return on != 0;
}

// nullptr: this fine level keeps the per-source smoother (asymmetric preconditioning, a smoother other than MR, 16-bit smoothing, links the
// multi-right-hand-side stencil does not read)
static BlockFineSmoother *blockFineCreate(const Dirac &dirac, const SolverParam &pre, const SolverParam &post, int flavor, int nsrc) {
  if (!blockFineSmootherOn()) return nullptr;
  const QudaDiracType ty = dirac.getDiracType();
  if (ty != QUDA_TWISTED_MASSPC_DIRAC && ty != QUDA_WILSONPC_DIRAC && ty != QUDA_TWISTED_CLOVERPC_DIRAC) return nullptr;
  const QudaMatPCType mt = dirac.getMatPCType();
  if (mt != QUDA_MATPC_EVEN_EVEN && mt != QUDA_MATPC_ODD_ODD) return nullptr;
  if (pre.inv_type != QUDA_MR_INVERTER || post.inv_type != QUDA_MR_INVERTER) return nullptr;
  if (pre.precision_sloppy != QUDA_SINGLE_PRECISION || post.precision_sloppy != QUDA_SINGLE_PRECISION) return nullptr;
  const GaugeField *U = dirac.Gauge();
  if (!U || !fineBlockSupported(*U, 8) || !fineBlockSupported(*U, 4) || !fineBlockDotsSupported(8) || !fineBlockDotsSupported(4)) return nullptr;
  const bool tmc = ty == QUDA_TWISTED_CLOVERPC_DIRAC;
  if (tmc && !(dirac.Clover() && dirac.Clover()->precision == QUDA_SINGLE_PRECISION)) return nullptr;
  BlockFineSmoother *f = new BlockFineSmoother;
  f->U = U;
  f->kappa = dirac.Kappa(); f->mu = dirac.Mu(); f->type = ty; f->matpcType = mt;
  f->a = ty == QUDA_WILSONPC_DIRAC ? 0.0 : 2.0 * dirac.Kappa() * (double)flavor * dirac.Mu();
  f->binv = 1.0 / (1.0 + f->a * f->a);
  f->omega = pre.omega;
  f->par = mt == QUDA_MATPC_ODD_ODD ? 1 : 0;
  f->nuPre = pre.maxiter; f->nuPost = post.maxiter;
  f->globalSums = pre.global_reduction && commReductionsNeeded();
  { const char *e = getenv("QUDA_AMD_MULTISRC_MR_PAIRS"); f->twoStep = !(e && !atoi(e)); }
  const int Vh = U->geom.Vh;
  if (tmc) {
    f->tmatBytes = (size_t)Vh * 144 * sizeof(float);
    for (int p = 0; p < 2; p++) {
      f->tmat[p] = (float *)poolDeviceMalloc(f->tmatBytes);
      cloverTwistDense(f->tmat[p], *dirac.Clover(), p, f->a, true);
    }
  }
  const int nGhost = blockGhost(U->geom.X, true).nGhost;
  for (int first = 0; first < nsrc;) {
    FineGroup g;
    const int left = nsrc - first;
    g.first = first; g.n = left >= 8 ? 8 : left; g.nrhs = g.n > 4 ? 8 : 4;
    g.B = new BlockField(Vh, 12, g.nrhs, nGhost); g.X = new BlockField(Vh, 12, g.nrhs, nGhost); g.R = new BlockField(Vh, 12, g.nrhs, nGhost);
    g.AR = new BlockField(Vh, 12, g.nrhs, nGhost); g.T = new BlockField(Vh, 12, g.nrhs, nGhost); g.W2 = new BlockField(Vh, 12, g.nrhs, nGhost);
    g.d_sums = (double *)poolDeviceMalloc(3 * 8 * sizeof(double));
    g.d_sums7 = (double *)poolDeviceMalloc(7 * 8 * sizeof(double));
    f->groups.push_back(g);
    first += g.n;
  }
  return f;
}

// ================================================================================================
// the multigrid cycle for several sources at once (level 0: per source; below: block fields)
// ================================================================================================
struct MGBlockState {
  int nsrc = 0, nb = 0;
  std::vector<Solver *> pre, post;
  std::vector<SolverParam *> prePar, postPar;
  std::vector<ColorSpinorField *> r, rc, xc, btilde;
  BlockCoarseCycle *coarse = nullptr;
  BlockFineSmoother *fine = nullptr;   // nullptr: the per-source smoothers above
  bool solutionOnBlocks = false;       // the last cycle left its solutions in fine->groups[].X (blockApplyLast)
  bool wantImage = false;              // the caller will ask for A x of the cycle's solutions: the post-smoother keeps its last residual
  bool residualOnBlocks = false;       // ... and did: fine->groups[].R = b - A x
  ~MGBlockState() {
    delete fine;
    for (Solver *s : pre) delete s;
    for (Solver *s : post) delete s;
    for (SolverParam *s : prePar) delete s;
    for (SolverParam *s : postPar) delete s;
    for (ColorSpinorField *f : r) delete f;
    for (ColorSpinorField *f : rc) delete f;
    for (ColorSpinorField *f : xc) delete f;
    for (ColorSpinorField *f : btilde) delete f;
    delete coarse;
  }
};

// QUDA_AMD_MULTISRC_QUAD=0: restrict / prolong the sources one at a time (the comparison leg of tools/multisrc_timing.py)
static bool blockQuadTransfer() {
  static int on = -1;
  if (on < 0) { const char *e = getenv("QUDA_AMD_MULTISRC_QUAD"); on = (e && !atoi(e)) ? 0 : 1; }
  return on != 0;
}

// QUDA_AMD_MULTISRC_DIRECT=0: the quads go through fields (unpack, R4 / P4, pack) instead of their block columns
static bool blockDirectTransfer() {
  static int on = -1;
  if (on < 0) { const char *e = getenv("QUDA_AMD_MULTISRC_DIRECT"); on = (e && !atoi(e)) ? 0 : 1; }
  return on != 0;
}

void MG::blockRelease() { delete blockState; blockState = nullptr; }

bool MG::blockPrepare(int nsrc) {
  if (blockState && blockState->nsrc == nsrc) return true;
  blockRelease();
  if (mgp.level != 0 || mgp.level == mgp.Nlevel - 1 || !pcSmooth || !coarse) return false;
  const int nb = (nsrc + 7) / 8 * 8;
  if (nb > kMaxBlockRhs) return false;
  MGBlockState *st = new MGBlockState;
  st->nsrc = nsrc; st->nb = nb;
  // nullptr (a K-cycle below the first coarse level, an operator the MFMA kernel does not take): the coarse solves run source by source through
  // the hierarchy's own coarse solver, the fine level keeps its block smoother and four-source transfers
  st->coarse = blockCoarseCreate(*coarse, nb, &mgp);
  if (st->coarse) st->coarse->nReal = nsrc;
  for (int i = 0; i < nsrc; i++) {
    st->prePar.push_back(new SolverParam(*param_presmooth));
    st->postPar.push_back(new SolverParam(*param_postsmooth));
    st->pre.push_back(Solver::create(*st->prePar.back(), mgp.matSmooth, mgp.matSmooth, mgp.matSmooth));
    st->post.push_back(Solver::create(*st->postPar.back(), mgp.matSmooth, mgp.matSmooth, mgp.matSmooth));
    ColorSpinorParam cp = r->param();
    cp.create = QUDA_ZERO_FIELD_CREATE;
    st->r.push_back(new ColorSpinorField(cp));
    ColorSpinorParam bp = r->Even().param();
    bp.create = QUDA_ZERO_FIELD_CREATE;
    st->btilde.push_back(new ColorSpinorField(bp));
  }
  for (int i = 0; i < nb; i++) { st->rc.push_back(transfer->createCoarseField()); st->xc.push_back(transfer->createCoarseField()); }
  st->fine = blockFineCreate(*mgp.matSmooth.Expose(), *param_presmooth, *param_postsmooth, (int)mgp.fineFlavor, nsrc);
  blockState = st;
  return true;
}

// the parity cycle of MG::cycleParity for all sources: x_i = K b_i (single-parity fields)
void MG::cycleParityBlock(std::vector<ColorSpinorField *> &x, std::vector<ColorSpinorField *> &b, const std::vector<char> &active, bool fullResidual) {
  MGBlockState &st = *blockState;
  const int nsrc = st.nsrc;
  const Dirac &dirac = *mgp.matSmooth.Expose();
  const bool odd = dirac.getMatPCType() == QUDA_MATPC_ODD_ODD || dirac.getMatPCType() == QUDA_MATPC_ODD_ODD_ASYMMETRIC;
  const bool symmetric = dirac.getMatPCType() == QUDA_MATPC_EVEN_EVEN || dirac.getMatPCType() == QUDA_MATPC_ODD_ODD;
  std::vector<const ColorSpinorField *> rin(nsrc, nullptr);
  // smoothing of all sources on block fields where the fine level allows it (BlockFineSmoother), else source by source
  bool blockSmooth = st.fine != nullptr;
  for (int i = 0; i < nsrc && blockSmooth; i++)
    if (active[i] && (b[i]->Precision() != QUDA_SINGLE_PRECISION || x[i]->Precision() != QUDA_SINGLE_PRECISION || (int)b[i]->twistFlavor != (int)mgp.fineFlavor)) blockSmooth = false;
  g_msStats[0]++;
  if (blockSmooth) g_msStats[1]++;
  st.solutionOnBlocks = blockSmooth;
  st.residualOnBlocks = blockSmooth && st.wantImage && st.fine->nuPost > 0;
  // quads of active sources for the four-source transfers; with the block smoother a quad sits inside one smoother group and (unless the local term has
  // to be applied to the residual first) is restricted from / prolongated onto its block columns directly
  const bool quad = transfer->canQuad() && blockQuadTransfer();
  const bool blockDirect = blockSmooth && quad && !(fullResidual && symmetric) && blockDirectTransfer();
  std::vector<char> quadAt(nsrc + 4, 0), inQuad(nsrc + 4, 0);
  auto groupOf = [&](int i, int &col0) -> FineGroup * {
    for (FineGroup &g : st.fine->groups) if (i >= g.first && i + 3 < g.first + g.n) { col0 = i - g.first; return &g; }
    return nullptr;
  };
  auto planQuads = [&](bool fieldsKnown) {
    for (int i = 0; quad && i + 4 <= nsrc;) {
      bool ok = true;
      int col0;
      for (int s = 0; s < 4; s++) ok = ok && active[i + s] && (!fieldsKnown || rin[i + s]->Precision() == QUDA_SINGLE_PRECISION);
      if (ok && blockSmooth && !groupOf(i, col0)) ok = false;
      if (ok) { quadAt[i] = 1; for (int s = 0; s < 4; s++) inQuad[i + s] = 1; i += 4; } else i++;
    }
  };
  if (blockSmooth) planQuads(false);
  if (blockSmooth) {
    BlockFineSmoother &F = *st.fine;
    for (FineGroup &g : F.groups) {
      const ColorSpinorField *src[8];
      ColorSpinorField *dst[8];
      bool anyField = false;
      for (int j = 0; j < g.n; j++) {
        const int i = g.first + j;
        src[j] = active[i] ? b[i] : nullptr;
        dst[j] = nullptr;
        if (!active[i]) continue;
        x[i]->twistFlavor = b[i]->twistFlavor;
        ColorSpinorField &rp = odd ? st.r[i]->Odd() : st.r[i]->Even();
        st.r[i]->twistFlavor = rp.twistFlavor = b[i]->twistFlavor;
        if (!(blockDirect && inQuad[i])) { dst[j] = &rp; anyField = true; }
      }
      blockPackParity(*g.B, src, g.n);
      if (F.nuPre > 0) {
        F.mr(g, F.nuPre, g.B, true);
        if (anyField) blockUnpackParity(dst, g.n, *g.R);
      }
      for (int j = 0; j < g.n; j++) {
        const int i = g.first + j;
        if (!active[i] || (blockDirect && inQuad[i])) continue;
        if (F.nuPre > 0) {
          if (fullResidual && symmetric) dirac.localTermParity(*dst[j], *dst[j], odd ? 1 : 0);
          rin[i] = dst[j];
        } else if (fullResidual && symmetric) {
          dirac.localTermParity(*dst[j], *b[i], odd ? 1 : 0);
          rin[i] = dst[j];
        } else {
          rin[i] = b[i];
        }
      }
    }
  }
  for (int i = 0; i < nsrc && !blockSmooth; i++) {
    if (!active[i]) continue;
    x[i]->twistFlavor = b[i]->twistFlavor;
    ColorSpinorField &rp = odd ? st.r[i]->Odd() : st.r[i]->Even();
    st.r[i]->twistFlavor = rp.twistFlavor = b[i]->twistFlavor;
    (*st.pre[i])(*x[i], *b[i]);
    const ColorSpinorField *res = mgp.nu_pre > 0 ? st.pre[i]->lastResidual() : nullptr;
    if (res && res->Precision() == QUDA_SINGLE_PRECISION && !(fullResidual && symmetric)) {
      rin[i] = res;
    } else if (res) {
      if (fullResidual && symmetric && res->Precision() == QUDA_SINGLE_PRECISION) dirac.localTermParity(rp, *res, odd ? 1 : 0);
      else { blas::copy(rp, *res); if (fullResidual && symmetric) dirac.localTermParity(rp, rp, odd ? 1 : 0); }
      rin[i] = &rp;
    } else {
      mgp.matSmooth(rp, *x[i]);
      blas::axpby(1.0, *b[i], -1.0, rp);
      if (fullResidual && symmetric) dirac.localTermParity(rp, rp, odd ? 1 : 0);
      rin[i] = &rp;
    }
  }
  transfer->setSiteSubset(QUDA_PARITY_SITE_SUBSET, odd ? QUDA_ODD_PARITY : QUDA_EVEN_PARITY);
  // V (2304 B per fine site) is the traffic of the restrictor and the prolongator: groups of four active sources share one pass over it
  if (!blockSmooth) planQuads(true);
  for (int i = 0; i < st.nb;) {
    if (i < nsrc && quadAt[i]) {
      ColorSpinorField *c4[4] = {st.rc[i], st.rc[i + 1], st.rc[i + 2], st.rc[i + 3]};
      if (blockDirect) {
        int col0;
        FineGroup *g = groupOf(i, col0);
        transfer->R4Block(c4, (st.fine->nuPre > 0 ? g->R : g->B)->v, g->nrhs, col0);
      } else {
        const ColorSpinorField *f4[4] = {rin[i], rin[i + 1], rin[i + 2], rin[i + 3]};
        transfer->R4(c4, f4);
      }
      g_msStats[2]++;
      i += 4;
      continue;
    }
    if (i < nsrc && active[i]) transfer->R(*st.rc[i], *rin[i]);
    else blas::zero(*st.rc[i]);
    i++;
  }
  if (st.coarse) {
    // everything below the fine level for all sources at once, on the matrix cores
    blockPack(st.coarse->source(0), st.rc);
    st.coarse->solve(0);
    blockUnpack(st.xc, *st.coarse->L[0].x);
  } else {
    for (int i = 0; i < nsrc; i++) {
      if (!active[i]) continue;
      blas::zero(*st.xc[i]);
      (*coarse_solver)(*st.xc[i], *st.rc[i]);
    }
  }
  // prolongation: sources outside the quads (and all of them without the block smoother) through fields ...
  for (int i = 0; i < nsrc;) {
    if (quadAt[i]) {
      if (!blockDirect) {
        ColorSpinorField *f4[4];
        const ColorSpinorField *c4[4] = {st.xc[i], st.xc[i + 1], st.xc[i + 2], st.xc[i + 3]};
        for (int s = 0; s < 4; s++) f4[s] = odd ? &st.r[i + s]->Odd() : &st.r[i + s]->Even();
        transfer->P4(f4, c4);
        g_msStats[2]++;
        if (!blockSmooth) for (int s = 0; s < 4; s++) blas::xpy(*f4[s], *x[i + s]);
      }
      i += 4;
      continue;
    }
    if (active[i]) {
      ColorSpinorField &rp = odd ? st.r[i]->Odd() : st.r[i]->Even();
      transfer->P(rp, *st.xc[i]);
      if (!blockSmooth) blas::xpy(rp, *x[i]);
    }
    i++;
  }
  if (blockSmooth) {
    BlockFineSmoother &F = *st.fine;
    for (FineGroup &g : F.groups) {
      const ColorSpinorField *corr[8];
      ColorSpinorField *dst[8];
      bool viaFields = false;
      for (int j = 0; j < g.n; j++) {
        const int i = g.first + j;
        const bool direct = blockDirect && inQuad[i];
        corr[j] = active[i] && !direct ? (odd ? &st.r[i]->Odd() : &st.r[i]->Even()) : nullptr;
        dst[j] = active[i] ? x[i] : nullptr;
        viaFields = viaFields || (active[i] && !direct);
      }
      // X (+)= P e: corrections that went through fields are packed onto X (columns without one get zeros, or stay as they are when X already
      // holds the pre-smoothed solution); ... the quads are prolongated straight onto their columns
      if (viaFields || F.nuPre == 0) blockPackParity(*g.X, corr, g.n, F.nuPre > 0);   // (after pre-smoothing the idle and padding columns of X are zero already)
      for (int j = 0; j + 3 < g.n; j++) {
        const int i = g.first + j;
        if (!blockDirect || !quadAt[i]) continue;
        const ColorSpinorField *c4[4] = {st.xc[i], st.xc[i + 1], st.xc[i + 2], st.xc[i + 3]};
        transfer->P4Block(g.X->v, g.nrhs, j, true, c4);
        g_msStats[2]++;
      }
      if (F.nuPost > 0) {
        F.residual(g);
        F.mr(g, F.nuPost, nullptr, st.wantImage);
      }
      blockUnpackParity(dst, g.n, *g.X);
    }
    transfer->setSiteSubset(QUDA_FULL_SITE_SUBSET, QUDA_INVALID_PARITY);
  } else {
    transfer->setSiteSubset(QUDA_FULL_SITE_SUBSET, QUDA_INVALID_PARITY);
    for (int i = 0; i < nsrc; i++) if (active[i]) (*st.post[i])(*x[i], *b[i]);
  }
  blas::setGlobalReduction(true);
}

void MG::blockWantImage(int nsrc, bool on) { if (blockPrepare(nsrc)) blockState->wantImage = on; }

bool MG::blockApplyLast(std::vector<ColorSpinorField *> &out, const Dirac &pc, const std::vector<char> &active) {
  if (!blockState || !blockState->fine || !blockState->solutionOnBlocks) return false;
  BlockFineSmoother &F = *blockState->fine;
  static int off = -1;
  if (off < 0) { const char *e = getenv("QUDA_AMD_MULTISRC_BLOCK_OUTER"); off = (e && !atoi(e)) ? 1 : 0; }
  if (off) return false;
  const GaugeField *W = pc.Gauge();
  if (pc.getDiracType() != F.type || pc.getMatPCType() != F.matpcType || pc.Kappa() != F.kappa || pc.Mu() != F.mu || !W || !fineBlockSupported(*W, 8) || !fineBlockSupported(*W, 4)) return false;
  if (F.tmat[0] && !(pc.Clover() && pc.Clover()->precision == QUDA_SINGLE_PRECISION)) return false;
  if ((int)out.size() != blockState->nsrc) return false;
  for (size_t i = 0; i < out.size(); i++)
    if (active[i] && (out[i]->Precision() != QUDA_SINGLE_PRECISION || out[i]->SiteSubset() != QUDA_PARITY_SITE_SUBSET)) return false;
  // the post-smoother's last residual r = b - A x is at hand (MR keeps it with the solution): A x = b - r, one sweep instead of two stencil launches
  // (the same operator in exact arithmetic; in fp32 the recursion of two MR steps from an explicitly computed residual is as good as the product)
  static int viaResidual = -1;
  if (viaResidual < 0) { const char *e = getenv("QUDA_AMD_MULTISRC_IMAGE_FROM_RESIDUAL"); viaResidual = e ? atoi(e) : 1; }
  for (FineGroup &g : F.groups) {
    ColorSpinorField *dst[8];
    for (int j = 0; j < g.n; j++) dst[j] = active[g.first + j] ? out[g.first + j] : nullptr;
    if (blockState->residualOnBlocks && viaResidual) {   // (sloppy and preconditioner links hold the same matrices, to the rounding of their storage)
      blockblas::xmy(*g.B, *g.R);    // R <- b - r
      blockUnpackParity(dst, g.n, *g.R);
    } else {
      F.matpc(g, *g.AR, *g.X, 0, W);
      blockUnpackParity(dst, g.n, *g.AR);
    }
  }
  return true;
}

// Full-system outer solve: M x_i for the reconstructed solutions of the last cycleBlock, from the post-smoother's residuals on the block fields —
// (M x)_q = b_q, (M x)_p = b_p - A_pp r~ (symmetric preconditioning; see MG::imageOfLast): unpack r~ into scratch fields, local term, subtract.
bool MG::blockImageFull(std::vector<ColorSpinorField *> &out, std::vector<ColorSpinorField *> &b, const Dirac &full, const std::vector<char> &active) {
  if (!blockState || !blockState->fine || !blockState->residualOnBlocks) return false;
  static int viaResidual = -1;
  if (viaResidual < 0) { const char *e = getenv("QUDA_AMD_MULTISRC_IMAGE_FROM_RESIDUAL"); viaResidual = e ? atoi(e) : 1; }
  if (!viaResidual) return false;
  BlockFineSmoother &F = *blockState->fine;
  MGBlockState &st = *blockState;
  const Dirac &S = *mgp.matSmooth.Expose();
  const QudaDiracType at = full.getDiracType();
  const bool pair = (F.type == QUDA_WILSONPC_DIRAC && at == QUDA_WILSON_DIRAC) || (F.type == QUDA_TWISTED_MASSPC_DIRAC && at == QUDA_TWISTED_MASS_DIRAC) ||
                    (F.type == QUDA_TWISTED_CLOVERPC_DIRAC && at == QUDA_TWISTED_CLOVER_DIRAC);
  if (!pair || full.Kappa() != F.kappa || full.Mu() != F.mu || (int)out.size() != st.nsrc) return false;
  for (size_t i = 0; i < out.size(); i++)
    if (active[i] && (out[i]->Precision() != QUDA_SINGLE_PRECISION || out[i]->SiteSubset() != QUDA_FULL_SITE_SUBSET || b[i]->Precision() != QUDA_SINGLE_PRECISION)) return false;
  const int par = F.par;
  for (FineGroup &g : F.groups) {
    ColorSpinorField *dst[8];
    for (int j = 0; j < g.n; j++) {
      const int i = g.first + j;
      dst[j] = active[i] ? (par ? &st.r[i]->Odd() : &st.r[i]->Even()) : nullptr;
      if (dst[j]) dst[j]->twistFlavor = b[i]->twistFlavor;
    }
    blockUnpackParity(dst, g.n, *g.R);
    for (int j = 0; j < g.n; j++) {
      const int i = g.first + j;
      if (!active[i]) continue;
      S.localTermParity(*dst[j], *dst[j], par);     // symmetric preconditioning only (the block smoother's condition)
      blas::copy(*out[i], *b[i]);
      blas::mxpy(*dst[j], par ? out[i]->Odd() : out[i]->Even());
    }
  }
  return true;
}

// x_i = K b_i for all sources (full fields: Schur prepare, parity cycle, reconstruct, as MG::operator(); parity fields: the parity cycle)
bool MG::cycleBlock(std::vector<ColorSpinorField *> &x, std::vector<ColorSpinorField *> &b, const std::vector<char> &active) {
  const int nsrc = (int)x.size();
  if (!blockPrepare(nsrc)) return false;
  if (b[0]->SiteSubset() != QUDA_FULL_SITE_SUBSET) { cycleParityBlock(x, b, active, false); return true; }
  MGBlockState &st = *blockState;
  const Dirac &dirac = *mgp.matSmooth.Expose();
  const bool matpc = mgp.mg_global.coarse_grid_solution_type[0] == QUDA_MATPC_SOLUTION;
  std::vector<ColorSpinorField *> in(nsrc, nullptr), out(nsrc, nullptr);
  for (int i = 0; i < nsrc; i++) {
    if (!active[i]) continue;
    st.r[i]->twistFlavor = x[i]->twistFlavor = b[i]->twistFlavor;
    // (no copies: prepare() reads b and leaves the prepared source in the parity of x that reconstruct() fills last — MG::cycleUnfused)
    ColorSpinorField *pin = nullptr, *pout = nullptr;
    dirac.prepare(pin, pout, *x[i], *b[i], QUDA_MAT_SOLUTION);
    pin->twistFlavor = b[i]->twistFlavor;
    in[i] = pin; out[i] = pout;
  }
  for (int i = 0; i < nsrc; i++) if (!active[i]) { in[i] = st.btilde[i]; out[i] = st.btilde[i]; }
  cycleParityBlock(out, in, active, !matpc);
  st.solutionOnBlocks = false;   // the outer solver's direction is the reconstructed full field
  for (int i = 0; i < nsrc; i++) if (active[i]) dirac.reconstruct(*x[i], *b[i], QUDA_MAT_SOLUTION);
  return true;
}

// ================================================================================================
// lockstep GCR (reference lib/inv_gcr_quda.cpp:235-516 per source; host form solver.cpp GCR::operator())
// ================================================================================================
static ColorSpinorField *likeF(const ColorSpinorField &x, QudaPrecision prec, bool zeroed = true) {
  ColorSpinorParam p = x.param();
  p.location = QUDA_CUDA_FIELD_LOCATION;
  p.precision = prec;
  p.create = zeroed ? QUDA_ZERO_FIELD_CREATE : QUDA_NULL_FIELD_CREATE;
  return new ColorSpinorField(p);
}

struct BlockGcrResult { int iter = 0; std::vector<double> r2, b2; double secs = 0; bool blockCycle = false; };

static BlockGcrResult blockGCR(std::vector<ColorSpinorField *> &x, std::vector<ColorSpinorField *> &b, DiracMatrix &mat, DiracMatrix &matSloppy, MG *K, SolverParam &param, const Dirac *sloppyPC,
                               const Dirac *sloppyFull = nullptr) {
  const int ns = (int)x.size(), nK = param.Nkrylov;
  BlockGcrResult res;
  res.r2.assign(ns, 0.0); res.b2.assign(ns, 0.0);
  const bool mixed = param.precision_sloppy != x[0]->Precision();
  const QudaPrecision ps = param.precision_sloppy;
  std::vector<ColorSpinorField *> r(ns), y(ns), xS(ns), rS(ns);
  std::vector<std::vector<ColorSpinorField *>> p(ns), Ap(ns);
  for (int i = 0; i < ns; i++) {
    r[i] = likeF(*x[i], x[i]->Precision()); y[i] = likeF(*x[i], x[i]->Precision());
    xS[i] = mixed ? likeF(*x[i], ps) : x[i]; rS[i] = mixed ? likeF(*x[i], ps) : r[i];
    p[i].assign(nK, nullptr); Ap[i].assign(nK, nullptr);   // the Krylov space grows with the iteration count (12 sources x 40 fields otherwise)
    for (ColorSpinorField *f : {r[i], y[i], xS[i], rS[i]}) f->twistFlavor = b[i]->twistFlavor;
  }
  auto grow = [&](int k) {   // direction k of every source: written in full before it is read, so not cleared
    for (int i = 0; i < ns; i++) {
      if (p[i][k]) continue;
      p[i][k] = likeF(*x[i], ps, false); Ap[i][k] = likeF(*x[i], ps, false);
      p[i][k]->twistFlavor = Ap[i][k]->twistFlavor = b[i]->twistFlavor;
    }
  };
  std::vector<std::vector<Complex>> alpha(ns, std::vector<Complex>(nK));
  std::vector<std::vector<double>> gamma(ns, std::vector<double>(nK));
  std::vector<std::vector<std::vector<Complex>>> beta(ns, std::vector<std::vector<Complex>>(nK, std::vector<Complex>(nK)));
  std::vector<double> stop(ns), r2(ns), r2_old(ns);
  std::vector<char> open(ns, 1);
  blas::setGlobalReduction(param.global_reduction);
  const double t0 = nowSec();
  for (int i = 0; i < ns; i++) {
    res.b2[i] = blas::norm2(*b[i]);
    if (res.b2[i] == 0.0) errorQuda("Source %d has zero norm", i);
    blas::copy(*r[i], *b[i]);
    r2[i] = r2_old[i] = res.b2[i];
    blas::zero(*x[i]);
    if (mixed) { blas::zero(*xS[i]); blas::copy(*rS[i], *r[i]); }
    stop[i] = Solver::stopping(param.tol, res.b2[i], param.residual_type);
  }
  // test hook (tests/test_multisrc_gpu.py): "i:f" loosens the tolerance of source i by the factor f, so that it converges — and stops being
  // updated — iterations before the others; with one operator and one preconditioner the sources otherwise finish together
  if (const char *e = getenv("QUDA_AMD_MULTISRC_TEST_LOOSE")) {
    int i = -1;
    double f = 1.0;
    if (sscanf(e, "%d:%lf", &i, &f) == 2 && i >= 0 && i < ns && f >= 1.0) stop[i] *= f * f;
  }
  auto anyOpen = [&]() { for (int i = 0; i < ns; i++) if (open[i]) return true; return false; };
  int k = 0, total = 0;
  std::vector<ColorSpinorField *> pk(ns), rk(ns);
  while (anyOpen() && total < param.maxiter) {
    // p_k = K r for every open source: one block cycle
    grow(k);
    for (int i = 0; i < ns; i++) { pk[i] = p[i][k]; rk[i] = rS[i]; }
    bool done = false;
    if (K) {
      if (sloppyPC || sloppyFull) K->blockWantImage(ns, true);
      done = K->cycleBlock(pk, rk, open);
      res.blockCycle = res.blockCycle || done;
    }
    // A p_k for all sources while p_k is still on the smoother's block fields (even-odd outer solve on the smoother's operator)
    bool applied = false;
    if (done && sloppyPC) {
      std::vector<ColorSpinorField *> apk(ns);
      for (int i = 0; i < ns; i++) apk[i] = Ap[i][k];
      applied = K->blockApplyLast(apk, *sloppyPC, open);
    } else if (done && sloppyFull) {
      std::vector<ColorSpinorField *> apk(ns);
      for (int i = 0; i < ns; i++) apk[i] = Ap[i][k];
      applied = K->blockImageFull(apk, rk, *sloppyFull, open);
    }
    for (int i = 0; i < ns; i++) {
      if (!open[i]) continue;
      if (!done) { if (K) (*K)(*p[i][k], *rS[i]); else blas::copy(*p[i][k], *rS[i]); }
      blas::setGlobalReduction(param.global_reduction);
      if (!applied) matSloppy(*Ap[i][k], *p[i][k]);
      // orthogonalisation against the source's own directions: the blocked two-sweep form where it applies (solver.cpp), else the chain
      bool blocked = false;
      if (blas::multiSupported(*Ap[i][k], k)) {
        std::vector<Complex> bk(k > 0 ? k : 1);
        Complex apr; double apn;
        blas::multiDot(bk.data(), apr, apn, Ap[i], k, *Ap[i][k], *rS[i]);
        double g2 = apn;
        for (int j = 0; j < k; j++) g2 -= std::norm(bk[j]);
        if (apn == 0.0) errorQuda("GCR breakdown");
        if (g2 > 1e-2 * apn) {
          for (int j = 0; j < k; j++) { beta[i][j][k] = bk[j]; bk[j] = -bk[j]; }
          gamma[i][k] = sqrt(g2);
          alpha[i][k] = apr / gamma[i][k];
          double y2;
          blas::multiCaxpyResidual(r2[i], y2, bk.data(), Ap[i], k, 1.0 / gamma[i][k], *Ap[i][k], alpha[i][k], *rS[i]);
          blocked = true;
        }
      }
      if (!blocked) {
        for (int j = 0; j < k; j++) {
          beta[i][j][k] = blas::cDotProduct(*Ap[i][j], *Ap[i][k]);
          blas::caxpy(-beta[i][j][k], *Ap[i][j], *Ap[i][k]);
        }
        const double3_t Apr = blas::cDotProductNormA(*Ap[i][k], *rS[i]);
        gamma[i][k] = sqrt(Apr.z);
        if (gamma[i][k] == 0.0) errorQuda("GCR breakdown");
        alpha[i][k] = Complex(Apr.x, Apr.y) / gamma[i][k];
        r2[i] = blas::cabxpyAxNorm(1.0 / gamma[i][k], -alpha[i][k], *Ap[i][k], *rS[i]);
      }
    }
    k++; total++;
    // reliable update / restart for ALL sources when the Krylov space is full or every open source has reached its sloppy target
    bool allSloppy = true, anyDelta = false;
    for (int i = 0; i < ns; i++) if (open[i]) { allSloppy = allSloppy && r2[i] < stop[i]; anyDelta = anyDelta || sqrt(r2[i] / r2_old[i]) < param.delta; }
    if (k == nK || total == param.maxiter || allSloppy || anyDelta) {
      for (int i = 0; i < ns; i++) {
        if (!open[i]) continue;
        std::vector<Complex> delta(k);
        for (int a = k - 1; a >= 0; a--) {
          delta[a] = alpha[i][a];
          for (int j = a + 1; j < k; j++) delta[a] -= beta[i][a][j] * delta[j];
          delta[a] /= gamma[i][a];
        }
        if (blas::multiSupported(*xS[i], k)) blas::multiCaxpy(delta.data(), p[i], k, *xS[i]);
        else for (int a = 0; a < k; a++) blas::caxpy(delta[a], *p[i][a], *xS[i]);
        if (mixed) blas::copy(*x[i], *xS[i]);
        blas::xpy(*x[i], *y[i]);
        mat(*r[i], *y[i]);
        r2[i] = blas::xmyNorm(*b[i], *r[i]);
        if (r2[i] <= stop[i]) {
          open[i] = 0;
        } else {
          if (mixed) blas::copy(*rS[i], *r[i]);
          blas::zero(*xS[i]);
          if (!mixed) blas::zero(*x[i]);
        }
        r2_old[i] = r2[i];
      }
      k = 0;
    }
  }
  for (int i = 0; i < ns; i++) { blas::copy(*x[i], *y[i]); res.r2[i] = r2[i]; }
  res.iter = total;
  res.secs = nowSec() - t0;
  for (int i = 0; i < ns; i++) {
    delete r[i]; delete y[i];
    if (mixed) { delete xS[i]; delete rS[i]; }
    for (int kk = 0; kk < nK; kk++) { delete p[i][kk]; delete Ap[i][kk]; }   // nullptr beyond the directions used
  }
  blas::setGlobalReduction(true);
  return res;
}

MultiSrcSolve solveMultiSrcGCR(std::vector<ColorSpinorField *> &x, std::vector<ColorSpinorField *> &b, DiracMatrix &mat, DiracMatrix &matSloppy, MG *K, SolverParam &param, const Dirac *sloppyPC) {
  const BlockGcrResult r = blockGCR(x, b, mat, matSloppy, K, param, sloppyPC);
  g_msStats[3]++;
  MultiSrcSolve out;
  out.iter = r.iter; out.secs = r.secs; out.r2 = r.r2; out.b2 = r.b2;
  return out;
}

}  // namespace quda

using namespace quda;

extern "C" {

void qudaAmdMultiSrcStats(long long out[4]) { for (int i = 0; i < 4; i++) out[i] = g_msStats[i]; }

// reference include/quda.h:647 (lib/interface_quda.cpp:2546: "currently that code is just a copy of invertQuda and cannot work"): param->num_src
// sources _hp_b[i] -> solutions _hp_x[i], one operator, one preconditioner.  GCR (optionally MG-preconditioned) direct solves; iter / secs
// are those of the lockstep solve, true_res the worst source's.
void invertMultiSrcQuda(void **_hp_x, void **_hp_b, QudaInvertParam *param) {
  if (!gaugePrecise) errorQuda("Gauge field not allocated");
  if (param->tune == QUDA_TUNE_YES || param->tune == QUDA_TUNE_NO) setTuning(param->tune);
  if (!cloverPrecise && param->dslash_type == QUDA_TWISTED_CLOVER_DSLASH) errorQuda("Clover field not allocated");
  const int ns = param->num_src;
  if (ns < 1 || ns > kMaxBlockRhs) errorQuda("num_src = %d outside 1 .. %d", ns, kMaxBlockRhs);
  const bool pc_solution = param->solution_type == QUDA_MATPC_SOLUTION;
  const bool pc_solve = param->solve_type == QUDA_DIRECT_PC_SOLVE;
  if (param->solve_type != QUDA_DIRECT_SOLVE && param->solve_type != QUDA_DIRECT_PC_SOLVE) errorQuda("invertMultiSrcQuda: direct solves only");
  if (param->solution_type != QUDA_MAT_SOLUTION && param->solution_type != QUDA_MATPC_SOLUTION) errorQuda("invertMultiSrcQuda: MAT / MATPC solutions only");
  if (param->inv_type != QUDA_GCR_INVERTER) errorQuda("invertMultiSrcQuda: the lockstep solver is GCR (inv_type %d)", param->inv_type);
  if (pc_solution && !pc_solve) errorQuda("Preconditioned (PC) solution_type requires a PC solve_type");
  param->secs = 0; param->gflops = 0; param->iter = 0;

  DiracParam dp, dpSloppy, dpPre;
  setDiracParam(dp, param, pc_solve);
  setDiracSloppyParam(dpSloppy, param, pc_solve);
  setDiracPreParam(dpPre, param, pc_solve);
  Dirac *d = Dirac::create(dp), *dSloppy = Dirac::create(dpSloppy), *dPre = Dirac::create(dpPre);
  const LatticeGeom &geom = residentGeom();
  std::vector<ColorSpinorField *> b(ns), x(ns), in(ns), out(ns);
  std::vector<double> nb(ns);
  for (int i = 0; i < ns; i++) {
    ColorSpinorParam cpuParam(_hp_b[i], *param, geom.X, pc_solution);
    ColorSpinorField h_b(cpuParam);
    ColorSpinorParam cp = deviceSpinorParam(param->cuda_prec, pc_solution ? QUDA_PARITY_SITE_SUBSET : QUDA_FULL_SITE_SUBSET, param->twist_flavor);
    cp.create = QUDA_ZERO_FIELD_CREATE;
    b[i] = new ColorSpinorField(cp); x[i] = new ColorSpinorField(cp);
    *b[i] = h_b;
    nb[i] = blas::norm2(*b[i]);
    if (nb[i] == 0.0) errorQuda("Source %d has zero norm", i);
    if (param->solver_normalization == QUDA_SOURCE_NORMALIZATION) blas::ax(1.0 / sqrt(nb[i]), *b[i]);
    massRescale(*b[i], *param);
    d->prepare(in[i], out[i], *x[i], *b[i], param->solution_type);
  }
  {
    DiracM m(*d), mSloppy(*dSloppy);
    SolverParam sp(*param);
    MG *K = nullptr;
    if (param->inv_type_precondition == QUDA_MG_INVERTER) {
      if (!param->preconditioner) errorQuda("GCR with multigrid preconditioner: param.preconditioner is NULL");
      K = static_cast<multigrid_solver *>(param->preconditioner)->mg;
    } else if (param->inv_type_precondition != QUDA_INVALID_INVERTER) {
      errorQuda("invertMultiSrcQuda: preconditioner %d not supported (none or QUDA_MG_INVERTER)", param->inv_type_precondition);
    }
    const BlockGcrResult res = blockGCR(out, in, m, mSloppy, K, sp, pc_solve ? dSloppy : nullptr, pc_solve ? nullptr : dSloppy);
    g_msStats[3]++;
    param->iter = res.iter;
    param->secs = res.secs;
    double worst = 0;
    for (int i = 0; i < ns; i++) worst = std::max(worst, sqrt(res.r2[i] / res.b2[i]));
    param->true_res = worst;
    // the hierarchy keeps its multi-source work space (solvers, residuals, block fields of the coarse levels) for the next call with the same
    // number of sources — a propagator is 12 of these calls' worth; released with the hierarchy or when num_src changes
  }
  for (int i = 0; i < ns; i++) {
    d->reconstruct(*x[i], *b[i], param->solution_type);
    if (param->solver_normalization == QUDA_SOURCE_NORMALIZATION) blas::ax(sqrt(nb[i]), *x[i]);
    ColorSpinorParam cpuParam(_hp_x[i], *param, geom.X, pc_solution);
    ColorSpinorField h_x(cpuParam);
    h_x = *x[i];
    delete b[i]; delete x[i];
  }
  delete d; delete dSloppy; delete dPre;
}

}
