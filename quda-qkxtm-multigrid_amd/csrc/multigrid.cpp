// multigrid.cpp — MG setup and cycle (see multigrid.h).
#include "multigrid.h"
#include "block.h"
#include "coarse_cycle.h"
#include "p2p.h"

#include <cmath>
#include <sys/time.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "interface_internal.h"

namespace quda {

static double now() {
  timeval t;
  gettimeofday(&t, nullptr);
  return t.tv_sec + 1e-6 * t.tv_usec;
}

MGParam::MGParam(QudaMultigridParam &g, std::vector<ColorSpinorField *> &B_, DiracMatrix &res, DiracMatrix &smooth, int level_, QudaTwistFlavorType flavor)
    : SolverParam(*g.invert_param), mg_global(g), level(level_), Nlevel(g.n_level), spinBlockSize(g.spin_block_size[level_]), Nvec(g.n_vec[level_]),
      B(B_), nu_pre(g.nu_pre[level_]), nu_post(g.nu_post[level_]), smoother_tol(g.smoother_tol[level_]), cycle_type(g.cycle_type[level_]),
      smoother(g.smoother[level_]), matResidual(res), matSmooth(smooth), fineFlavor(flavor) {
  for (int d = 0; d < 4; d++) geoBlockSize[d] = g.geo_block_size[level_][d];
  omega = g.omega[level_];
  global_reduction = g.global_reduction[level_] != QUDA_BOOLEAN_NO;
  precision = precision_sloppy = precision_precondition = QUDA_SINGLE_PRECISION;
  inv_type_precondition = QUDA_INVALID_INVERTER;
  preconditioner = nullptr;
  use_init_guess = QUDA_USE_INIT_GUESS_NO;
  iter = 0; gflops = 0; secs = 0;
}

// QUDA_AMD_MG_PROFILE=1: synchronised wall-clock per cycle stage and level, printed when the hierarchy is destroyed
// (the reference keeps a TimeProfile per level, lib/multigrid.cpp:16-17)
static double g_mgProf[QUDA_MAX_MG_LEVEL][6];
static long g_mgCalls[QUDA_MAX_MG_LEVEL];
static int mgProfiling() {
  static int on = -1;
  if (on < 0) { const char *e = getenv("QUDA_AMD_MG_PROFILE"); on = e ? atoi(e) : 0; }
  return on;
}
void multigridSetHalfStorage(multigrid_solver &mgs, bool on);

static ColorSpinorField *likeField(const ColorSpinorField &x) {
  ColorSpinorParam p = x.param();
  p.create = QUDA_ZERO_FIELD_CREATE;
  return new ColorSpinorField(p);
}

MG::MG(MGParam &p)
    : Solver(p), mgp(p), transfer(nullptr), presmoother(nullptr), postsmoother(nullptr), coarse_solver(nullptr), param_presmooth(nullptr),
      param_postsmooth(nullptr), param_coarse_solver(nullptr), coarse(nullptr), param_coarse(nullptr), r(nullptr), r_coarse(nullptr), x_coarse(nullptr),
      b_tilde(nullptr), diracCoarse(nullptr), diracCoarseSmoother(nullptr), matCoarse(nullptr), matCoarseSmoother(nullptr), pcSmooth(false),
      ownCoarseSolver(false) {
  if (p.level >= QUDA_MAX_MG_LEVEL) errorQuda("Level=%d is greater than limit of multigrid recursion depth", p.level + 1);
  if (getVerbosity() >= QUDA_SUMMARIZE) printfQuda("MG level %d: creating level %d of %d\n", p.level + 1, p.level + 1, p.Nlevel);
  const bool coarsest = p.level == p.Nlevel - 1;
  pcSmooth = p.mg_global.smoother_solve_type[p.level] == QUDA_DIRECT_PC_SOLVE;
  if (pcSmooth != p.matSmooth.isPC()) errorQuda("smoother_solve_type[%d] and the smoother operator disagree about even-odd preconditioning", p.level);

  if (!coarsest) {
    if (p.vectorsPreset && p.level == 0) {
      // refinement pass: the caller has put the vectors into B
    } else if (p.mg_global.compute_null_vector == QUDA_COMPUTE_NULL_VECTOR_YES && (p.mg_global.generate_all_levels == QUDA_BOOLEAN_YES || p.level == 0)) {
      const double t0 = now();
      generateNullVectors(p.B);
      HIP_CHECK(hipStreamSynchronize(computeStream()));
      if (getVerbosity() >= QUDA_SUMMARIZE || mgProfiling()) printfQuda("MG level %d: %d null vectors generated in %.3f s\n", p.level + 1, p.Nvec, now() - t0);
      saveVectors(p.B);
    } else if (p.mg_global.compute_null_vector == QUDA_COMPUTE_NULL_VECTOR_NO && (p.mg_global.generate_all_levels == QUDA_BOOLEAN_YES || p.level == 0)) {
      loadVectors(p.B);   // reference :25-32: a previously saved null space instead of a fresh setup
    }
  }

  // smoothers (reference :40-85)
  param_presmooth = new SolverParam(p);
  param_presmooth->inv_type = p.smoother;
  param_presmooth->inv_type_precondition = QUDA_INVALID_INVERTER;
  param_presmooth->is_preconditioner = true;
  param_presmooth->preserve_source = QUDA_PRESERVE_SOURCE_YES;
  param_presmooth->use_init_guess = QUDA_USE_INIT_GUESS_NO;
  param_presmooth->maxiter = p.nu_pre;
  param_presmooth->Nkrylov = 4;
  param_presmooth->tol = p.smoother_tol;
  param_presmooth->global_reduction = p.global_reduction;
  param_presmooth->compute_true_res = false;
  param_presmooth->precision = param_presmooth->precision_sloppy = param_presmooth->precision_precondition = QUDA_SINGLE_PRECISION;
  if (coarsest) {
    // coarsest-grid solver: GCR(20) to smoother_tol (reference :63-70)
    param_presmooth->inv_type = QUDA_GCR_INVERTER;
    param_presmooth->Nkrylov = 20;
    param_presmooth->maxiter = 1000;
    param_presmooth->delta = 1e-8;
    param_presmooth->pipeline = 1;
    param_presmooth->global_reduction = true;
  }
  presmoother = Solver::create(*param_presmooth, p.matSmooth, p.matSmooth, p.matSmooth);
  if (!coarsest) {
    param_postsmooth = new SolverParam(*param_presmooth);
    param_postsmooth->use_init_guess = QUDA_USE_INIT_GUESS_YES;
    param_postsmooth->maxiter = p.nu_post;
    postsmoother = Solver::create(*param_postsmooth, p.matSmooth, p.matSmooth, p.matSmooth);
  }

  r = likeField(*p.B[0]);
  r->twistFlavor = p.fineFlavor;
  if (pcSmooth) {
    ColorSpinorParam bp = r->Even().param();
    bp.create = QUDA_ZERO_FIELD_CREATE;
    b_tilde = new ColorSpinorField(bp);
    b_tilde->twistFlavor = p.fineFlavor;
  }
  if (!coarsest) {
    const double tT = now();
    transfer = new Transfer(p.B, p.Nvec, p.geoBlockSize, p.spinBlockSize);
    if (getVerbosity() >= QUDA_SUMMARIZE || mgProfiling()) printfQuda("MG level %d: transfer (fill + block Gram-Schmidt) in %.3f s\n", p.level + 1, now() - tT);
    for (int d = 0; d < 4; d++) p.mg_global.geo_block_size[p.level][d] = p.geoBlockSize[d];
    r_coarse = transfer->createCoarseField();
    x_coarse = transfer->createCoarseField();

    // Galerkin coarse operator (reference :150-190)
    DiracParam dp;
    dp.type = QUDA_COARSE_DIRAC;
    dp.transfer = transfer;
    dp.dirac = p.matResidual.Expose();
    dp.kappa = 1.0;  // the hopping normalisation is folded into the coarse links (coarse.h)
    dp.dagger = QUDA_DAG_NO;
    dp.matpcType = p.mg_global.invert_param->matpc_type;
    dp.twistFlavor = p.fineFlavor;
    const double t0 = now();
    diracCoarse = new DiracCoarse(dp);
    matCoarse = new DiracM(*diracCoarse);
    if (getVerbosity() >= QUDA_SUMMARIZE || mgProfiling()) printfQuda("MG level %d: coarse operator %d^2 x 9 per site on %d x %d x %d x %d built in %.3f s\n", p.level + 1, 2 * p.Nvec, transfer->Xc[0], transfer->Xc[1], transfer->Xc[2], transfer->Xc[3], now() - t0);

    // coarse null vectors: restricted fine ones unless every level generates its own (reference :196-208)
    const int nVecCoarse = std::max(p.Nvec, p.level + 1 < p.Nlevel ? p.mg_global.n_vec[p.level + 1] : p.Nvec);
    B_coarse.resize(nVecCoarse, nullptr);
    for (int i = 0; i < nVecCoarse; i++) B_coarse[i] = transfer->createCoarseField();
    if (p.mg_global.generate_all_levels != QUDA_BOOLEAN_YES)
      for (int i = 0; i < p.Nvec; i++) transfer->R(*B_coarse[i], *p.B[i]);

    // smoother operator of the next level: the coarse operator itself or its even-odd preconditioned form (reference :184-190)
    if (p.mg_global.smoother_solve_type[p.level + 1] == QUDA_DIRECT_PC_SOLVE) {
      DiracParam dps = dp;
      dps.type = QUDA_COARSEPC_DIRAC;
      if (dps.matpcType != QUDA_MATPC_EVEN_EVEN && dps.matpcType != QUDA_MATPC_ODD_ODD) dps.matpcType = QUDA_MATPC_EVEN_EVEN;
      diracCoarseSmoother = new DiracCoarsePC(*diracCoarse, dps);
      matCoarseSmoother = new DiracM(*diracCoarseSmoother);
    }
    param_coarse = new MGParam(p.mg_global, B_coarse, *matCoarse, matCoarseSmoother ? *matCoarseSmoother : *matCoarse, p.level + 1, QUDA_TWIST_NO);
    param_coarse->delta = 1e-20;
    coarse = new MG(*param_coarse);

    if (p.cycle_type == QUDA_MG_CYCLE_VCYCLE || p.level == p.Nlevel - 2) {
      coarse_solver = coarse;
    } else if (p.cycle_type == QUDA_MG_CYCLE_RECURSIVE) {
      // K-cycle: GCR(10), at most 11 iterations, preconditioned by the coarse V-cycle (reference :225-250)
      param_coarse_solver = new SolverParam(*param_coarse);
      param_coarse_solver->inv_type = QUDA_GCR_INVERTER;
      param_coarse_solver->inv_type_precondition = QUDA_MG_INVERTER;
      param_coarse_solver->is_preconditioner = true;
      param_coarse_solver->preserve_source = QUDA_PRESERVE_SOURCE_YES;
      param_coarse_solver->use_init_guess = QUDA_USE_INIT_GUESS_NO;
      param_coarse_solver->maxiter = 11;
      param_coarse_solver->Nkrylov = 10;
      param_coarse_solver->tol = p.mg_global.smoother_tol[p.level + 1];
      param_coarse_solver->global_reduction = true;
      param_coarse_solver->compute_true_res = false;
      param_coarse_solver->delta = 1e-8;
      param_coarse_solver->pipeline = 1;
      param_coarse_solver->precision = param_coarse_solver->precision_sloppy = param_coarse_solver->precision_precondition = QUDA_SINGLE_PRECISION;
      coarse_solver = new GCR(*matCoarse, *coarse, *matCoarse, *matCoarse, *param_coarse_solver);
      ownCoarseSolver = true;
    } else {
      errorQuda("Multigrid cycle type %d not supported", p.cycle_type);
    }
  }
  if (getVerbosity() >= QUDA_SUMMARIZE) printfQuda("MG level %d: setup completed\n", p.level + 1);
}

void MG::dropFusedCycle() {
  coarseCycleDestroy(fused);
  fused = nullptr; fusedTried = false; fusedVerified = false;
  if (coarse) coarse->dropFusedCycle();
}

MG::~MG() {
  blockRelease();
  coarseCycleDestroy(fused);
  if (ownCoarseSolver) delete coarse_solver;
  delete param_coarse_solver;
  delete coarse;
  delete param_coarse;
  for (ColorSpinorField *f : B_coarse) delete f;
  delete matCoarseSmoother;
  delete diracCoarseSmoother;
  delete matCoarse;
  delete diracCoarse;
  delete transfer;
  delete postsmoother;
  delete presmoother;
  delete param_presmooth;
  delete param_postsmooth;
  delete r; delete r_coarse; delete x_coarse; delete b_tilde;
}

unsigned long long MG::flops() const {
  unsigned long long f = 0;
  if (coarse) f += coarse->flops();
  if (param_presmooth) { f += (unsigned long long)(param_presmooth->gflops * 1e9); param_presmooth->gflops = 0; }
  if (param_postsmooth) { f += (unsigned long long)(param_postsmooth->gflops * 1e9); param_postsmooth->gflops = 0; }
  if (transfer) f += transfer->flops();
  return f;
}

// reference generateNullVectors :693-779
// After the lockstep solves.  The reference orthonormalises every finished vector against its predecessors
// (lib/multigrid.cpp:757-771) before Transfer orthonormalises them again block by block.  For the hierarchy that global pass is a
// no-op in exact arithmetic: it replaces V by V T with T upper triangular with a positive diagonal, and the block-local QR of
// V_b T has the same Q as that of V_b (R_b T is again upper triangular with a positive diagonal, QR is unique).  It costs
// Nvec (Nvec - 1) field passes (0.27 s of a 3.2 s setup at 48^3 x 96), so the block path only NORMALISES the vectors, which keeps
// the Gram matrices of the block CholeskyQR (transfer.hip) well scaled; QUDA_AMD_NULL_ORTHO=gs restores the global pass.
static void orthonormaliseNullVectors(std::vector<ColorSpinorField *> &B, int Nvec) {
  static int gs = -1;
  if (gs < 0) { const char *e = getenv("QUDA_AMD_NULL_ORTHO"); gs = e && !strcmp(e, "gs") ? 1 : 0; }
  for (int i = 0; i < Nvec; i++) {
    ColorSpinorField &x = *B[i];
    if (gs)
      for (int j = 0; j < i; j++) {
        const Complex alpha = blas::cDotProduct(*B[j], x);
        blas::caxpy(-alpha, *B[j], x);
      }
    const double nrm2 = blas::norm2(x);
    if (nrm2 > 1e-16) blas::ax(1.0 / sqrt(nrm2), x);
    else errorQuda("Cannot orthogonalize %d vector", i);
  }
}

void MG::generateNullVectors(std::vector<ColorSpinorField *> &B) {
  if (getVerbosity() >= QUDA_SUMMARIZE) printfQuda("MG level %d: generating %d null vectors (BiCGstab, maxiter %d, tol %g)\n", mgp.level + 1, mgp.Nvec, mgp.mg_global.setup_maxiter, mgp.mg_global.setup_tol);
  SolverParam sp(mgp);
  sp.maxiter = mgp.mg_global.setup_maxiter;
  sp.tol = mgp.mg_global.setup_tol;
  sp.use_init_guess = QUDA_USE_INIT_GUESS_YES;
  sp.delta = 1e-7;
  sp.inv_type = QUDA_BICGSTAB_INVERTER;
  sp.inv_type_precondition = QUDA_INVALID_INVERTER;
  sp.residual_type = QUDA_L2_RELATIVE_RESIDUAL;
  sp.compute_null_vector = QUDA_COMPUTE_NULL_VECTOR_YES;
  sp.is_preconditioner = true;
  sp.global_reduction = true;
  sp.precision = sp.precision_sloppy = sp.precision_precondition = QUDA_SINGLE_PRECISION;
  // Coarse levels: the Nvec solves run in lockstep on one block field, so every operator application reads the dense coarse
  // links once for all of them and runs on the matrix cores (block.h; reference: the multi-source 5th dimension of its coarse
  // kernel, lib/dslash_coarse.cu:294-333).  Same recurrences per vector as the loop below; the orthonormalisation that the loop
  // interleaves with the solves only touches finished vectors, so it is done afterwards.
  const DiracCoarse *dc = dynamic_cast<const DiracCoarse *>(mgp.matResidual.Expose());
  if (dc && dc->getDiracType() == QUDA_COARSE_DIRAC && blockCoarseSupported(dc->Links(), mgp.Nvec) && !coarseHalfStorage()) {
    const double t0 = now();
    for (int i = 0; i < mgp.Nvec; i++) { B[i]->twistFlavor = mgp.fineFlavor; spinorRandom(*B[i], 0x5eedULL + 7919ULL * (mgp.level * 131 + i)); }
    BlockField X(dc->Links().nSites, dc->Links().n, mgp.Nvec, blockGhost(dc->Links().Xc, false).nGhost);
    std::vector<ColorSpinorField *> Bv(B.begin(), B.begin() + mgp.Nvec);
    blockPack(X, Bv);
    struct Ctx { const CoarseGauge *G; long applies; } ctx = {&dc->Links(), 0};
    int iters[kMaxBlockRhs];
    const int kmax = blockBiCGstabNull(X, [](BlockField &out, BlockField &in, void *c) { Ctx *x = (Ctx *)c; applyCoarseBlock(out, in, *x->G); x->applies++; }, &ctx, sp.tol, sp.maxiter, iters);
    blockUnpack(Bv, X);
    orthonormaliseNullVectors(B, mgp.Nvec);
    nullVectorMethod = 2; nullVectorIterations = kmax;
    if (getVerbosity() >= QUDA_SUMMARIZE || mgProfiling()) {
      int imin = iters[0], imax = iters[0];
      for (int i = 1; i < mgp.Nvec; i++) { imin = iters[i] < imin ? iters[i] : imin; imax = iters[i] > imax ? iters[i] : imax; }
      printfQuda("MG level %d: %d null vectors by block BiCGstab on the MFMA coarse operator: %d lockstep iterations (per vector %d..%d), %ld block applications, %.3f s\n",
                 mgp.level + 1, mgp.Nvec, kmax, imin, imax, ctx.applies, now() - t0);
    }
    return;
  }
  // Fine level: the same lockstep solve on the multi-right-hand-side stencil (dslash.h applyFineBlockM), 8 vectors per link load —
  // batches of 8 keep a site's panel small enough for the neighbour re-use of the XCD-slab block order
  {
    const Dirac *df = mgp.matResidual.Expose();
    const QudaDiracType ty = df ? df->getDiracType() : QUDA_INVALID_DIRAC;
    static int nbEnv = -1;
    if (nbEnv < 0) { const char *e = getenv("QUDA_AMD_BLOCK_FINE_NRHS"); nbEnv = e ? atoi(e) : 8; if (nbEnv != 8 && nbEnv != 16 && nbEnv != 24 && nbEnv != 32) nbEnv = 8; }
    const int nb = nbEnv;
    const bool tmc = ty == QUDA_TWISTED_CLOVER_DIRAC && df->Clover() && df->Clover()->precision == QUDA_SINGLE_PRECISION;
    if (df && (ty == QUDA_TWISTED_MASS_DIRAC || ty == QUDA_WILSON_DIRAC || tmc) && B[0]->Nspin() == 4 && B[0]->Precision() == QUDA_SINGLE_PRECISION &&
        mgp.Nvec % nb == 0 && df->Gauge() && fineBlockSupported(*df->Gauge(), nb)) {
      const double t0 = now();
      const double a = ty == QUDA_WILSON_DIRAC ? 0.0 : 2.0 * df->Kappa() * (double)mgp.fineFlavor * df->Mu();
      // The solves run on the EVEN-ODD PRECONDITIONED system: M x = 0 with x = (x_e, x_o) is  Mhat x_e = 0,  x_o = kappa A^-1 D_oe x_e
      // (Mhat = 1 - kappa^2 A^-1 D_eo A^-1 D_oe, A = 1 + i a g5: reference DiracTwistedMassPC::M / reconstruct with b = 0,
      // lib/dirac_twisted_mass.cpp:340-393, :526-548) — the same null space, but the BiCGstab vectors are half as long (its 21
      // field passes per iteration, not the links, bound the stage) and the Schur complement is better conditioned.
      // Twisted clover: the same with A = clover + i a g5 (reference DiracTwistedCloverPC::M / reconstruct,
      // lib/dirac_twisted_clover.cpp:296-330, :400-421); A^-1 resp. A enter the stencil's epilogue as dense 6 x 6 site matrices
      // per chirality (cloverTwistDense), built once per setup.
      // QUDA_AMD_NULL_FULL=1 keeps the solves on the full operator as the reference has them.
      static int fullOp = -1;
      if (fullOp < 0) { const char *e = getenv("QUDA_AMD_NULL_FULL"); fullOp = e ? atoi(e) : 0; }
      const double kappa = df->Kappa(), binv = 1.0 / (1.0 + a * a);
      const int Vh = B[0]->VolumeCB();
      // grid-decomposed lattice: every operator input carries the faces of the neighbour ranks behind its local panels (block.h BlockGhost)
      const int nGhostPar = blockGhost(df->Gauge()->geom.X, true).nGhost;
      float *tmat[2] = {nullptr, nullptr};
      const size_t tmatBytes = (size_t)Vh * 144 * sizeof(float);
      if (tmc)
        for (int p = 0; p < 2; p++) {
          tmat[p] = (float *)poolDeviceMalloc(tmatBytes);
          cloverTwistDense(tmat[p], *df->Clover(), p, a, !fullOp);
        }
      struct Ctx { const GaugeField *U; double kappa, a, binv; BlockField *tmp; long applies; const float *tmat[2]; } ctx = {df->Gauge(), kappa, a, binv, nullptr, 0, {tmat[0], tmat[1]}};
      int kmaxAll = 0, imin = 1 << 30, imax = 0;
      for (int i = 0; i < mgp.Nvec; i++) { B[i]->twistFlavor = mgp.fineFlavor; spinorRandom(*B[i], 0x5eedULL + 7919ULL * (mgp.level * 131 + i)); }
      double tPack = 0, tSolve = 0;
      for (int g0 = 0; g0 < mgp.Nvec; g0 += nb) {
        std::vector<ColorSpinorField *> Bv(B.begin() + g0, B.begin() + g0 + nb);
        double tp = now();
        int iters[kMaxBlockRhs], kmax;
        if (fullOp) {
          BlockField X(B[0]->Volume(), 12, nb, 2 * nGhostPar);
          blockPack(X, Bv);
          if (mgProfiling()) { HIP_CHECK(hipStreamSynchronize(computeStream())); tPack += now() - tp; tp = now(); }
          kmax = blockBiCGstabNull(X, [](BlockField &out, BlockField &in, void *c) {
            Ctx *x = (Ctx *)c; applyFineBlockM(out.v, in.v, in.nrhs, *x->U, x->kappa, x->a, x->tmat[0] ? x->tmat : nullptr); x->applies++; },
                                   &ctx, sp.tol, sp.maxiter, iters);
          if (mgProfiling()) { HIP_CHECK(hipStreamSynchronize(computeStream())); tSolve += now() - tp; tp = now(); }
          blockUnpack(Bv, X);
        } else {
          BlockField Xe(Vh, 12, nb, nGhostPar), Xo(Vh, 12, nb, nGhostPar);
          blockPack(Xe, Bv, 0);
          ctx.tmp = &Xo;
          if (mgProfiling()) { HIP_CHECK(hipStreamSynchronize(computeStream())); tPack += now() - tp; tp = now(); }
          auto ghostOf = [](BlockField &f) { return f.nGhost ? f.v + f.elems() : nullptr; };
          kmax = blockBiCGstabNull(Xe, [](BlockField &out, BlockField &in, void *c) {
            Ctx *x = (Ctx *)c;
            float2 *gin = in.nGhost ? in.v + in.elems() : nullptr, *gtmp = x->tmp->nGhost ? x->tmp->v + x->tmp->elems() : nullptr;
            // tmp_o = A^-1 D_oe in_e ;  out_e = in_e - kappa^2 A^-1 D_eo tmp_o          (twisted mass: A^-1 = binv (1 - i a g5))
            if (x->tmat[0]) {
              applyFineBlockParity(x->tmp->v, nullptr, in.v, in.nrhs, *x->U, 1, 0.0, 0.0, 1.0, 0.0, x->tmat[1], 1, gin);
              applyFineBlockParity(out.v, in.v, x->tmp->v, in.nrhs, *x->U, 0, 1.0, 0.0, -x->kappa * x->kappa, 0.0, x->tmat[0], 1, gtmp);
            } else {
              applyFineBlockParity(x->tmp->v, nullptr, in.v, in.nrhs, *x->U, 1, 0.0, 0.0, x->binv, -x->a, nullptr, 0, gin);
              applyFineBlockParity(out.v, in.v, x->tmp->v, in.nrhs, *x->U, 0, 1.0, 0.0, -x->kappa * x->kappa * x->binv, -x->a, nullptr, 0, gtmp);
            }
            x->applies++;
          }, &ctx, sp.tol, sp.maxiter, iters,
          // the same operator with the inner products of the BiCGstab half step taken in the second launch's epilogue (dslash.h FineBlockDots)
          !fineBlockDotsSupported(nb) ? (BlockMatVecDots) nullptr : [](BlockField &out, BlockField &in, void *c, const BlockField &r0, int mode, double *sums) {
            Ctx *x = (Ctx *)c;
            float2 *gin = in.nGhost ? in.v + in.elems() : nullptr, *gtmp = x->tmp->nGhost ? x->tmp->v + x->tmp->elems() : nullptr;
            const FineBlockDots dots = {r0.v, mode};
            if (x->tmat[0]) {
              applyFineBlockParity(x->tmp->v, nullptr, in.v, in.nrhs, *x->U, 1, 0.0, 0.0, 1.0, 0.0, x->tmat[1], 1, gin);
              applyFineBlockParity(out.v, in.v, x->tmp->v, in.nrhs, *x->U, 0, 1.0, 0.0, -x->kappa * x->kappa, 0.0, x->tmat[0], 1, gtmp, &dots);
            } else {
              applyFineBlockParity(x->tmp->v, nullptr, in.v, in.nrhs, *x->U, 1, 0.0, 0.0, x->binv, -x->a, nullptr, 0, gin);
              applyFineBlockParity(out.v, in.v, x->tmp->v, in.nrhs, *x->U, 0, 1.0, 0.0, -x->kappa * x->kappa * x->binv, -x->a, nullptr, 0, gtmp, &dots);
            }
            fineBlockDotsFinish(sums, in.nrhs, mode);
            x->applies++;
          });
          // x_o = kappa A^-1 D_oe x_e
          if (tmc) applyFineBlockParity(Xo.v, nullptr, Xe.v, nb, *df->Gauge(), 1, 0.0, 0.0, kappa, 0.0, tmat[1], 1, ghostOf(Xe));
          else applyFineBlockParity(Xo.v, nullptr, Xe.v, nb, *df->Gauge(), 1, 0.0, 0.0, kappa * binv, -a, nullptr, 0, ghostOf(Xe));
          if (mgProfiling()) { HIP_CHECK(hipStreamSynchronize(computeStream())); tSolve += now() - tp; tp = now(); }
          blockUnpack(Bv, Xe, 0);
          blockUnpack(Bv, Xo, 1);
        }
        if (mgProfiling()) { HIP_CHECK(hipStreamSynchronize(computeStream())); tPack += now() - tp; }
        kmaxAll = kmax > kmaxAll ? kmax : kmaxAll;
        for (int i = 0; i < nb; i++) { imin = iters[i] < imin ? iters[i] : imin; imax = iters[i] > imax ? iters[i] : imax; }
      }
      if (tmc) {
        HIP_CHECK(hipStreamSynchronize(computeStream()));
        for (int p = 0; p < 2; p++) poolDeviceFree(tmat[p], tmatBytes);
      }
      orthonormaliseNullVectors(B, mgp.Nvec);
      nullVectorMethod = 1; nullVectorIterations = kmaxAll;
      if (mgProfiling()) printfQuda("MG level %d: block null-vector stage: pack/unpack + field allocation %.3f s, lockstep solves %.3f s, orthonormalisation %.3f s\n", mgp.level + 1, tPack, tSolve, now() - t0 - tPack - tSolve);
      if (getVerbosity() >= QUDA_SUMMARIZE || mgProfiling())
        printfQuda("MG level %d: %d null vectors by block BiCGstab on the %d-right-hand-side %s stencil: up to %d lockstep iterations (per vector %d..%d), %ld block applications, %.3f s\n",
                   mgp.level + 1, mgp.Nvec, nb, tmc ? "twisted-clover" : "twisted-mass / Wilson", kmaxAll, imin, imax, ctx.applies, now() - t0);
      return;
    }
  }
  ColorSpinorField *b = likeField(*B[0]);
  b->twistFlavor = mgp.fineFlavor;
  const QudaVerbosity v0 = getVerbosity();
  for (int i = 0; i < mgp.Nvec; i++) {
    ColorSpinorField &x = *B[i];
    x.twistFlavor = mgp.fineFlavor;
    spinorRandom(x, 0x5eedULL + 7919ULL * (mgp.level * 131 + i));
    blas::zero(*b);
    Solver *solve = Solver::create(sp, mgp.matResidual, mgp.matResidual, mgp.matResidual);
    (*solve)(x, *b);
    delete solve;
    // global orthonormalisation against the previous vectors
    for (int j = 0; j < i; j++) {
      const Complex alpha = blas::cDotProduct(*B[j], x);
      blas::caxpy(-alpha, *B[j], x);
    }
    const double nrm2 = blas::norm2(x);
    if (nrm2 > 1e-16) blas::ax(1.0 / sqrt(nrm2), x);
    else errorQuda("Cannot orthogonalize %d vector", i);
  }
  (void)v0;
  delete b;
}

// ---- null-vector persistence (reference MG::saveVectors / loadVectors, lib/multigrid.cpp:607-691: "<file>_level_<l>", all
// Nvec vectors of a level in one file).  The reference hands the vectors to QIO (lib/qio_field.cpp write_spinor_field); files are
// written in that container — SciDAC records inside LIME, one field record with datacount = Nvec, global lexicographic site order,
// big-endian fp32, restated in csrc/lime_io.cpp — by all ranks into ONE file, as QIO_PARALLEL / QIO_SINGLEFILE does.  Vectors go
// to the file in the host order the reference keeps them in: even-odd sites, (spin, colour, re/im), DeGrand-Rossi basis.
// Files of this library's first two rounds (64-byte header "QAMDNV01/02" + raw vectors, one file per rank) are still read. ----
struct NullVecHeader { char magic[8]; int X[4]; int nSpin, nColor, Nvec, precision; int grid[4]; int rank, order; };
constexpr int kNullVecOrder = 0x01020304;
static std::string nullVecFile(const char *base, int level, bool perRank) {
  std::string f(base);
  f += "_level_" + std::to_string(level);
  if (perRank && commGrid().size > 1) f += ".rank" + std::to_string(commGrid().rank);
  return f;
}
static ColorSpinorParam hostParamLike(const ColorSpinorField &dev, void *ptr) {
  ColorSpinorParam p = dev.param();
  p.location = QUDA_CPU_FIELD_LOCATION; p.precision = QUDA_SINGLE_PRECISION; p.fieldOrder = QUDA_SPACE_SPIN_COLOR_FIELD_ORDER;
  p.gammaBasis = QUDA_DEGRAND_ROSSI_GAMMA_BASIS; p.create = QUDA_REFERENCE_FIELD_CREATE; p.pad = 0; p.v = ptr;
  return p;
}
void MG::saveVectors(std::vector<ColorSpinorField *> &B) {
  if (!mgp.mg_global.vec_outfile[0]) return;
  const std::string f = nullVecFile(mgp.mg_global.vec_outfile, mgp.level, false);
  if (getVerbosity() >= QUDA_SUMMARIZE) printfQuda("Start saving %zu vectors to %s\n", B.size(), f.c_str());
  const size_t n = (size_t)B[0]->Volume() * B[0]->Nspin() * B[0]->Ncolor() * 2;
  std::vector<std::vector<float>> host(B.size());
  std::vector<const float *> ptrs(B.size());
  for (size_t i = 0; i < B.size(); i++) {
    host[i].resize(n);
    ColorSpinorField h(hostParamLike(*B[i], host[i].data()));
    h = *B[i];
    ptrs[i] = host[i].data();
  }
  int X[4];
  for (int d = 0; d < 4; d++) X[d] = B[0]->X(d);
  scidacWriteSpinors(f.c_str(), ptrs, X, B[0]->Nspin(), B[0]->Ncolor());
  if (getVerbosity() >= QUDA_SUMMARIZE) printfQuda("Done saving vectors\n");
}
void MG::loadVectors(std::vector<ColorSpinorField *> &B) {
  if (!mgp.mg_global.vec_infile[0]) errorQuda("compute_null_vector = NO needs vec_infile (no null-space file defined)");
  std::string f = nullVecFile(mgp.mg_global.vec_infile, mgp.level, false);
  const size_t n = (size_t)B[0]->Volume() * B[0]->Nspin() * B[0]->Ncolor() * 2;
  if (scidacIsContainer(f.c_str())) {
    if (getVerbosity() >= QUDA_SUMMARIZE) printfQuda("Start loading %zu vectors from %s\n", B.size(), f.c_str());
    std::vector<std::vector<float>> host(B.size());
    std::vector<float *> ptrs(B.size());
    for (size_t i = 0; i < B.size(); i++) { host[i].resize(n); ptrs[i] = host[i].data(); }
    int X[4];
    for (int d = 0; d < 4; d++) X[d] = B[0]->X(d);
    scidacReadSpinors(f.c_str(), ptrs, X, B[0]->Nspin(), B[0]->Ncolor());
    for (size_t i = 0; i < B.size(); i++) {
      ColorSpinorField h(hostParamLike(*B[i], host[i].data()));
      *B[i] = h;
      B[i]->twistFlavor = mgp.fineFlavor;
    }
    if (getVerbosity() >= QUDA_SUMMARIZE) printfQuda("Done loading vectors\n");
    return;
  }
  // the private format of rounds 1 and 2: one file per rank
  f = nullVecFile(mgp.mg_global.vec_infile, mgp.level, true);
  if (getVerbosity() >= QUDA_SUMMARIZE) printfQuda("Start loading %zu vectors from %s\n", B.size(), f.c_str());
  FILE *fp = fopen(f.c_str(), "rb");
  if (!fp) errorQuda("cannot open %s", f.c_str());
  NullVecHeader h;
  if (fread(&h, sizeof(h), 1, fp) != 1 || (memcmp(h.magic, "QAMDNV01", 8) && memcmp(h.magic, "QAMDNV02", 8)))
    errorQuda("%s is neither a SciDAC / LIME vector file nor a null-vector file of this library", f.c_str());
  if (!memcmp(h.magic, "QAMDNV02", 8) && h.order != kNullVecOrder) errorQuda("%s was written on a machine of the other byte order", f.c_str());
  for (int d = 0; d < 4; d++)
    if (h.X[d] != B[0]->X(d) || h.grid[d] != commGrid().dims[d]) errorQuda("%s was written for lattice %d %d %d %d on grid %d %d %d %d", f.c_str(), h.X[0], h.X[1], h.X[2], h.X[3], h.grid[0], h.grid[1], h.grid[2], h.grid[3]);
  if (h.nSpin != B[0]->Nspin() || h.nColor != B[0]->Ncolor() || h.Nvec < (int)B.size() || h.rank != commGrid().rank) errorQuda("%s does not match this level (nSpin %d nColor %d Nvec %d rank %d)", f.c_str(), h.nSpin, h.nColor, h.Nvec, h.rank);
  if (h.Nvec > (int)B.size()) warningQuda("%s holds %d vectors, this level uses the first %zu", f.c_str(), h.Nvec, B.size());
  std::vector<float> buf(n);
  for (size_t i = 0; i < B.size(); i++) {
    if (fread(buf.data(), sizeof(float), n, fp) != n) errorQuda("short read on %s", f.c_str());
    ColorSpinorField host(hostParamLike(*B[i], buf.data()));
    *B[i] = host;
    B[i]->twistFlavor = mgp.fineFlavor;
  }
  fclose(fp);
}

// reference :488-604 (outer and inner solution type QUDA_MAT_SOLUTION; smoother full or even-odd preconditioned)
struct StageTimer {
  int level, stage; double t0; bool on;
  StageTimer(int l, int s) : level(l), stage(s), t0(0), on(mgProfiling() != 0) {
    if (on) { HIP_CHECK(hipStreamSynchronize(computeStream())); t0 = now(); }
  }
  ~StageTimer() {
    if (!on) return;
    const double t1 = now();
    HIP_CHECK(hipStreamSynchronize(computeStream()));
    const double t2 = now();
    g_mgProf[level][stage] += t2 - t0;
    if (mgProfiling() == 2 && t2 - t0 > 2e-3) printfQuda("MG stage: level %d stage %d took %.3f ms (host %.3f ms + drain %.3f ms)\n", level, stage, 1e3 * (t2 - t0), 1e3 * (t1 - t0), 1e3 * (t2 - t1));
  }
};
void mgProfilePrint() {
  if (!mgProfiling()) return;
  static const char *names[6] = {"pre-smooth", "residual", "restrict", "coarse solve", "prolong", "post-smooth"};
  for (int l = 0; l < QUDA_MAX_MG_LEVEL; l++) {
    if (!g_mgCalls[l]) continue;
    printfQuda("MG profile level %d (%ld cycles):", l, g_mgCalls[l]);
    for (int s = 0; s < 6; s++) printfQuda("  %s %.3f ms", names[s], 1e3 * g_mgProf[l][s]);
    printfQuda("\n");
    g_mgCalls[l] = 0;
    for (int s = 0; s < 6; s++) g_mgProf[l][s] = 0;
  }
}

// Outer even-odd preconditioned solve (the way the QKXTM drivers run: solve_type = QUDA_DIRECT_PC_SOLVE, reference
// lib/interface_quda.cpp:6041, :6427-6445): b and x live on one parity and the cycle preconditions Mhat x = b.  As in the
// reference (lib/multigrid.cpp:492-560, outer_solution_type = QUDA_MATPC_SOLUTION) this needs the even-odd preconditioned
// smoother; the single-parity residual is injected into the coarse grid (Transfer::setSiteSubset), the coarse problem is
// the usual full coarse operator, and only the solved parity of the prolongated correction is added.
// fullResidual: the caller solves the FULL system behind an even-odd preconditioned smoother (prepare -> this -> reconstruct).  With the
// reconstructed odd sites x_o = A^-1 (b_o + kappa D x_e) the full residual b - M x vanishes on the odd sites and equals A_ee r_hat on the
// even ones for the symmetric preconditioning (r_hat = b_tilde - Mhat x_e; plain r_hat for the asymmetric one) — so the "restrict the
// full residual" cycle (coarse_grid_solution_type = QUDA_MAT_SOLUTION, reference lib/multigrid.cpp:540-560) is THIS cycle with the
// site-diagonal term applied to r_hat: no reconstruction before the coarse-grid correction, no application of the full operator, and
// R and P touch one parity (half of V with the parity-major aggregates of transfer.hip).  The smoother hands over the residual it
// ended with (Solver::lastResidual) instead of the cycle applying Mhat once more.
void MG::cycleParity(ColorSpinorField &x, ColorSpinorField &b, bool fullResidual) {
  if (mgp.level != 0) errorQuda("a single-parity source can only enter the multigrid cycle on the finest level");
  if (!pcSmooth) errorQuda("For this coarse grid solution type, a preconditioned smoother is required");
  const Dirac &dirac = *mgp.matSmooth.Expose();
  const bool odd = dirac.getMatPCType() == QUDA_MATPC_ODD_ODD || dirac.getMatPCType() == QUDA_MATPC_ODD_ODD_ASYMMETRIC;
  const bool symmetric = dirac.getMatPCType() == QUDA_MATPC_EVEN_EVEN || dirac.getMatPCType() == QUDA_MATPC_ODD_ODD;
  x.twistFlavor = b.twistFlavor;
  if (mgp.level == mgp.Nlevel - 1) { (*presmoother)(x, b); return; }
  ColorSpinorField &rp = odd ? r->Odd() : r->Even();
  r->twistFlavor = rp.twistFlavor = b.twistFlavor;
  g_mgCalls[mgp.level]++;
  { StageTimer t(mgp.level, 0); (*presmoother)(x, b); }
  const ColorSpinorField *rin = &rp;
  {
    StageTimer t(mgp.level, 1);
    static int reuse = -1;
    if (reuse < 0) { const char *e = getenv("QUDA_AMD_MG_SMOOTHER_RESIDUAL"); reuse = e ? atoi(e) : 1; }
    const ColorSpinorField *res = reuse && mgp.nu_pre > 0 ? presmoother->lastResidual() : nullptr;
    if (res && res->Precision() == QUDA_SINGLE_PRECISION && !(fullResidual && symmetric)) {
      rin = res;                          // restricted as it is
    } else if (res) {
      if (fullResidual && symmetric && res->Precision() == QUDA_SINGLE_PRECISION) dirac.localTermParity(rp, *res, odd ? 1 : 0);
      else { blas::copy(rp, *res); if (fullResidual && symmetric) dirac.localTermParity(rp, rp, odd ? 1 : 0); }   // 16-bit smoother fields
    } else {
      mgp.matSmooth(rp, x);
      blas::axpby(1.0, b, -1.0, rp);   // preconditioned residual rhat = b - Mhat x
      if (fullResidual && symmetric) dirac.localTermParity(rp, rp, odd ? 1 : 0);
    }
  }
  {
    StageTimer t(mgp.level, 2);
    transfer->setSiteSubset(QUDA_PARITY_SITE_SUBSET, odd ? QUDA_ODD_PARITY : QUDA_EVEN_PARITY);
    transfer->R(*r_coarse, *rin);
    if (ownCoarseSolver) blas::zero(*x_coarse);   // a V-cycle below (coarse_solver IS the next level's cycle) defines x in full from a zero guess of its own
  }
  { StageTimer t(mgp.level, 3); (*coarse_solver)(*x_coarse, *r_coarse); }
  {
    StageTimer t(mgp.level, 4);
    transfer->P(rp, *x_coarse);
    transfer->setSiteSubset(QUDA_FULL_SITE_SUBSET, QUDA_INVALID_PARITY);
    blas::xpy(rp, x);
  }
  { StageTimer t(mgp.level, 5); (*postsmoother)(x, b); }
  if (mgp.nu_post > 0) lastParityCycle = odd ? 1 : 0;
  blas::setGlobalReduction(true);
}

// A x for the x = K b of the last call, from the post-smoother's residual r~ = b~ - Mhat x_p of the even-odd system it worked on (MR keeps it):
//   A = Mhat (even-odd outer solve on the smoother's operator):   A x = b - r~
//   A = M (full system; x reconstructed from x_p):  (M x)_q = b_q exactly, (M x)_p = b_p - A_pp r~ (symmetric preconditioning) resp. b_p - r~ (asymmetric)
// — the relation MG::cycleParity uses the other way round to restrict the full residual behind an even-odd smoother.  One or two sweeps instead of
// an application of A: 2 of the 12 stencil launches of an outer GCR iteration.
bool MG::imageOfLast(ColorSpinorField &Ax, const ColorSpinorField &b, const DiracMatrix &A) {
  static int on = -1;
  if (on < 0) { const char *e = getenv("QUDA_AMD_MG_IMAGE_FROM_RESIDUAL"); on = e ? atoi(e) : 1; }
  if (!on || lastParityCycle < 0 || !A.isM() || !pcSmooth) return false;
  const Dirac *Ad = A.Expose(), *S = mgp.matSmooth.Expose();
  const ColorSpinorField *res = postsmoother->lastResidual();
  if (!Ad || !S || !res || res->Precision() != Ax.Precision() || b.Precision() != Ax.Precision() || Ax.V() == b.V()) return false;
  if (Ad->Kappa() != S->Kappa() || Ad->Mu() != S->Mu()) return false;
  const QudaDiracType st = S->getDiracType(), at = Ad->getDiracType();
  const QudaMatPCType mt = S->getMatPCType();
  const bool symmetric = mt == QUDA_MATPC_EVEN_EVEN || mt == QUDA_MATPC_ODD_ODD;
  const int par = lastParityCycle;
  if (b.SiteSubset() == QUDA_PARITY_SITE_SUBSET) {
    if (at != st || Ad->getMatPCType() != mt || res->VolumeCB() != b.VolumeCB()) return false;
    blas::xmyz(b, *res, Ax);   // (Ax is only written: it may hold anything, NaNs of recycled memory included)
    return true;
  }
  // (coarse levels: the K-cycle's GCR around this level's cycle asks for M_c x — the same relation with the coarse local term X_pp)
  const bool pair = (st == QUDA_WILSONPC_DIRAC && at == QUDA_WILSON_DIRAC) || (st == QUDA_TWISTED_MASSPC_DIRAC && at == QUDA_TWISTED_MASS_DIRAC) ||
                    (st == QUDA_TWISTED_CLOVERPC_DIRAC && at == QUDA_TWISTED_CLOVER_DIRAC) || (st == QUDA_COARSEPC_DIRAC && at == QUDA_COARSE_DIRAC);
  if (!pair || res->VolumeCB() != b.VolumeCB() || res->Ncolor() != b.Ncolor()) return false;
  ColorSpinorField &target = par ? Ax.Odd() : Ax.Even();
  const ColorSpinorField &bp = par ? b.Odd() : b.Even();
  blas::copy(par ? Ax.Even() : Ax.Odd(), par ? b.Even() : b.Odd());
  if (symmetric) {
    target.twistFlavor = b.twistFlavor;
    S->localTermParity(target, *res, par);
    blas::xmyz(bp, target, target);
  } else {
    blas::xmyz(bp, *res, target);
  }
  return true;
}

void MG::operator()(ColorSpinorField &x, ColorSpinorField &b) {
  lastParityCycle = -1;
  if (b.SiteSubset() != QUDA_FULL_SITE_SUBSET) { cycleParity(x, b, false); return; }
  if (mgp.level >= 1 && coarseCycleEnabled() && !mgProfiling()) {
    // the whole cycle from this level down in one launch (coarse_cycle.h); the first use is checked against the kernel-per-operation path
    // on every rank, and any disagreement (or a wait that ran out) sends all ranks back to that path for good
    if (!fusedTried) { fusedTried = true; fused = coarseCycleCreate(*this); }
    if (fused && !fusedVerified) {
      // (the fused cycle reads the fp32 coarse links whatever the storage switch says: the reference cycle of this check does the same)
      const bool half = coarseHalfStorage();
      if (half) setCoarseHalfStorage(false);
      cycleUnfused(x, b);
      if (half) setCoarseHalfStorage(true);
      ColorSpinorField ref(x);
      double fail = coarseCycleApply(fused, x, b) ? 0.0 : 1.0;
      lastParityCycle = -1;   // x is the fused kernel's now: the smoother residual of the reference cycle above does not belong to it
      HIP_CHECK(hipStreamSynchronize(computeStream()));
      if (p2pTakeError()) fail = 1.0;
      if (fail == 0.0) {
        const bool wasGlobal = blas::globalReduction();
        blas::setGlobalReduction(false);
        const double n2 = blas::norm2(ref), d2 = blas::xmyNorm(x, ref);   // ref <- x - ref
        blas::setGlobalReduction(wasGlobal);
        if (!(d2 <= 1e-6 * n2)) fail = 1.0;
        if (getVerbosity() >= QUDA_VERBOSE || getenv("QUDA_AMD_MG_FUSED_VERBOSE")) printfQuda("MG level %d: fused coarse cycle against the kernel-per-operation path: |dx|^2 / |x|^2 = %e\n", mgp.level + 1, n2 > 0 ? d2 / n2 : d2);
      }
      if (getenv("QUDA_AMD_MG_FUSED_VERIFY_FAIL")) fail = 1.0;
      comm_allreduce(&fail, 1);
      if (fail != 0.0) {
        if (commGrid().rank == 0) warningQuda("MG level %d: the fused coarse cycle disagrees with the kernel-per-operation path on its first use: staying with the latter", mgp.level + 1);
        coarseCycleDestroy(fused);
        fused = nullptr;
        cycleUnfused(x, b);
      } else {
        fusedVerified = true;
      }
      blas::setGlobalReduction(true);
      return;
    }
    if (fused && coarseCycleApply(fused, x, b)) { blas::setGlobalReduction(true); return; }
  }
  cycleUnfused(x, b);
}

void MG::cycleUnfused(ColorSpinorField &x, ColorSpinorField &b) {
  const Dirac &dirac = *mgp.matSmooth.Expose();
  static int parityRoute = -1;
  if (parityRoute < 0) { const char *e = getenv("QUDA_AMD_MG_PARITY_CYCLE"); parityRoute = e ? atoi(e) : 1; }
  const bool matpc = mgp.mg_global.coarse_grid_solution_type[0] == QUDA_MATPC_SOLUTION;
  if (mgp.level == 0 && mgp.level < mgp.Nlevel - 1 && pcSmooth && (matpc || parityRoute)) {
    // full-system outer solve through the parity cycle: Schur-prepare the source, run the cycle on the solved parity, reconstruct the
    // other one.  coarse_grid_solution_type QUDA_MATPC_SOLUTION: single-parity injection of the preconditioned residual (reference outer
    // QUDA_MAT_SOLUTION / inner QUDA_MATPC_SOLUTION, lib/multigrid.cpp:513-560); QUDA_MAT_SOLUTION: the full residual, which behind an
    // even-odd smoother lives on that parity as well (cycleParity, fullResidual) — the same cycle as the full-field code below in exact
    // arithmetic, without its reconstruct + full-operator application per cycle and with half the transfer traffic
    // (QUDA_AMD_MG_PARITY_CYCLE=0 keeps the full-field form)
    ColorSpinorField *out = nullptr, *in = nullptr;
    r->twistFlavor = x.twistFlavor = b.twistFlavor;
    // no copy of the source and none of the prepared source: prepare() of the even-odd operators of this level reads b only and builds its source in
    // the parity of x that reconstruct() fills last — the cycle between them works on the other parity of x and on r (three device copies per
    // cycle less: 3 % of a 48^3 x 96 solve)
    dirac.prepare(in, out, x, b, QUDA_MAT_SOLUTION);
    in->twistFlavor = b.twistFlavor;
    cycleParity(*out, *in, !matpc);
    dirac.reconstruct(x, b, QUDA_MAT_SOLUTION);
    return;
  }
  ColorSpinorField *out = nullptr, *in = nullptr;
  r->twistFlavor = b.twistFlavor;
  x.twistFlavor = b.twistFlavor;
  if (mgp.level < mgp.Nlevel - 1) {
    blas::copy(*r, b);  // the source is copied: prepare() of a preconditioned smoother builds its source from it
    dirac.prepare(in, out, x, *r, QUDA_MAT_SOLUTION);
    if (pcSmooth) { b_tilde->twistFlavor = b.twistFlavor; blas::copy(*b_tilde, *in); }  // keep the prepared source for the post-smoother
    g_mgCalls[mgp.level]++;
    { StageTimer t(mgp.level, 0); (*presmoother)(*out, *in); }
    {
      StageTimer t(mgp.level, 1);
      // behind an even-odd smoother the full residual of the reconstructed x lives on the solved parity: r_p = A_pp r~ (r~ = the residual MR ended with),
      // r_q = 0 — MG::imageOfLast; one local-term application instead of the full operator (12^3 x 24 coarse level of a 48^3 x 96 solve: 0.09 against
      // 1.3 ms per cycle).  QUDA_AMD_MG_SMOOTHER_RESIDUAL=0: by the operator, as the reference
      static int reuse = -1;
      if (reuse < 0) { const char *e = getenv("QUDA_AMD_MG_SMOOTHER_RESIDUAL"); reuse = e ? atoi(e) : 1; }
      const QudaMatPCType mt = dirac.getMatPCType();
      const ColorSpinorField *res = (reuse && pcSmooth && mgp.nu_pre > 0 && (mt == QUDA_MATPC_EVEN_EVEN || mt == QUDA_MATPC_ODD_ODD)) ? presmoother->lastResidual() : nullptr;
      if (res && res->Precision() == r->Precision() && res->VolumeCB() == r->VolumeCB()) {
        const int par = mt == QUDA_MATPC_ODD_ODD ? 1 : 0;
        // (no reconstruction of the other parity of x here: only the operator residual below needs it; the post-smoother works on the solved parity and
        // the reconstruction at the end rebuilds the other one from it)
        dirac.localTermParity(par ? r->Odd() : r->Even(), *res, par);
        blas::zero(par ? r->Even() : r->Odd());
      } else {
        dirac.reconstruct(x, b, QUDA_MAT_SOLUTION);
        mgp.matResidual(*r, x);
        blas::axpby(1.0, b, -1.0, *r);  // r = b - A x   (full residual: coarse_grid_solution_type = MAT)
      }
    }
    {
      StageTimer t(mgp.level, 2);
      transfer->R(*r_coarse, *r);
      if (ownCoarseSolver) blas::zero(*x_coarse);
    }
    {
      StageTimer t(mgp.level, 3);
      (*coarse_solver)(*x_coarse, *r_coarse);
    }
    {
      StageTimer t(mgp.level, 4);
      transfer->P(*r, *x_coarse);     // repurpose residual storage
      blas::xpy(*r, x);
    }
    StageTimer tpost(mgp.level, 5);
    if (pcSmooth) {
      in = b_tilde;
      out = dirac.getMatPCType() == QUDA_MATPC_ODD_ODD || dirac.getMatPCType() == QUDA_MATPC_ODD_ODD_ASYMMETRIC ? &x.Odd() : &x.Even();
    } else {
      blas::copy(*r, b);
      in = r;
      out = &x;
    }
    (*postsmoother)(*out, *in);
    dirac.reconstruct(x, b, QUDA_MAT_SOLUTION);
    if (pcSmooth && mgp.nu_post > 0) lastParityCycle = dirac.getMatPCType() == QUDA_MATPC_ODD_ODD || dirac.getMatPCType() == QUDA_MATPC_ODD_ODD_ASYMMETRIC ? 1 : 0;
  } else {
    // coarsest-grid solve
    g_mgCalls[mgp.level]++;
    StageTimer t(mgp.level, 0);
    blas::copy(*r, b);
    dirac.prepare(in, out, x, *r, QUDA_MAT_SOLUTION);
    (*presmoother)(*out, *in);
    dirac.reconstruct(x, b, QUDA_MAT_SOLUTION);
  }
  blas::setGlobalReduction(true);
}

void MG::setSmootherSloppy(DiracMatrix *sloppy) {
  if (mgp.level != 0 || mgp.level == mgp.Nlevel - 1) return;
  delete presmoother; delete postsmoother;
  const QudaPrecision sp = sloppy ? QUDA_HALF_PRECISION : QUDA_SINGLE_PRECISION;
  param_presmooth->precision_sloppy = param_postsmooth->precision_sloppy = sp;
  DiracMatrix &ms = sloppy ? *sloppy : mgp.matSmooth;
  presmoother = Solver::create(*param_presmooth, mgp.matSmooth, ms, ms);
  postsmoother = Solver::create(*param_postsmooth, mgp.matSmooth, ms, ms);
}

void MG::makeHalfMirrors() {
  if (transfer) transfer->makeHalf();
  if (diracCoarse) diracCoarse->Links().makeHalf();
  if (diracCoarseSmoother) diracCoarseSmoother->HatLinks().makeHalf();
  if (coarse) coarse->makeHalfMirrors();
}

void MG::verify(double dev[3]) {
  dev[0] = dev[1] = dev[2] = 0.0;
  if (mgp.level >= mgp.Nlevel - 1) return;
  ColorSpinorField *tmp1 = likeField(*mgp.B[0]), *tmp2 = likeField(*mgp.B[0]);
  ColorSpinorField *c1 = transfer->createCoarseField(), *c2 = transfer->createCoarseField(), *c3 = transfer->createCoarseField();
  tmp1->twistFlavor = tmp2->twistFlavor = mgp.fineFlavor;
  // (1) v_k - P P^dag v_k = 0 for the vectors the transfer was built from
  for (int i = 0; i < mgp.Nvec; i++) {
    transfer->R(*c1, *mgp.B[i]);
    transfer->P(*tmp2, *c1);
    const double n = blas::norm2(*mgp.B[i]);
    blas::copy(*tmp1, *mgp.B[i]);
    const double d = blas::xmyNorm(*tmp2, *tmp1);
    if (n > 0) dev[0] = std::max(dev[0], sqrt(d / n));
  }
  // (2) eta_c - P^dag P eta_c = 0
  spinorRandom(*c1, 0xabcdefULL + mgp.level);
  transfer->P(*tmp1, *c1);
  transfer->R(*c2, *tmp1);
  {
    const double n = blas::norm2(*c1);
    blas::copy(*c3, *c1);
    dev[1] = sqrt(blas::xmyNorm(*c2, *c3) / n);
  }
  // (3) emulated R D P eta_c against the native coarse operator
  transfer->P(*tmp1, *c1);
  mgp.matResidual(*tmp2, *tmp1);
  transfer->R(*c2, *tmp2);
  (*matCoarse)(*c3, *c1);
  {
    const double n = blas::norm2(*c3);
    dev[2] = sqrt(blas::xmyNorm(*c2, *c3) / n);
  }
  delete tmp1; delete tmp2; delete c1; delete c2; delete c3;
  if (getVerbosity() >= QUDA_SUMMARIZE)
    printfQuda("MG level %d verify: |(1-PP^dag)v|/|v| = %e  |(1-P^dag P)eta|/|eta| = %e  |RDP eta - Dc eta|/|Dc eta| = %e\n", mgp.level + 1, dev[0], dev[1], dev[2]);
  if (coarse) {
    double dc[3];
    coarse->verify(dc);
    for (int k = 0; k < 3; k++) dev[k] = std::max(dev[k], dc[k]);
  }
}

// ================================================================================================
// reference multigrid_solver ctor, lib/interface_quda.cpp:2161-2255
multigrid_solver::multigrid_solver(QudaMultigridParam &mg_param)
    : d(nullptr), m(nullptr), dSmooth(nullptr), mSmooth(nullptr), gaugeHalf(nullptr), cloverHalf(nullptr), dSmoothHalf(nullptr), mSmoothHalf(nullptr), mgParam(nullptr), mg(nullptr) {
  QudaInvertParam *param = mg_param.invert_param;
  if (!param) errorQuda("QudaMultigridParam.invert_param is NULL");
  if (mg_param.n_level < 2 || mg_param.n_level > QUDA_MAX_MG_LEVEL) errorQuda("Requested MG levels %d outside 2..%d", mg_param.n_level, QUDA_MAX_MG_LEVEL);
  for (int i = 0; i < mg_param.n_level; i++) {
    if (mg_param.smoother_solve_type[i] != QUDA_DIRECT_SOLVE && mg_param.smoother_solve_type[i] != QUDA_DIRECT_PC_SOLVE)
      errorQuda("Unsupported smoother solve type %d on level %d", mg_param.smoother_solve_type[i], i);
    // QUDA_MAT_SOLUTION: the full residual is restricted; QUDA_MATPC_SOLUTION (what the harness sets for an outer even-odd
    // solve, tests/multigrid_invert_test.cpp:246-252): single-parity injection, needs the even-odd smoother (reference
    // lib/multigrid.cpp:503-504).  Honoured on the finest level; the coarse levels always restrict their full residual.
    if (mg_param.coarse_grid_solution_type[i] != QUDA_MAT_SOLUTION && mg_param.coarse_grid_solution_type[i] != QUDA_MATPC_SOLUTION)
      errorQuda("coarse_grid_solution_type[%d] = %d not supported (QUDA_MAT_SOLUTION, QUDA_MATPC_SOLUTION)", i, mg_param.coarse_grid_solution_type[i]);
    if (mg_param.coarse_grid_solution_type[i] == QUDA_MATPC_SOLUTION && mg_param.smoother_solve_type[i] != QUDA_DIRECT_PC_SOLVE)
      errorQuda("For this coarse grid solution type, a preconditioned smoother is required");
  }
  if (param->solve_type != QUDA_DIRECT_SOLVE) errorQuda("Outer MG solver can only use QUDA_DIRECT_SOLVE at present");
  GaugeField *g = gaugePrecondition ? gaugePrecondition : (gaugeSloppy ? gaugeSloppy : gaugePrecise);
  if (!g) errorQuda("Gauge field not allocated");
  if (g->precision != QUDA_SINGLE_PRECISION) errorQuda("the multigrid hierarchy runs in fp32: load the gauge field with cuda_prec_precondition = QUDA_SINGLE_PRECISION (got %d)", g->precision);
  mg_param.secs = 0; mg_param.gflops = 0;
  const double t0 = now();

  // QKXTM: kappa / mu rescaled for the P,R setup only (reference :2196-2211)
  const double orig_mu = param->mu, orig_kappa = param->kappa, orig_mass = param->mass;
  param->kappa *= mg_param.delta_kappaPR;
  param->mu *= mg_param.delta_muPR;
  param->mass = 0.5 / param->kappa - 4.0;
  DiracParam dp;
  setDiracPreParam(dp, param, false);
  if (param->dslash_type == QUDA_TWISTED_CLOVER_DSLASH && dp.clover && dp.clover->precision != QUDA_SINGLE_PRECISION) errorQuda("multigrid needs an fp32 precondition clover field");
  d = Dirac::create(dp);
  m = new DiracM(*d);
  if (mg_param.smoother_solve_type[0] == QUDA_DIRECT_PC_SOLVE) {
    DiracParam dps;
    setDiracPreParam(dps, param, true);
    dSmooth = Dirac::create(dps);
    mSmooth = new DiracM(*dSmooth);
  }
  param->kappa = orig_kappa; param->mu = orig_mu; param->mass = orig_mass;

  ColorSpinorParam cp = deviceSpinorParam(QUDA_SINGLE_PRECISION, QUDA_FULL_SITE_SUBSET, param->twist_flavor);
  cp.create = QUDA_ZERO_FIELD_CREATE;
  B.resize(mg_param.n_vec[0]);
  for (int i = 0; i < mg_param.n_vec[0]; i++) B[i] = new ColorSpinorField(cp);
  mgParam = new MGParam(mg_param, B, *m, mSmooth ? *mSmooth : *m, 0, param->twist_flavor);
  mg = new MG(*mgParam);
  if (mg_param.run_verify == QUDA_BOOLEAN_YES) { double dev[3]; mg->verify(dev); }
  mg_param_copy = mg_param;
  inv_param_copy = *param;
  { const char *e = getenv("QUDA_AMD_MG_REFINE"); if (e && atoi(e) > 0) refine(atoi(e), 1); }
  { const char *e = getenv("QUDA_AMD_MG_HALF"); if (e && atoi(e)) multigridSetHalfStorage(*this, true); }
  mg_param.secs = now() - t0;
  // the block work fields of the lockstep solves (7 x 4 GB at 48^3 x 96) stay parked in the pool for the next hierarchy (the down-flavour
  // one of the QKXTM drivers is built right after this one); QUDA_AMD_POOL_KEEP=0 hands everything of 256 MB and more back now
  { const char *e = getenv("QUDA_AMD_POOL_KEEP"); if (e && !atoi(e)) { HIP_CHECK(hipStreamSynchronize(computeStream())); poolDeviceFlush((size_t)256 << 20); } }
}

void multigrid_solver::refine(int passes, int cycles) {
  if (passes <= 0) return;
  const double t0 = now();
  // half-precision storage is a property of the hierarchy objects that are rebuilt below: remember it and put it back (a refined
  // hierarchy silently lost its fp16 mirrors and 16-bit smoother while the process-wide switch stayed on)
  const bool hadHalf = coarseHalfStorage() && mSmoothHalf != nullptr;
  if (hadHalf) multigridSetHalfStorage(*this, false);
  ColorSpinorField *y = likeField(*B[0]);
  const QudaTwistFlavorType flavor = B[0]->TwistFlavor();
  y->twistFlavor = flavor;
  mg_param_copy.invert_param = &inv_param_copy;
  for (int pass = 0; pass < passes; pass++) {
    for (size_t i = 0; i < B.size(); i++) {
      for (int c = 0; c < (cycles > 0 ? cycles : 1); c++) {
        (*mg)(*y, *B[i]);          // y = K v: one cycle of the current hierarchy
        blas::copy(*B[i], *y);
        const double n2 = blas::norm2(*B[i]);
        if (!(n2 > 0.0) || !std::isfinite(n2)) errorQuda("set-up refinement: null vector %zu collapsed (|K v|^2 = %e)", i, n2);
        blas::ax(1.0 / sqrt(n2), *B[i]);
      }
    }
    HIP_CHECK(hipStreamSynchronize(computeStream()));
    delete mg;
    delete mgParam;
    mgParam = new MGParam(mg_param_copy, B, *m, mSmooth ? *mSmooth : *m, 0, flavor);
    mgParam->vectorsPreset = true;
    mg = new MG(*mgParam);
    if (getVerbosity() >= QUDA_SUMMARIZE || mgProfiling()) printfQuda("MG set-up refinement pass %d of %d done (%.3f s so far)\n", pass + 1, passes, now() - t0);
  }
  delete y;
  if (hadHalf) multigridSetHalfStorage(*this, true);
  mg_param_copy.secs += now() - t0;
}

void multigridSetHalfStorage(multigrid_solver &mgs, bool on) {
  if (on) {
    mgs.mg->makeHalfMirrors();
    // 16-bit level-0 smoother (the work fields and the operator of the MR iterations; source and result stay fp32) with the even-odd
    // smoother; twisted clover (round 3): a 16-bit copy of the clover term is made too, its (A^2 + mu2)^-1 recomputed on the device
    QudaInvertParam *param = &mgs.inv_param_copy;
    if (mgs.mSmooth && !mgs.mSmoothHalf) {
      GaugeField *g = gaugePrecondition ? gaugePrecondition : (gaugeSloppy ? gaugeSloppy : gaugePrecise);
      mgs.gaugeHalf = new GaugeField(g->geom, QUDA_HALF_PRECISION, g->reconstruct, g->t_boundary, g->anisotropy);
      mgs.gaugeHalf->copyFrom(*g);
      const double orig_mu = param->mu, orig_kappa = param->kappa, orig_mass = param->mass;
      param->kappa *= mgs.mg_param_copy.delta_kappaPR;
      param->mu *= mgs.mg_param_copy.delta_muPR;
      param->mass = 0.5 / param->kappa - 4.0;
      DiracParam dps;
      setDiracPreParam(dps, param, true);
      dps.gauge = mgs.gaugeHalf;
      if (param->dslash_type == QUDA_TWISTED_CLOVER_DSLASH) {
        if (!dps.clover) errorQuda("Clover field not allocated");
        std::vector<double> host((size_t)g->geom.V * 72);
        dps.clover->savePacked(host.data(), QUDA_DOUBLE_PRECISION);
        mgs.cloverHalf = new CloverField(g->geom, QUDA_HALF_PRECISION);
        mgs.cloverHalf->loadPacked(host.data(), nullptr, QUDA_DOUBLE_PRECISION);
        mgs.cloverHalf->computeInverse(4.0 * param->kappa * param->kappa * param->mu * param->mu);
        dps.clover = mgs.cloverHalf;
      }
      mgs.dSmoothHalf = Dirac::create(dps);
      mgs.mSmoothHalf = new DiracM(*mgs.dSmoothHalf);
      param->kappa = orig_kappa; param->mu = orig_mu; param->mass = orig_mass;
    }
    if (mgs.mSmoothHalf) mgs.mg->setSmootherSloppy(mgs.mSmoothHalf);
    HIP_CHECK(hipStreamSynchronize(computeStream()));
  } else {
    mgs.mg->setSmootherSloppy(nullptr);
  }
  setCoarseHalfStorage(on);
}

multigrid_solver::~multigrid_solver() {
  mgProfilePrint();
  delete mg;
  delete mgParam;
  for (ColorSpinorField *f : B) delete f;
  delete mSmoothHalf;
  delete dSmoothHalf;
  delete gaugeHalf;
  delete cloverHalf;
  delete mSmooth;
  delete dSmooth;
  delete m;
  delete d;
}

}  // namespace quda
