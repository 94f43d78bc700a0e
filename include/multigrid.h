// multigrid.h — adaptive multigrid preconditioner (reference include/multigrid.h, lib/multigrid.cpp:11-779).
//
// Hierarchy: level 0 = the full fine operator (Wilson / twisted-mass / twisted-clover, fp32 "precondition" links),
// level l+1 = Galerkin coarse operator of level l through a Transfer built from Nvec block-orthonormalised null
// vectors (BiCGstab on M x = 0 from random guesses, or the restricted vectors of the level above).
// Cycle (reference MG::operator(), :488-604): pre-smooth (MR, nu_pre) -> residual -> R -> coarse solve
// (V-cycle: the coarse MG itself; K-cycle = QUDA_MG_CYCLE_RECURSIVE: GCR(10) preconditioned by it) -> P, correct ->
// post-smooth (MR, nu_post); coarsest level: GCR to smoother_tol.
// Smoother per level: the full operator (smoother_solve_type = QUDA_DIRECT_SOLVE) or — the reference's default — the
// even-odd preconditioned one (QUDA_DIRECT_PC_SOLVE: DiracTwistedMassPC / DiracTwistedCloverPC / DiracWilsonPC on level 0,
// DiracCoarsePC with Xinv / Yhat on coarse levels) wrapped in Dirac::prepare / reconstruct exactly as the reference's
// cycle does.  Outer solve on the full system (QUDA_DIRECT_SOLVE: full residual restricted, coarse-grid solution type
// QUDA_MAT_SOLUTION) or — the way the QKXTM drivers run it (lib/interface_quda.cpp:6041) — on the even-odd preconditioned
// system (QUDA_DIRECT_PC_SOLVE: single-parity residual injected, reference QUDA_MATPC_SOLUTION branch lib/multigrid.cpp:492-560).
#pragma once

#include <vector>

#include "coarse.h"
#include "solver.h"
#include "transfer.h"

namespace quda {

struct MGParam : SolverParam {
  QudaMultigridParam &mg_global;
  int level, Nlevel;
  int geoBlockSize[4];
  int spinBlockSize;
  int Nvec;
  std::vector<ColorSpinorField *> &B;   // null vectors of THIS level (fp32 device full fields)
  int nu_pre, nu_post;
  double smoother_tol;
  QudaMultigridCycleType cycle_type;
  QudaInverterType smoother;
  DiracMatrix &matResidual, &matSmooth;
  QudaTwistFlavorType fineFlavor;
  bool vectorsPreset = false;   // level 0: B already holds the vectors to build the hierarchy from (multigrid_solver::refine)
  MGParam(QudaMultigridParam &g, std::vector<ColorSpinorField *> &B, DiracMatrix &matResidual, DiracMatrix &matSmooth, int level, QudaTwistFlavorType flavor);
};

class CoarseCycle;   // coarse_cycle.h: the cycle below a coarse level as one persistent kernel
struct MGBlockState;  // block_solver.cpp: per-source smoothers / work fields and the block-field cycle below the fine level (invertMultiSrcQuda)

class MG : public Solver {
  MGParam &mgp;
  Transfer *transfer;
  Solver *presmoother, *postsmoother, *coarse_solver;
  SolverParam *param_presmooth, *param_postsmooth, *param_coarse_solver;
  MG *coarse;
  MGParam *param_coarse;
  std::vector<ColorSpinorField *> B_coarse;
  ColorSpinorField *r, *r_coarse, *x_coarse, *b_tilde;
  DiracCoarse *diracCoarse;
  DiracCoarse *diracCoarseSmoother;
  DiracM *matCoarse, *matCoarseSmoother;
  bool pcSmooth;
  bool ownCoarseSolver;
  // levels >= 1 of a small (launch-latency-bound) sub-hierarchy run as ONE persistent kernel (coarse_cycle.h); created at the first cycle,
  // checked once against the kernel-per-operation path below (cycleUnfused), dropped for good if the two disagree
  CoarseCycle *fused = nullptr;
  bool fusedTried = false, fusedVerified = false;
  void cycleUnfused(ColorSpinorField &out, ColorSpinorField &in);
  MGBlockState *blockState = nullptr;
  int lastParityCycle = -1;   // parity of the even-odd system the last operator() call post-smoothed with MR (imageOfLast), -1: none
  bool blockPrepare(int nsrc);
  void cycleParityBlock(std::vector<ColorSpinorField *> &x, std::vector<ColorSpinorField *> &b, const std::vector<char> &active, bool fullResidual);
 public:
  // how generateNullVectors produced this level's vectors: 0 loaded / restricted / sequential BiCGstab solves (the reference's loop),
  // 1 lockstep block BiCGstab on the multi-right-hand-side fine stencil, 2 the same on the MFMA coarse operator; and the lockstep iteration count
  int nullVectorMethod = 0, nullVectorIterations = 0;
 private:
  void generateNullVectors(std::vector<ColorSpinorField *> &B);
  void cycleParity(ColorSpinorField &x, ColorSpinorField &b, bool fullResidual);
  void saveVectors(std::vector<ColorSpinorField *> &B);
  void loadVectors(std::vector<ColorSpinorField *> &B);

 public:
  explicit MG(MGParam &param);
  ~MG() override;
  void operator()(ColorSpinorField &out, ColorSpinorField &in) override;
  unsigned long long flops() const override;
  // self-consistency identities of the reference's MG::verify() (lib/multigrid.cpp:372-486); returns the three worst
  // relative deviations over the hierarchy: |(1 - P P^dag) v_k| / |v_k|, |(1 - P^dag P) eta| / |eta|,
  // |R D P eta - D_c eta| / |D_c eta|
  void verify(double dev[3]);
  const Transfer *getTransfer() const { return transfer; }
  const DiracCoarse *getCoarseDirac() const { return diracCoarse; }
  MG *getCoarse() const { return coarse; }
  // fp16 mirrors of V and of the coarse links (plain and preconditioned) on every level below this one
  void makeHalfMirrors();
  // level-0 smoothers re-created with `sloppy` as their inner (MR work-field) operator in its precision; nullptr: back to fp32
  void setSmootherSloppy(DiracMatrix *sloppy);
  const std::vector<ColorSpinorField *> &nullVectors() const { return mgp.B; }
  // structure of the hierarchy, for the fused coarse cycle
  const MGParam &params() const { return mgp; }
  bool smootherIsPC() const { return pcSmooth; }
  const SolverParam *preSmootherParam() const { return param_presmooth; }
  const CoarseCycle *fusedCycle() const { return fused; }
  void dropFusedCycle();
  // x_i = K b_i for several sources at once (level 0 only): smoothing / R / P per source, everything below the fine level on block fields through
  // the multi-right-hand-side MFMA coarse operator (block_solver.cpp).  false: this hierarchy does not qualify, the caller applies K per source.
  // active[i] = 0: source i is left alone.  blockRelease frees the per-source state.
  bool cycleBlock(std::vector<ColorSpinorField *> &x, std::vector<ColorSpinorField *> &b, const std::vector<char> &active);
  // out_i = Mhat x_i for the solutions x_i of the LAST cycleBlock on single-parity fields, while they are still on the block fields of the fine
  // smoother (two multi-right-hand-side stencil launches per group instead of two stencils per source): the outer solver's `A p_k`.  pc: the
  // even-odd preconditioned operator of the outer solver — it must be the smoother's operator (type, kappa, mu, preconditioning) on fp32 links
  // the block stencil reads.  false: not available, the caller applies its operator source by source.
  bool imageOfLast(ColorSpinorField &Ax, const ColorSpinorField &b, const DiracMatrix &A) override;
  bool blockImageFull(std::vector<ColorSpinorField *> &out, std::vector<ColorSpinorField *> &b, const Dirac &full, const std::vector<char> &active);
  void blockWantImage(int nsrc, bool on);   // the next cycleBlock (of nsrc sources) keeps what blockApplyLast needs to answer from the post-smoother's residual
  bool blockApplyLast(std::vector<ColorSpinorField *> &out, const Dirac &pc, const std::vector<char> &active);
  void blockRelease();   // hierarchy contents changed (half-precision mirrors switched on): rebuild or abandon at the next cycle
  DiracMatrix &residualMatrix() const { return mgp.matResidual; }
};

// opaque object handed out by newMultigridQuda (reference include/multigrid.h:375-411)
struct multigrid_solver {
  Dirac *d;       // level-0 full operator in the preconditioner precision
  DiracM *m;
  Dirac *dSmooth; // level-0 smoother operator (== d, or its even-odd preconditioned form)
  DiracM *mSmooth;
  GaugeField *gaugeHalf;   // 16-bit copy of the links for the half-precision smoother (multigridSetHalfStorage), owned
  CloverField *cloverHalf; // and of the clover term (twisted clover), owned
  Dirac *dSmoothHalf;
  DiracM *mSmoothHalf;
  std::vector<ColorSpinorField *> B;
  MGParam *mgParam;
  MG *mg;
  QudaMultigridParam mg_param_copy;
  QudaInvertParam inv_param_copy;
  explicit multigrid_solver(QudaMultigridParam &mg_param);
  // Set-up refinement (not in the reference, whose null vectors are final once BiCGstab stops): `passes` times, every null vector v of
  // level 0 is replaced by K^cycles v, K = one multigrid cycle of the CURRENT hierarchy as approximate inverse — an inverse iteration
  // through the hierarchy, which enriches the vectors in exactly the low modes the first set-up resolved poorly (at a critical kappa its
  // BiCGstab solves stop at their iteration cap) — and the hierarchy is rebuilt from them (transfer, Galerkin operators, coarse levels).
  void refine(int passes, int cycles);
  ~multigrid_solver();
};

// R, P and the coarse operators of every level stream fp16 mirrors of V and of the coarse links instead of the fp32 masters
// (those kernels are HBM-bound on exactly these bytes); setup, verify and introspection keep using fp32.  Not in the
// reference (its MG is fp32 throughout): opt-in, QUDA_AMD_MG_HALF=1 or qudaAmdMultigridSetHalfStorage.
void multigridSetHalfStorage(multigrid_solver &mgs, bool on);


// Several sources through ONE lockstep (optionally multigrid-preconditioned) GCR — block_solver.cpp, the solver behind invertMultiSrcQuda and the
// propagator loop of the QKXTM drivers.  x_i = A^-1 b_i (x is zeroed: no initial guess), K may be nullptr, sloppyPC: the even-odd preconditioned
// sloppy operator when A is one (lets A p_k come from the smoother's block fields), else nullptr.
struct MultiSrcSolve { int iter = 0; double secs = 0; std::vector<double> r2, b2; };
MultiSrcSolve solveMultiSrcGCR(std::vector<ColorSpinorField *> &x, std::vector<ColorSpinorField *> &b, DiracMatrix &mat, DiracMatrix &matSloppy, MG *K, SolverParam &param, const Dirac *sloppyPC);

}  // namespace quda
