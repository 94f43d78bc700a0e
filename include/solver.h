// solver.h — Krylov layer of the MG-GCR path: GCR (outer solver, K-cycle coarse solver), MR (smoother),
// BiCGstab (null-vector setup).  Interface mirrors the reference (include/invert_quda.h:15-331: SolverParam,
// Solver::create, operator()(x, b)); algorithms restated from lib/inv_gcr_quda.cpp:53-516,
// lib/inv_mr_quda.cpp:40-200, lib/inv_bicgstab_quda.cpp:40-354, lib/solver.cpp:13-150.
#pragma once

#include <vector>

#include "blas.h"
#include "dirac.h"

namespace quda {

struct SolverParam {
  QudaInverterType inv_type = QUDA_GCR_INVERTER;
  QudaInverterType inv_type_precondition = QUDA_INVALID_INVERTER;
  void *preconditioner = nullptr;  // multigrid_solver* when inv_type_precondition == QUDA_MG_INVERTER
  QudaResidualType residual_type = QUDA_L2_RELATIVE_RESIDUAL;
  QudaUseInitGuess use_init_guess = QUDA_USE_INIT_GUESS_NO;
  QudaComputeNullVector compute_null_vector = QUDA_COMPUTE_NULL_VECTOR_NO;
  double delta = 1e-4;  // reliable-update threshold
  bool use_sloppy_partial_accumulator = false;
  int max_res_increase = 1, max_res_increase_total = 10, heavy_quark_check = 10, pipeline = 0;
  double tol = 1e-10, tol_restart = 5e-3, tol_hq = 0;
  double true_res = 0, true_res_hq = 0;
  int maxiter = 1000, iter = 0;
  QudaPrecision precision = QUDA_DOUBLE_PRECISION, precision_sloppy = QUDA_DOUBLE_PRECISION, precision_precondition = QUDA_DOUBLE_PRECISION;
  QudaPreserveSource preserve_source = QUDA_PRESERVE_SOURCE_YES;
  int Nkrylov = 20;
  int precondition_cycle = 1;
  double tol_precondition = 1e-1;
  int maxiter_precondition = 10;
  double omega = 1.0;
  QudaSchwarzType schwarz_type = QUDA_ADDITIVE_SCHWARZ;
  double secs = 0, gflops = 0;
  QudaVerbosity verbosity_precondition = QUDA_SILENT;
  bool is_preconditioner = false;
  bool global_reduction = true;
  bool compute_true_res = true;
  SolverParam() {}
  explicit SolverParam(const QudaInvertParam &p);   // reference include/invert_quda.h:197-255
  void updateInvertParam(QudaInvertParam &p) const; // reference :262-300
};

class Solver {
 protected:
  SolverParam &param;
 public:
  explicit Solver(SolverParam &p) : param(p) {}
  virtual ~Solver() {}
  virtual void operator()(ColorSpinorField &out, ColorSpinorField &in) = 0;
  virtual unsigned long long flops() const { return 0; }
  // b - A x of the system the last call worked on, in the solver's work precision, if the solver keeps it (MR); else nullptr
  virtual const ColorSpinorField *lastResidual() const { return nullptr; }
  // used as a preconditioner x = K b: A x for the x of the LAST call, if the solver can give it without applying A (a multigrid cycle that ends in
  // an MR smoother on A's own even-odd system holds b - A x: multigrid.cpp); false: not available, the caller applies A
  virtual bool imageOfLast(ColorSpinorField &, const ColorSpinorField &, const DiracMatrix &) { return false; }
  static Solver *create(SolverParam &param, DiracMatrix &mat, DiracMatrix &matSloppy, DiracMatrix &matPrecon);  // reference lib/solver.cpp:13
  static double stopping(double tol, double b2, QudaResidualType type);
  bool convergence(double r2, double hq2, double r2_tol, double hq_tol) const;
  void PrintStats(const char *name, int k, double r2, double b2, double hq2) const;
  void PrintSummary(const char *name, int k, double r2, double b2) const;
};

class MR : public Solver {
  const DiracMatrix &mat, &matSloppy;
  ColorSpinorField *rp, *Arp, *tmpp, *yp;
  bool residualValid;
 public:
  MR(DiracMatrix &mat, DiracMatrix &matSloppy, SolverParam &param);
  ~MR() override;
  void operator()(ColorSpinorField &out, ColorSpinorField &in) override;
  const ColorSpinorField *lastResidual() const override { return residualValid ? rp : nullptr; }
};

class BiCGstab : public Solver {
  DiracMatrix &mat, &matSloppy, &matPrecon;
  ColorSpinorField *yp, *rp, *pp, *vp, *tp, *r0p, *xsp, *rsp;
 public:
  BiCGstab(DiracMatrix &mat, DiracMatrix &matSloppy, DiracMatrix &matPrecon, SolverParam &param);
  ~BiCGstab() override;
  void operator()(ColorSpinorField &out, ColorSpinorField &in) override;
};

class GCR : public Solver {
  const DiracMatrix &mat, &matSloppy, &matPrecon;
  Solver *K;
  bool ownK;
  SolverParam Kparam;
  int nKrylov;
  Complex *alpha, **beta;
  double *gamma;
  bool init;
  ColorSpinorField *rp, *yp, *x_sloppy, *r_sloppy, *p_pre, *r_pre, *rM;
  std::vector<ColorSpinorField *> p, Ap;
 public:
  GCR(DiracMatrix &mat, DiracMatrix &matSloppy, DiracMatrix &matPrecon, SolverParam &param);
  GCR(DiracMatrix &mat, Solver &K, DiracMatrix &matSloppy, DiracMatrix &matPrecon, SolverParam &param);
  ~GCR() override;
  void operator()(ColorSpinorField &out, ColorSpinorField &in) override;
};

void fillInnerSolveParam(SolverParam &inner, const SolverParam &outer);  // reference lib/inv_gcr_quda.cpp:17-50

}  // namespace quda
