// solver.cpp — GCR, MR, BiCGstab.  Algorithms restated from the reference (see solver.h); every vector
// operation is one fused HBM-streaming kernel of blas.hip, every matrix application one or two stencil launches.
#include "solver.h"

#include <cmath>
#include <sys/time.h>

#include "multigrid.h"

namespace quda {

static double now() {
  timeval t;
  gettimeofday(&t, nullptr);
  return t.tv_sec + 1e-6 * t.tv_usec;
}

SolverParam::SolverParam(const QudaInvertParam &p)
    : inv_type(p.inv_type), inv_type_precondition(p.inv_type_precondition), preconditioner(p.preconditioner), residual_type(p.residual_type),
      use_init_guess(p.use_init_guess), delta(p.reliable_delta), use_sloppy_partial_accumulator(p.use_sloppy_partial_accumulator != 0),
      max_res_increase(p.max_res_increase), max_res_increase_total(p.max_res_increase_total), heavy_quark_check(p.heavy_quark_check),
      pipeline(p.pipeline), tol(p.tol), tol_restart(p.tol_restart), tol_hq(p.tol_hq), true_res(p.true_res), true_res_hq(p.true_res_hq),
      maxiter(p.maxiter), iter(p.iter), precision(p.cuda_prec), precision_sloppy(p.cuda_prec_sloppy),
      precision_precondition(p.cuda_prec_precondition), preserve_source(p.preserve_source), Nkrylov(p.gcrNkrylov),
      precondition_cycle(p.precondition_cycle), tol_precondition(p.tol_precondition), maxiter_precondition(p.maxiter_precondition),
      omega(p.omega), schwarz_type(p.schwarz_type), secs(p.secs), gflops(p.gflops), verbosity_precondition(p.verbosity_precondition) {
  if (precondition_cycle < 1) precondition_cycle = 1;
  if ((int)residual_type == QUDA_INVALID_ENUM || residual_type == 0) residual_type = QUDA_L2_RELATIVE_RESIDUAL;
}

void SolverParam::updateInvertParam(QudaInvertParam &p) const {
  p.true_res = true_res;
  p.true_res_hq = true_res_hq;
  p.iter += iter;
  p.gflops += gflops;
  p.secs += secs;
}

void fillInnerSolveParam(SolverParam &inner, const SolverParam &outer) {
  inner.tol = outer.tol_precondition;
  inner.maxiter = outer.maxiter_precondition;
  inner.delta = 1e-20;  // no reliable updates within the inner solver
  inner.precision = outer.precision_precondition;
  inner.precision_sloppy = outer.precision_precondition;
  inner.iter = 0; inner.gflops = 0; inner.secs = 0;
  inner.inv_type_precondition = QUDA_INVALID_INVERTER;
  inner.is_preconditioner = true;
  inner.global_reduction = false;
  inner.use_init_guess = QUDA_USE_INIT_GUESS_NO;
  inner.preserve_source = (outer.inv_type == QUDA_GCR_INVERTER && outer.precision_sloppy != outer.precision_precondition) ? QUDA_PRESERVE_SOURCE_NO
                                                                                                                          : QUDA_PRESERVE_SOURCE_YES;
}

double Solver::stopping(double tol, double b2, QudaResidualType type) {  // reference lib/solver.cpp:109-125
  double stop = 0.0;
  if (type & QUDA_L2_ABSOLUTE_RESIDUAL) stop = tol * tol;
  else stop = b2 * tol * tol;
  return stop;
}
bool Solver::convergence(double r2, double hq2, double r2_tol, double hq_tol) const {
  if ((param.residual_type & QUDA_HEAVY_QUARK_RESIDUAL) && hq2 > hq_tol) return false;
  if ((param.residual_type & (QUDA_L2_RELATIVE_RESIDUAL | QUDA_L2_ABSOLUTE_RESIDUAL)) && r2 > r2_tol) return false;
  return true;
}
void Solver::PrintStats(const char *name, int k, double r2, double b2, double hq2) const {
  if (getVerbosity() >= QUDA_VERBOSE) printfQuda("%s: %d iterations, <r,r> = %e, |r|/|b| = %e\n", name, k, r2, sqrt(r2 / b2));
  if (std::isnan(r2)) errorQuda("Solver appears to have diverged");
}
void Solver::PrintSummary(const char *name, int k, double r2, double b2) const {
  if (getVerbosity() >= QUDA_SUMMARIZE)
    printfQuda("%s: Convergence at %d iterations, L2 relative residual: iterated = %e, true = %e\n", name, k, sqrt(r2 / b2), param.true_res);
}

Solver *Solver::create(SolverParam &param, DiracMatrix &mat, DiracMatrix &matSloppy, DiracMatrix &matPrecon) {
  switch (param.inv_type) {
    case QUDA_GCR_INVERTER:
      if (param.inv_type_precondition == QUDA_MG_INVERTER) {
        if (!param.preconditioner) errorQuda("GCR with multigrid preconditioner: param.preconditioner is NULL");
        multigrid_solver *mg = static_cast<multigrid_solver *>(param.preconditioner);
        return new GCR(mat, *(mg->mg), matSloppy, matPrecon, param);
      }
      return new GCR(mat, matSloppy, matPrecon, param);
    case QUDA_MR_INVERTER: return new MR(mat, matSloppy, param);
    case QUDA_BICGSTAB_INVERTER: return new BiCGstab(mat, matSloppy, matPrecon, param);
    default: errorQuda("Invalid solver type %d (GCR, MR and BiCGstab are on the MG-GCR path)", param.inv_type);
  }
  return nullptr;
}

static ColorSpinorField *like(const ColorSpinorField &x, QudaPrecision prec, bool zero) {
  ColorSpinorParam p = x.param();
  p.location = QUDA_CUDA_FIELD_LOCATION;
  p.precision = prec;
  p.create = zero ? QUDA_ZERO_FIELD_CREATE : QUDA_NULL_FIELD_CREATE;
  return new ColorSpinorField(p);
}

// ================================================================================================
// MR — used as the smoother (fixed iteration count, relaxation omega)
// ================================================================================================
MR::MR(DiracMatrix &mat_, DiracMatrix &matSloppy_, SolverParam &p) : Solver(p), mat(mat_), matSloppy(matSloppy_), rp(nullptr), Arp(nullptr), tmpp(nullptr), yp(nullptr), residualValid(false) {}
MR::~MR() { delete rp; delete Arp; delete tmpp; delete yp; }

// The recurrence is the reference's (lib/inv_mr_quda.cpp:40-200: alpha = (Ar, r) / |Ar|^2, x += omega alpha r, r -= omega alpha Ar); what
// is re-designed is the traffic around it, because as the multigrid smoother it runs two iterations at a time and the BLAS around
// four stencil applications was 10 % of a 48^3 x 96 solve (profiles/r03a_c5_mg_solve_table.json: copy b, norm, zero x, zero y,
// normalise r, ..., scale y, copy to x = 25 field passes for 2 iterations):
//   * no normalisation of the residual (alpha is scale-invariant; 16-bit fields carry a scale per site) — the source's norm is only
//     computed where somebody asks for it;
//   * the iteration runs on x itself when x has the work precision, and the first one of a zero start reads b in place of r and writes
//     x and r without reading them (CaxInitF): 13 passes for 2 iterations, 17 with an initial guess;
//   * the residual the iteration ends with, b - A x in the work precision, stays available (lastResidual): the multigrid cycle
//     restricts it instead of applying the operator once more.
void MR::operator()(ColorSpinorField &x, ColorSpinorField &b) {
  blas::setGlobalReduction(param.global_reduction);
  auto fits = [&](ColorSpinorField *f) { return f && f->VolumeCB() == x.VolumeCB() && f->SiteSubset() == x.SiteSubset() && f->Ncolor() == x.Ncolor(); };
  if (!fits(Arp)) {
    delete rp; delete Arp; delete tmpp; delete yp;
    Arp = like(x, param.precision_sloppy, true);
    rp = like(x, param.precision_sloppy, true);
    yp = nullptr;
    tmpp = nullptr;
  }
  const bool same = x.Precision() == param.precision_sloppy;
  if (!same && !yp) yp = like(x, param.precision_sloppy, true);
  for (ColorSpinorField *f : {rp, Arp, yp}) if (f) f->twistFlavor = b.twistFlavor;
  ColorSpinorField &r = *rp, &Ar = *Arp;
  ColorSpinorField &y = same ? x : *yp;   // where the iteration accumulates
  const double t0 = now();
  const bool guess = param.use_init_guess == QUDA_USE_INIT_GUESS_YES;
  const double omega = param.omega;
  residualValid = false;
  if (!param.is_preconditioner) blas::flops = 0;
  const double b2 = param.is_preconditioner ? 0.0 : blas::norm2(b);

  bool fresh = !guess;   // y (and, for `same`, r) not written yet: the first iteration defines them
  if (guess) {
    if (same) { matSloppy(r, x); blas::axpby(1.0, b, -1.0, r); }                           // r = b - A x, then iterate on x itself
    else { blas::copy(y, x); matSloppy(r, y); blas::copy(y, b); blas::axpby(1.0, y, -1.0, r); blas::zero(y); }
  } else if (!same) {
    blas::copy(r, b);   // precision change
  }
  int k = 0;
  // as a smoother (fixed number of steps, rank-local sums): alpha stays on the device, no host round trip per step (blas.h deviceScalars)
  const bool devAlpha = param.is_preconditioner && getVerbosity() < QUDA_DEBUG_VERBOSE && blas::deviceScalars();
  while (k < param.maxiter) {
    const ColorSpinorField &rin = (fresh && same) ? b : r;   // zero start in the work precision: b IS the residual
    matSloppy(Ar, rin);
    if (devAlpha) {
      blas::cDotProductNormADev(Ar, rin);
      if (fresh && same) blas::caxInitDev(omega, b, y, Ar, r);
      else if (fresh) blas::caxXmazDev(omega, r, y, Ar);
      else blas::caxpyXmazDev(omega, r, y, Ar);
      fresh = false;
      k++;
      continue;
    }
    const double3_t Ar3 = blas::cDotProductNormA(Ar, rin);
    if (!(Ar3.z > 0.0)) break;   // zero source (or breakdown): nothing to add
    const Complex alpha = omega * Complex(Ar3.x, Ar3.y) / Ar3.z;
    if (fresh && same) blas::caxInit(alpha, b, y, Ar, r);        // y = a b ; r = b - a Ar
    else if (fresh) blas::caxXmaz(alpha, r, y, Ar);              // y = a r ; r -= a Ar
    else blas::caxpyXmaz(alpha, r, y, Ar);                       // y += a r ; r -= a Ar
    fresh = false;
    k++;
    if (getVerbosity() >= QUDA_DEBUG_VERBOSE) printfQuda("MR: %d iterations, <r|A|r> = (%e, %e)\n", k, Ar3.x, Ar3.y);
  }
  if (fresh) {   // no iteration happened
    if (!guess) { blas::zero(x); if (same) blas::copy(r, b); }
  } else if (!same) {
    if (guess) { ColorSpinorField *t = like(x, x.Precision(), false); blas::copy(*t, y); blas::xpy(*t, x); delete t; }
    else blas::copy(x, y);
  }
  residualValid = true;
  if (!param.is_preconditioner) {
    param.secs += now() - t0;
    param.gflops += (blas::flops + mat.flops() + matSloppy.flops()) * 1e-9;
    param.iter += k;
    if (param.preserve_source == QUDA_PRESERVE_SOURCE_YES && param.compute_true_res) {
      ColorSpinorField *t = like(x, x.Precision(), false), *bb = like(x, x.Precision(), false);
      t->twistFlavor = bb->twistFlavor = b.twistFlavor;
      mat(*t, x);
      blas::copy(*bb, b);
      param.true_res = sqrt(blas::xmyNorm(*bb, *t) / b2);
      delete t; delete bb;
      if (getVerbosity() >= QUDA_SUMMARIZE) printfQuda("MR: Converged after %d iterations, relative residual: true = %e\n", k, param.true_res);
    }
    blas::flops = 0;
  }
  blas::setGlobalReduction(true);  // re-enable global reductions for the outer solver (reference :198)
}

// ================================================================================================
// BiCGstab — null-vector generation and generic use
// ================================================================================================
BiCGstab::BiCGstab(DiracMatrix &mat_, DiracMatrix &matSloppy_, DiracMatrix &matPrecon_, SolverParam &p)
    : Solver(p), mat(mat_), matSloppy(matSloppy_), matPrecon(matPrecon_), yp(nullptr), rp(nullptr), pp(nullptr), vp(nullptr), tp(nullptr), r0p(nullptr),
      xsp(nullptr), rsp(nullptr) {}
BiCGstab::~BiCGstab() { delete yp; delete rp; delete pp; delete vp; delete tp; delete r0p; delete xsp; delete rsp; }

static int reliable(double &rNorm, double &maxrx, double &maxrr, const double &r2, const double &delta) {  // reference lib/inv_bicgstab_quda.cpp:28-38
  rNorm = sqrt(r2);
  if (rNorm > maxrx) maxrx = rNorm;
  if (rNorm > maxrr) maxrr = rNorm;
  int updateR = (rNorm < delta * maxrr) ? 1 : 0;
  return updateR;
}

void BiCGstab::operator()(ColorSpinorField &x, ColorSpinorField &b) {
  delete yp; delete rp; delete pp; delete vp; delete tp; delete r0p; delete xsp; delete rsp;
  yp = like(x, x.Precision(), true); rp = like(x, x.Precision(), true);
  pp = like(x, param.precision_sloppy, true); vp = like(x, param.precision_sloppy, true); tp = like(x, param.precision_sloppy, true);
  r0p = like(x, param.precision_sloppy, true);
  const bool mixed = param.precision_sloppy != x.Precision();
  rsp = mixed ? like(x, param.precision_sloppy, true) : nullptr;
  xsp = mixed ? like(x, param.precision_sloppy, true) : nullptr;
  for (ColorSpinorField *f : {yp, rp, pp, vp, tp, r0p, rsp, xsp}) if (f) f->twistFlavor = x.twistFlavor;
  ColorSpinorField &y = *yp, &r = *rp, &p = *pp, &v = *vp, &t = *tp, &r0 = *r0p;
  const double t0 = now();

  double b2 = blas::norm2(b), r2;
  if (param.use_init_guess == QUDA_USE_INIT_GUESS_YES) {
    mat(r, x);
    r2 = blas::xmyNorm(b, r);
    blas::copy(y, x);
  } else {
    blas::copy(r, b);
    r2 = b2;
    blas::zero(x);
  }
  if (b2 == 0) {
    if (param.compute_null_vector == QUDA_COMPUTE_NULL_VECTOR_NO) {
      warningQuda("inverting on zero-field source");
      blas::copy(x, b);
      param.true_res = 0.0; param.true_res_hq = 0.0;
      return;
    } else if (param.use_init_guess == QUDA_USE_INIT_GUESS_YES) {
      b2 = r2;  // null-vector mode: b = 0, x0 random, b2 := |A x0|^2 (reference :96-127)
    } else {
      errorQuda("Null vector computing requires non-zero guess!");
    }
  }
  ColorSpinorField &rSloppy = mixed ? *rsp : r;
  ColorSpinorField &xSloppy = mixed ? *xsp : x;
  if (mixed) blas::copy(rSloppy, r);
  blas::copy(r0, rSloppy);  // shadow residual
  blas::zero(xSloppy);

  const double stop = stopping(param.tol, b2, param.residual_type);
  const double delta = param.delta;
  int k = 0, rUpdate = 0;
  Complex rho(1.0, 0.0), rho0 = rho, alpha(1.0, 0.0), omega(1.0, 0.0), beta;
  double rNorm = sqrt(r2), maxrr = rNorm, maxrx = rNorm;
  PrintStats("BiCGstab", k, r2, b2, 0.0);
  if (!param.is_preconditioner) blas::flops = 0;
  rho = r2;
  blas::copy(p, rSloppy);
  while (!convergence(r2, 0.0, stop, param.tol_hq) && k < param.maxiter) {
    matSloppy(v, p);
    const Complex r0v = blas::cDotProduct(r0, v);
    alpha = (std::abs(rho) == 0.0) ? Complex(0.0) : rho / r0v;
    blas::caxpy(-alpha, v, rSloppy);   // r -= alpha v
    matSloppy(t, rSloppy);
    const double3_t ot = blas::cDotProductNormA(t, rSloppy);
    omega = Complex(ot.x / ot.z, ot.y / ot.z);
    // x += alpha p + omega r ; r -= omega t ; rho = (r0, r), r2 = |r|^2
    blas::caxpbypzYmbw(alpha, p, omega, rSloppy, xSloppy, t);
    const double3_t rr = blas::cDotProductNormB(r0, rSloppy);
    rho0 = rho;
    rho = Complex(rr.x, rr.y);
    r2 = rr.z;
    const int updateR = reliable(rNorm, maxrx, maxrr, r2, delta);
    if (updateR) {
      if (mixed) blas::copy(x, xSloppy);
      blas::xpy(x, y);
      mat(r, y);
      r2 = blas::xmyNorm(b, r);
      if (mixed) blas::copy(rSloppy, r);
      blas::zero(xSloppy);
      rNorm = sqrt(r2); maxrr = rNorm; maxrx = rNorm;
      rUpdate++;
    }
    k++;
    PrintStats("BiCGstab", k, r2, b2, 0.0);
    beta = (std::abs(rho * alpha) == 0.0) ? Complex(0.0) : (rho / rho0) * (alpha / omega);
    blas::cxpaypbz(rSloppy, -beta * omega, v, beta, p);  // p = r - beta omega v + beta p
  }
  if (mixed) blas::copy(x, xSloppy);
  blas::xpy(y, x);
  param.secs += now() - t0;
  param.gflops += (blas::flops + mat.flops() + matSloppy.flops() + matPrecon.flops()) * 1e-9;
  param.iter += k;
  if (k == param.maxiter && param.compute_null_vector == QUDA_COMPUTE_NULL_VECTOR_NO) warningQuda("Exceeded maximum iterations %d", param.maxiter);
  if (getVerbosity() >= QUDA_VERBOSE) printfQuda("BiCGstab: Reliable updates = %d\n", rUpdate);
  if (!param.is_preconditioner) {
    mat(r, x);
    param.true_res = sqrt(blas::xmyNorm(b, r) / b2);
    PrintSummary("BiCGstab", k, r2, b2);
  }
  blas::flops = 0;
}

// ================================================================================================
// GCR — flexible, restarted, with reliable updates (outer solver; K = multigrid V/K-cycle)
// ================================================================================================
static void allocCoeffs(int n, Complex *&alpha, Complex **&beta, double *&gamma) {
  alpha = new Complex[n];
  beta = new Complex *[n];
  for (int i = 0; i < n; i++) beta[i] = new Complex[n];
  gamma = new double[n];
}

GCR::GCR(DiracMatrix &mat_, DiracMatrix &matSloppy_, DiracMatrix &matPrecon_, SolverParam &p)
    : Solver(p), mat(mat_), matSloppy(matSloppy_), matPrecon(matPrecon_), K(nullptr), ownK(true), Kparam(p), nKrylov(p.Nkrylov), init(false), rp(nullptr),
      yp(nullptr), x_sloppy(nullptr), r_sloppy(nullptr), p_pre(nullptr), r_pre(nullptr), rM(nullptr) {
  fillInnerSolveParam(Kparam, p);
  if (p.inv_type_precondition == QUDA_MR_INVERTER) K = new MR(matPrecon_, matPrecon_, Kparam);
  else if (p.inv_type_precondition == QUDA_BICGSTAB_INVERTER) K = new BiCGstab(matPrecon_, matPrecon_, matPrecon_, Kparam);
  else if (p.inv_type_precondition == QUDA_INVALID_INVERTER) K = nullptr;
  else errorQuda("Unsupported preconditioner %d", p.inv_type_precondition);
  this->p.resize(nKrylov, nullptr);
  Ap.resize(nKrylov, nullptr);
  allocCoeffs(nKrylov, alpha, beta, gamma);
}

GCR::GCR(DiracMatrix &mat_, Solver &K_, DiracMatrix &matSloppy_, DiracMatrix &matPrecon_, SolverParam &p)
    : Solver(p), mat(mat_), matSloppy(matSloppy_), matPrecon(matPrecon_), K(&K_), ownK(false), Kparam(p), nKrylov(p.Nkrylov), init(false), rp(nullptr),
      yp(nullptr), x_sloppy(nullptr), r_sloppy(nullptr), p_pre(nullptr), r_pre(nullptr), rM(nullptr) {
  this->p.resize(nKrylov, nullptr);
  Ap.resize(nKrylov, nullptr);
  allocCoeffs(nKrylov, alpha, beta, gamma);
}

GCR::~GCR() {
  delete[] alpha;
  for (int i = 0; i < nKrylov; i++) delete[] beta[i];
  delete[] beta;
  delete[] gamma;
  if (K && ownK) delete K;
  delete rM;
  if (x_sloppy && param.precision_sloppy != param.precision) { delete x_sloppy; delete r_sloppy; }
  if (p_pre) { delete p_pre; delete r_pre; }
  for (int i = 0; i < nKrylov; i++) { delete p[i]; delete Ap[i]; }
  delete rp;
  delete yp;
}

// reference :86-123 (pipeline 0/1 forms)
static void orthoDir(Complex **beta, std::vector<ColorSpinorField *> &Ap, int k, int pipeline) {
  if (pipeline == 0) {
    for (int i = 0; i < k; i++) {
      beta[i][k] = blas::cDotProduct(*Ap[i], *Ap[k]);
      blas::caxpy(-beta[i][k], *Ap[i], *Ap[k]);
    }
  } else {
    if (k == 0) return;
    beta[0][k] = blas::cDotProduct(*Ap[0], *Ap[k]);
    for (int i = 0; i < k - 1; i++) beta[i + 1][k] = blas::caxpyDotzy(-beta[i][k], *Ap[i], *Ap[k], *Ap[i + 1]);
    blas::caxpy(-beta[k - 1][k], *Ap[k - 1], *Ap[k]);
  }
}

// reference :125-157
static void updateSolution(ColorSpinorField &x, const Complex *alpha, Complex **const beta, double *gamma, int k, std::vector<ColorSpinorField *> &p) {
  std::vector<Complex> delta(k);
  for (int i = k - 1; i >= 0; i--) {
    delta[i] = alpha[i];
    for (int j = i + 1; j < k; j++) delta[i] -= beta[i][j] * delta[j];
    delta[i] /= gamma[i];
  }
  // one pass over the k directions (blas::multiCaxpy) instead of k read-modify-write passes over x
  if (blas::multiSupported(x, k)) blas::multiCaxpy(delta.data(), p, k, x);
  else for (int i = 0; i < k; i++) blas::caxpy(delta[i], *p[i], x);
}

void GCR::operator()(ColorSpinorField &x, ColorSpinorField &b) {
  if (init && (rp->VolumeCB() != x.VolumeCB() || rp->SiteSubset() != x.SiteSubset() || rp->Ncolor() != x.Ncolor() || rp->Precision() != x.Precision())) {
    // operand geometry changed (solver object reused on another level / subset): rebuild the work space
    if (x_sloppy && param.precision_sloppy != param.precision) { delete x_sloppy; delete r_sloppy; }
    if (p_pre) { delete p_pre; delete r_pre; p_pre = r_pre = nullptr; }
    for (int i = 0; i < nKrylov; i++) { delete p[i]; delete Ap[i]; p[i] = Ap[i] = nullptr; }
    delete rp; delete yp; delete rM; rM = nullptr;
    init = false;
  }
  if (!init) {
    rp = like(x, x.Precision(), false);
    yp = like(x, x.Precision(), false);
    for (int i = 0; i < nKrylov; i++) { p[i] = like(x, param.precision_sloppy, false); Ap[i] = like(x, param.precision_sloppy, false); }
    if (param.precision_sloppy != x.Precision()) { x_sloppy = like(x, param.precision_sloppy, false); r_sloppy = like(x, param.precision_sloppy, false); }
    else { x_sloppy = nullptr; r_sloppy = nullptr; }
    if (param.precision_precondition != param.precision_sloppy || param.precondition_cycle > 1) {
      p_pre = like(x, param.precision_precondition, false);
      r_pre = like(x, param.precision_precondition, false);
    }
    if (param.precondition_cycle > 1) rM = like(x, param.precision_sloppy, false);
    init = true;
  }
  const bool mixed = param.precision_sloppy != x.Precision();
  ColorSpinorField &r = *rp, &y = *yp;
  ColorSpinorField &xSloppy = mixed ? *x_sloppy : x;
  ColorSpinorField &rSloppy = mixed ? *r_sloppy : r;
  const bool precMatch = !(param.precision_precondition != param.precision_sloppy || param.precondition_cycle > 1);
  ColorSpinorField &rPre = precMatch ? rSloppy : *r_pre;
  for (ColorSpinorField *f : {rp, yp, x_sloppy, r_sloppy, p_pre, r_pre, rM}) if (f) f->twistFlavor = b.twistFlavor;
  for (int i = 0; i < nKrylov; i++) { p[i]->twistFlavor = b.twistFlavor; Ap[i]->twistFlavor = b.twistFlavor; }
  blas::setGlobalReduction(param.global_reduction);
  const double t0 = now();
  blas::zero(y);

  const double b2 = blas::norm2(b);
  double r2;
  if (param.use_init_guess == QUDA_USE_INIT_GUESS_YES) {
    mat(r, x);
    r2 = blas::xmyNorm(b, r);
    blas::copy(y, x);
    if (&x == &xSloppy) blas::zero(x);
  } else {
    blas::copy(r, b);
    r2 = b2;
    blas::zero(x);
    if (&x != &xSloppy) blas::zero(xSloppy);
  }
  if (b2 == 0) {
    if (!param.is_preconditioner) warningQuda("inverting on zero-field source");
    blas::copy(x, b);
    param.true_res = 0.0; param.true_res_hq = 0.0;
    blas::setGlobalReduction(true);
    return;
  }
  const double stop = stopping(param.tol, b2, param.residual_type);
  const int maxResIncrease = param.max_res_increase, maxResIncreaseTotal = param.max_res_increase_total;
  int resIncrease = 0, resIncreaseTotal = 0;
  if (!param.is_preconditioner) blas::flops = 0;
  if (mixed) blas::copy(rSloppy, r);
  int total_iter = 0, restart = 0;
  double r2_old = r2;
  bool l2_converge = false;
  const int pipeline = param.pipeline > 1 ? 1 : (param.pipeline == 0 ? 1 : param.pipeline);
  int k = 0;
  bool blockedOrthoOk = true;   // cleared for the rest of a Krylov cycle once the blocked orthogonalisation has shown a loss of orthogonality
  PrintStats("GCR", total_iter + k, r2, b2, 0.0);
  while (!convergence(r2, 0.0, stop, param.tol_hq) && total_iter < param.maxiter) {
    for (int m = 0; m < param.precondition_cycle; m++) {
      if (K) {
        ColorSpinorField &pPre = precMatch ? *p[k] : *p_pre;
        if (m == 0) { if (!precMatch) blas::copy(rPre, rSloppy); }
        else { blas::copy(*rM, rSloppy); blas::axpy(-1.0, *Ap[k], *rM); blas::copy(rPre, *rM); }
        (*K)(pPre, rPre);
        blas::setGlobalReduction(param.global_reduction);
        if (m == 0) { if (!precMatch) blas::copy(*p[k], pPre); }
        else { blas::copy(*Ap[k], pPre); blas::xpy(*Ap[k], *p[k]); }
      } else {
        blas::copy(*p[k], rSloppy);
      }
      // a multigrid cycle that ends in an MR smoother on this operator's even-odd system knows A p_k already (MG::imageOfLast)
      if (!(K && param.precondition_cycle == 1 && precMatch && K->imageOfLast(*Ap[k], rSloppy, matSloppy))) matSloppy(*Ap[k], *p[k]);
    }
    // Blocked orthogonalisation (reference :53-84, :103-121 pipelined forms): all k inner products (Ap_i, Ap_k), (Ap_k, r) and |Ap_k|^2
    // in ONE sweep, then the k updates, the normalisation and the residual update in ONE sweep — 2 k + 6 field passes instead of the
    // 4 (k - 1) + 9 of the one-direction-at-a-time chain below.  Classical Gram-Schmidt: the norm of the orthogonalised vector is
    // |Ap_k|^2 - sum |beta_i|^2 (the Ap_i are orthonormal) and (Ap_i, r) = 0 for i < k, so (Ap_k', r) = (Ap_k, r).  Where that
    // difference loses more than two digits (the new direction lies almost in the span of the old ones) the iteration falls back to
    // the sequential chain, which measures the norm instead of inferring it.
    bool blocked = false;
    if (blockedOrthoOk && blas::multiSupported(*Ap[k], k)) {
      std::vector<Complex> bk(k > 0 ? k : 1);
      Complex apr; double apn;
      blas::multiDot(bk.data(), apr, apn, Ap, k, *Ap[k], rSloppy);
      double g2 = apn;
      for (int i = 0; i < k; i++) g2 -= std::norm(bk[i]);
      if (apn == 0.0) errorQuda("GCR breakdown");
      if (g2 > 1e-2 * apn) {
        for (int i = 0; i < k; i++) { beta[i][k] = bk[i]; bk[i] = -bk[i]; }
        gamma[k] = sqrt(g2);
        alpha[k] = apr / gamma[k];
        double y2;
        blas::multiCaxpyResidual(r2, y2, bk.data(), Ap, k, 1.0 / gamma[k], *Ap[k], alpha[k], rSloppy);   // Ap_k = (Ap_k - sum beta_i Ap_i) / gamma ; r -= alpha Ap_k
        blocked = true;
        // y2 is the MEASURED |Ap_k|^2 after the normalisation with the inferred gamma: 1 up to round-off while the old directions are
        // orthonormal.  Where it is not (single-pass classical Gram-Schmidt in fp32 over up to 20 directions), the direction is put right
        // with the measured norm — Ap_k, gamma and alpha rescaled consistently, r untouched: alpha Ap_k is invariant under that rescaling
        // — and the rest of this Krylov cycle goes through the sequential chain, which measures instead of inferring.
        if (k > 0 && std::fabs(y2 - 1.0) > 1e-3 && y2 > 0.0) {
          const double s = sqrt(y2);
          blas::ax(1.0 / s, *Ap[k]);
          gamma[k] *= s;
          alpha[k] *= s;
          blockedOrthoOk = false;
        }
      }
    }
    if (!blocked) {
      orthoDir(beta, Ap, k, pipeline);
      const double3_t Apr = blas::cDotProductNormA(*Ap[k], rSloppy);
      gamma[k] = sqrt(Apr.z);
      if (gamma[k] == 0.0) errorQuda("GCR breakdown");
      alpha[k] = Complex(Apr.x, Apr.y) / gamma[k];
      r2 = blas::cabxpyAxNorm(1.0 / gamma[k], -alpha[k], *Ap[k], rSloppy);  // Ap /= |Ap| ; r -= alpha Ap
    }
    k++;
    total_iter++;
    PrintStats("GCR", total_iter, r2, b2, 0.0);
    if (k == nKrylov || total_iter == param.maxiter || (r2 < stop && !l2_converge) || sqrt(r2 / r2_old) < param.delta) {
      updateSolution(xSloppy, alpha, beta, gamma, k, p);
      if (mixed) blas::copy(x, xSloppy);
      blas::xpy(x, y);
      mat(r, y);
      r2 = blas::xmyNorm(b, r);
      if (r2 > r2_old) {
        resIncrease++;
        resIncreaseTotal++;
        warningQuda("GCR: new reliable residual norm %e is greater than previous reliable residual norm %e (total #inc %i)", sqrt(r2), sqrt(r2_old), resIncreaseTotal);
        if (resIncrease > maxResIncrease || resIncreaseTotal > maxResIncreaseTotal) {
          warningQuda("GCR: solver exiting due to too many true residual norm increases");
          break;
        }
      } else {
        resIncrease = 0;
      }
      k = 0;
      blockedOrthoOk = true;
      if (!convergence(r2, 0.0, stop, param.tol_hq)) {
        restart++;
        PrintStats("GCR (restart)", restart, r2, b2, 0.0);
        if (mixed) blas::copy(rSloppy, r);
        blas::zero(xSloppy);
        r2_old = r2;
        if (r2 < stop) l2_converge = true;
      }
      r2_old = r2;
    }
  }
  if (total_iter > 0) blas::copy(x, y);
  param.secs += now() - t0;
  double gf = (blas::flops + mat.flops() + matSloppy.flops() + matPrecon.flops()) * 1e-9;
  if (K) gf += K->flops() * 1e-9;
  if (getVerbosity() >= QUDA_VERBOSE && !param.is_preconditioner) printfQuda("GCR: number of restarts = %d\n", restart);
  if (param.compute_true_res) {
    mat(r, x);
    const double true_res = blas::xmyNorm(b, r);
    param.true_res = sqrt(true_res / b2);
    param.true_res_hq = 0.0;
  }
  param.gflops += gf;
  param.iter += total_iter;
  if (!param.is_preconditioner) { blas::flops = 0; PrintSummary("GCR", total_iter, r2, b2); }
  blas::setGlobalReduction(true);
}

}  // namespace quda
