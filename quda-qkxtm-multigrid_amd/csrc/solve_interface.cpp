// solve_interface.cpp — invertQuda / newMultigridQuda / destroyMultigridQuda (reference lib/interface_quda.cpp:1386-1490
// createDirac / massRescale, :2161-2275 multigrid, :2276-2540 invertQuda).
#include <cmath>

#include <sys/time.h>
#include <cstdlib>
#include "blas.h"
#include "interface_internal.h"
#include "multigrid.h"
#include "quda_amd_ext.h"
#include "solver.h"

void setTuning(QudaTune tune);   // include/util_quda.h

using namespace quda;

namespace quda {

// reference massRescale :1412-1480 (Wilson-type branch)
void massRescale(ColorSpinorField &b, QudaInvertParam &param) {
  const double kappa = param.kappa;
  switch (param.solution_type) {
    case QUDA_MAT_SOLUTION:
      if (param.mass_normalization == QUDA_MASS_NORMALIZATION || param.mass_normalization == QUDA_ASYMMETRIC_MASS_NORMALIZATION) blas::ax(2.0 * kappa, b);
      break;
    case QUDA_MATDAG_MAT_SOLUTION:
      if (param.mass_normalization == QUDA_MASS_NORMALIZATION || param.mass_normalization == QUDA_ASYMMETRIC_MASS_NORMALIZATION) blas::ax(4.0 * kappa * kappa, b);
      break;
    case QUDA_MATPC_SOLUTION:
      if (param.mass_normalization == QUDA_MASS_NORMALIZATION) blas::ax(4.0 * kappa * kappa, b);
      else if (param.mass_normalization == QUDA_ASYMMETRIC_MASS_NORMALIZATION) blas::ax(2.0 * kappa, b);
      break;
    case QUDA_MATPCDAG_MATPC_SOLUTION:
      if (param.mass_normalization == QUDA_MASS_NORMALIZATION) blas::ax(16.0 * pow(kappa, 4), b);
      else if (param.mass_normalization == QUDA_ASYMMETRIC_MASS_NORMALIZATION) blas::ax(4.0 * kappa * kappa, b);
      break;
    default: errorQuda("Solution type %d not supported", param.solution_type);
  }
}

}  // namespace quda

extern "C" {

void invertQuda(void *hp_x, void *hp_b, QudaInvertParam *param) {
  if (!gaugePrecise) errorQuda("Gauge field not allocated");
  if (param->tune == QUDA_TUNE_YES || param->tune == QUDA_TUNE_NO) setTuning(param->tune);   // reference invertQuda: setTuning(param->tune)
  if (!cloverPrecise && param->dslash_type == QUDA_TWISTED_CLOVER_DSLASH) errorQuda("Clover field not allocated");
  const bool pc_solution = param->solution_type == QUDA_MATPC_SOLUTION || param->solution_type == QUDA_MATPCDAG_MATPC_SOLUTION;
  const bool pc_solve = param->solve_type == QUDA_DIRECT_PC_SOLVE || param->solve_type == QUDA_NORMOP_PC_SOLVE;
  const bool mat_solution = param->solution_type == QUDA_MAT_SOLUTION || param->solution_type == QUDA_MATPC_SOLUTION;
  const bool direct_solve = param->solve_type == QUDA_DIRECT_SOLVE || param->solve_type == QUDA_DIRECT_PC_SOLVE;
  if (pc_solution && !pc_solve) errorQuda("Preconditioned (PC) solution_type requires a PC solve_type");
  if (!mat_solution && !pc_solution && pc_solve) errorQuda("Unpreconditioned MATDAG_MAT solution_type requires an unpreconditioned solve_type");
  if (param->inv_type_precondition == QUDA_MG_INVERTER && (!direct_solve || !mat_solution)) errorQuda("Multigrid preconditioning only supported for direct solves");
  param->secs = 0; param->gflops = 0; param->iter = 0;
  const bool prof = getenv("QUDA_AMD_INVERT_PROFILE") != nullptr;
  auto stamp = [&](const char *what) {
    static double last = 0;
    if (!prof) return;
    HIP_CHECK(hipDeviceSynchronize());
    timeval tv; gettimeofday(&tv, nullptr);
    const double t = tv.tv_sec + 1e-6 * tv.tv_usec;
    if (what) printfQuda("invertQuda: %-12s %.3f ms\n", what, 1e3 * (t - last));
    last = t;
  };
  stamp(nullptr);

  // reference createDirac :1386-1410
  DiracParam dp, dpSloppy, dpPre;
  setDiracParam(dp, param, pc_solve);
  setDiracSloppyParam(dpSloppy, param, pc_solve);
  setDiracPreParam(dpPre, param, pc_solve);
  Dirac *d = Dirac::create(dp), *dSloppy = Dirac::create(dpSloppy), *dPre = Dirac::create(dpPre);
  Dirac &dirac = *d;
  stamp("operators");

  const LatticeGeom &geom = residentGeom();
  ColorSpinorParam cpuParam(hp_b, *param, geom.X, pc_solution);
  ColorSpinorField h_b(cpuParam);
  cpuParam.v = hp_x;
  ColorSpinorField h_x(cpuParam);
  ColorSpinorParam cp = deviceSpinorParam(param->cuda_prec, pc_solution ? QUDA_PARITY_SITE_SUBSET : QUDA_FULL_SITE_SUBSET, param->twist_flavor);
  cp.create = QUDA_ZERO_FIELD_CREATE;
  ColorSpinorField *b = new ColorSpinorField(cp), *x = new ColorSpinorField(cp);
  *b = h_b;
  if (param->use_init_guess == QUDA_USE_INIT_GUESS_YES) *x = h_x;
  const double nb = blas::norm2(*b);
  if (nb == 0.0) errorQuda("Source has zero norm");
  if (param->solver_normalization == QUDA_SOURCE_NORMALIZATION) { blas::ax(1.0 / sqrt(nb), *b); blas::ax(1.0 / sqrt(nb), *x); }
  massRescale(*b, *param);
  stamp("upload");

  ColorSpinorField *in = nullptr, *out = nullptr;
  dirac.prepare(in, out, *x, *b, param->solution_type);
  stamp("prepare");

  if (mat_solution && !direct_solve) {  // prepare source: b' = A^dag b
    ColorSpinorField tmp(*in);
    dirac.Mdag(*in, tmp);
  } else if (!mat_solution && direct_solve) {  // first of two solves: A^dag y = b
    DiracMdag m(dirac), mSloppy(*dSloppy), mPre(*dPre);
    SolverParam sp(*param);
    Solver *solve = Solver::create(sp, m, mSloppy, mPre);
    (*solve)(*out, *in);
    blas::copy(*in, *out);
    sp.updateInvertParam(*param);
    delete solve;
  }
  if (direct_solve) {
    DiracM m(dirac), mSloppy(*dSloppy), mPre(*dPre);
    SolverParam sp(*param);
    Solver *solve = Solver::create(sp, m, mSloppy, mPre);
    (*solve)(*out, *in);
    sp.updateInvertParam(*param);
    delete solve;
  } else {
    DiracMdagM m(dirac), mSloppy(*dSloppy), mPre(*dPre);
    SolverParam sp(*param);
    Solver *solve = Solver::create(sp, m, mSloppy, mPre);
    (*solve)(*out, *in);
    sp.updateInvertParam(*param);
    delete solve;
  }
  stamp("solve");
  dirac.reconstruct(*x, *b, param->solution_type);
  if (param->solver_normalization == QUDA_SOURCE_NORMALIZATION) blas::ax(sqrt(nb), *x);
  stamp("reconstruct");
  h_x = *x;
  stamp("download");
  delete b; delete x;
  delete d; delete dSloppy; delete dPre;
  stamp("free");
}

void *newMultigridQuda(QudaMultigridParam *mg_param) { return new multigrid_solver(*mg_param); }
void destroyMultigridQuda(void *mg) { delete static_cast<multigrid_solver *>(mg); }

// reference MG::verify (lib/multigrid.cpp:372-486): worst relative deviations of the three identities over all levels
void qudaAmdMultigridVerify(void *mg_instance, double dev[3]) { static_cast<multigrid_solver *>(mg_instance)->mg->verify(dev); }

// one application of the preconditioner (a V/K-cycle) to a host vector, for tests: x = K b
void qudaAmdMultigridCycle(void *mg_instance, void *h_x, void *h_b, QudaInvertParam *param) {
  multigrid_solver *mgs = static_cast<multigrid_solver *>(mg_instance);
  const LatticeGeom &geom = residentGeom();
  ColorSpinorParam cpuParam(h_b, *param, geom.X, false);
  ColorSpinorField hb(cpuParam);
  cpuParam.v = h_x;
  ColorSpinorField hx(cpuParam);
  ColorSpinorParam cp = deviceSpinorParam(QUDA_SINGLE_PRECISION, QUDA_FULL_SITE_SUBSET, param->twist_flavor);
  cp.create = QUDA_ZERO_FIELD_CREATE;
  ColorSpinorField b(cp), x(cp);
  b = hb;
  (*mgs->mg)(x, b);
  hx = x;
}

}
