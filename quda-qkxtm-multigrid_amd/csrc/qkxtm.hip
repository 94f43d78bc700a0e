// qkxtm.hip — the solve loop of the QKXTM correlator drivers behind the library: source preparation
// (point source, Gaussian smearing with the APE-smeared links), even-odd preconditioned GCR with the resident multigrid
// hierarchies of the two twist flavours, reconstruction, and the propagators handed back in the drivers' own layout.
//
// Reference: calcMG_threepTwop_EvenOdd (lib/interface_quda.cpp:6018-6531; the same loop opens calcMG_loop_wOneD_TSM_*,
// :7093, :8535), QKXTM_Vector_Kepler::gaussianSmearing (lib/qudaQKXTM_Vector_Kepler.cpp:386-421),
// lib/code_pieces_Kepler/Gauss_core_Kepler.h, uploadToCuda / downloadFromCuda (lib/qudaQKXTM_Kepler_kernels.cu:972-1056).
// Contractions, momentum projection and the HDF5 / ASCII writers that follow the loop in the reference are out of scope
// (SURVEY 2 row 20): the caller gets the propagators instead.
//
// Everything stays on the device between the point source and the finished propagator:
//  * smearing acts on colour only, so ONE smearing pass per source position serves all twelve spin-colour sources: the
//    point sources of the three colours sit in spin slots 0..2 of a single vector (the reference smears 24 times);
//  * a smearing step is six covariant shifts accumulated in place (applyCovariantShift: the ghost-aware single-direction
//    hop of the multigrid setup without the spin projector), so it runs unchanged on a grid-decomposed lattice;
//  * the QKXTM host layouts are lexicographic in the sites and UKQCD in spin; site reordering and the spin rotation to the
//    device basis happen in the copy kernels.
#include <cmath>
#include <cstring>
#include <vector>

#include "basis.h"
#include "blas.h"
#include "device_io.h"
#include "dslash.h"
#include "interface_internal.h"
#include "multigrid.h"
#include "p2p.h"
#include "quda_amd_ext.h"
#include "solver.h"

namespace quda {

void *stagingBuffer(size_t bytes);   // fields.hip

// ---- site order / basis: QKXTM host vector (lexicographic sites, UKQCD spin) <-> device full field (even-odd, DeGrand-Rossi) ----
__device__ __forceinline__ long lex_of(int idx, int parity, int Xh, int Y, int Z) {
  int l = idx / Xh;
  const int y = l % Y; l /= Y;
  const int z = l % Z, t = l / Z;
  return 2l * idx + ((y + z + t + parity) & 1);   // checkerboard index = lexicographic index / 2
}

__global__ void __launch_bounds__(256) lex_to_dev_kernel(double *dev, int stride, size_t parityDoubles, const double *lex, int Vh, int Xh, int Y, int Z, int change) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x, parity = blockIdx.y;
  if (idx >= Vh) return;
  const double *h = lex + lex_of(idx, parity, Xh, Y, Z) * 24;
  double r[24], q[24];
#pragma unroll
  for (int k = 0; k < 24; k++) r[k] = h[k];
  if (change) rotate_basis(q, r, change);
  Planar<double, 24>::store(change ? q : r, dev + parity * parityDoubles, stride, idx, nullptr, idx);
}

__global__ void __launch_bounds__(256) dev_to_lex_kernel(double *lex, const double *dev, int stride, size_t parityDoubles, int Vh, int Xh, int Y, int Z, int change,
                                                         double scale) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x, parity = blockIdx.y;
  if (idx >= Vh) return;
  double r[24], q[24];
  Planar<double, 24>::load(r, dev + parity * parityDoubles, stride, idx, nullptr, idx);
  if (change) rotate_basis(q, r, change);
  double *h = lex + lex_of(idx, parity, Xh, Y, Z) * 24;
#pragma unroll
  for (int k = 0; k < 24; k++) h[k] = scale * (change ? q[k] : r[k]);
}

// colour point sources in spin slots: psi(site)[slot c][colour c] = 1 for c = 0, 1, 2 (host-basis slots, rotated like any host vector)
__global__ void point_slots_kernel(double *dev, int stride, int idx) {
  double r[24], q[24];
  for (int k = 0; k < 24; k++) r[k] = 0;
  for (int c = 0; c < 3; c++) r[6 * c + 2 * c] = 1.0;
  rotate_basis(q, r, BASIS_UKQCD_TO_DR);
  Planar<double, 24>::store(q, dev, stride, idx, nullptr, idx);
}

// out = e_{spin s0} (x) [colour vector held in spin slot `slot` of in], both in the device basis of host-basis slots
__global__ void __launch_bounds__(256) spread_slot_kernel(double *out, const double *in, int stride, size_t parityDoubles, int Vh, int s0, int slot) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x, parity = blockIdx.y;
  if (idx >= Vh) return;
  double r[24], q[24];
  Planar<double, 24>::load(r, in + parity * parityDoubles, stride, idx, nullptr, idx);
  rotate_basis(q, r, BASIS_DR_TO_UKQCD);
#pragma unroll
  for (int k = 0; k < 24; k++) r[k] = 0;
#pragma unroll
  for (int k = 0; k < 6; k++) r[6 * s0 + k] = q[6 * slot + k];
  rotate_basis(q, r, BASIS_UKQCD_TO_DR);
  Planar<double, 24>::store(q, out + parity * parityDoubles, stride, idx, nullptr, idx);
}

static size_t parityDoubles(const ColorSpinorField &f) { return (size_t)((const char *)f.Odd().V() - (const char *)f.Even().V()) / sizeof(double); }

static void checkFullDouble(const ColorSpinorField &f) {
  if (f.Location() != QUDA_CUDA_FIELD_LOCATION || f.Precision() != QUDA_DOUBLE_PRECISION || f.SiteSubset() != QUDA_FULL_SITE_SUBSET || f.Nspin() != 4 || f.Ncolor() != 3)
    errorQuda("expected a full fp64 device spinor");
}

static void lexToDevice(ColorSpinorField &dst, const double *h_lex, const LatticeGeom &g, bool ukqcd) {
  checkFullDouble(dst);
  const size_t bytes = (size_t)g.V * 24 * sizeof(double);
  double *stage = (double *)stagingBuffer(bytes);
  HIP_CHECK(hipMemcpyAsync(stage, h_lex, bytes, hipMemcpyHostToDevice, computeStream()));
  hipLaunchKernelGGL(lex_to_dev_kernel, dim3((g.Vh + 255) / 256, 2), dim3(256), 0, computeStream(), (double *)dst.V(), dst.Stride(), parityDoubles(dst), stage, g.Vh, g.Xh,
                     g.X[1], g.X[2], ukqcd ? BASIS_UKQCD_TO_DR : BASIS_NONE);
  HIP_CHECK(hipGetLastError());
  HIP_CHECK(hipStreamSynchronize(computeStream()));
}

static void deviceToLex(double *h_lex, const ColorSpinorField &src, const LatticeGeom &g, bool ukqcd, double scale) {
  checkFullDouble(src);
  const size_t bytes = (size_t)g.V * 24 * sizeof(double);
  double *stage = (double *)stagingBuffer(bytes);
  hipLaunchKernelGGL(dev_to_lex_kernel, dim3((g.Vh + 255) / 256, 2), dim3(256), 0, computeStream(), stage, (const double *)src.V(), src.Stride(), parityDoubles(src), g.Vh, g.Xh,
                     g.X[1], g.X[2], ukqcd ? BASIS_DR_TO_UKQCD : BASIS_NONE, scale);
  HIP_CHECK(hipGetLastError());
  HIP_CHECK(hipMemcpyAsync(h_lex, stage, bytes, hipMemcpyDeviceToHost, computeStream()));
  HIP_CHECK(hipStreamSynchronize(computeStream()));
  p2pCheck("deviceToLex");
}

// ---- the smearing links: QKXTM host layout gauge[dir][lexicographic site][3][3][2] (lib/qudaQKXTM_Gauge_Kepler.cpp:73-89) ----
static GaugeField *loadLexGauge(void **gauge_lex, const LatticeGeom &g) {
  // reorder to the QDP host order (even sites then odd) the loader takes; host loop, once per call
  std::vector<std::vector<double>> eo(4, std::vector<double>((size_t)g.V * 18));
  void *ptr[4];
  for (int d = 0; d < 4; d++) {
    const double *src = (const double *)gauge_lex[d];
    if (!src) errorQuda("gauge_APE[%d] is NULL", d);
    for (long iv = 0; iv < g.V; iv++) {
      long l = iv / g.X[0];
      const int x = (int)(iv % g.X[0]), y = (int)(l % g.X[1]); l /= g.X[1];
      const int z = (int)(l % g.X[2]), t = (int)(l / g.X[2]);
      const int parity = (x + y + z + t) & 1;
      memcpy(&eo[d][((size_t)parity * g.Vh + iv / 2) * 18], src + iv * 18, 18 * sizeof(double));
    }
    ptr[d] = eo[d].data();
  }
  GaugeField *U = new GaugeField(g, QUDA_DOUBLE_PRECISION, QUDA_RECONSTRUCT_NO, QUDA_PERIODIC_T, 1.0);
  loadGaugeWithHalo(*U, ptr, QUDA_DOUBLE_PRECISION);
  return U;
}

// v <- smear^n(v):  psi' = (psi + alpha sum_{i<3} [U_i(x) psi(x+i) + U_i(x-i)^dag psi(x-i)]) / (1 + 6 alpha)   (Gauss_core_Kepler.h)
static void gaussianSmear(ColorSpinorField &v, const GaugeField &U, double alpha, int nsmear) {
  checkFullDouble(v);
  ColorSpinorField tmp(v);
  ColorSpinorField *src = &v, *dst = &tmp;
  const double normalize = 1.0 / (1.0 + 6.0 * alpha);
  for (int it = 0; it < nsmear; it++) {
    for (int parity = 0; parity < 2; parity++) {
      ColorSpinorField &o = parity ? dst->Odd() : dst->Even();
      const ColorSpinorField &same = parity ? src->Odd() : src->Even();
      const ColorSpinorField &other = parity ? src->Even() : src->Odd();
      for (int dir = 0; dir < 6; dir++)
        applyCovariantShift(o, other, U, parity, dir, alpha * normalize, dir == 0 ? &same : &o, dir == 0 ? normalize : 1.0);
    }
    std::swap(src, dst);
  }
  if (src != &v) blas::copy(v, *src);
}

static void checkCalcParam(const QudaInvertParam *param, const char *fname) {
  // reference :6041-6054
  if (param->solve_type != QUDA_DIRECT_PC_SOLVE) errorQuda("%s: This function works only with Direct solve and even odd preconditioning", fname);
  if (param->inv_type != QUDA_GCR_INVERTER) errorQuda("%s: This function works only with GCR method", fname);
  if (param->gamma_basis != QUDA_UKQCD_GAMMA_BASIS) errorQuda("%s: This function works only with ukqcd gamma basis", fname);
  if (param->dirac_order != QUDA_DIRAC_ORDER) errorQuda("%s: This function works only with colors inside the spins", fname);
  if (param->matpc_type != QUDA_MATPC_EVEN_EVEN && param->matpc_type != QUDA_MATPC_ODD_ODD) errorQuda("%s: matpc_type %d not supported (symmetric even-even / odd-odd only)", fname, param->matpc_type);
  if (param->solution_type != QUDA_MAT_SOLUTION) errorQuda("%s: solution_type %d not supported (the drivers ask for QUDA_MAT_SOLUTION)", fname, param->solution_type);
  if (param->dslash_type != QUDA_TWISTED_MASS_DSLASH && param->dslash_type != QUDA_TWISTED_CLOVER_DSLASH) errorQuda("%s: twisted-mass / twisted-clover operators only", fname);
  if (param->inv_type_precondition == QUDA_MG_INVERTER && (!param->preconditionerUP || !param->preconditionerDN))
    errorQuda("%s: preconditionerUP / preconditionerDN not set (one multigrid hierarchy per twist flavour)", fname);
}

}  // namespace quda

using namespace quda;

extern "C" {

void qudaAmdGaussianSmear(void *h_out, const void *h_in, void **gauge_APE, int nsmear, double alpha) {
  if (!gaugePrecise) errorQuda("Gauge field not allocated");   // the lattice geometry comes from the resident field, as in the reference
  if (nsmear < 0) errorQuda("nsmear = %d", nsmear);
  const LatticeGeom &g = residentGeom();
  if (!gauge_APE && !gaugeSmeared) errorQuda("qudaAmdGaussianSmear: gauge_APE is NULL and no smeared field is resident (performAPEnStep)");
  GaugeField *U = gauge_APE ? loadLexGauge(gauge_APE, g) : gaugeSmeared;
  ColorSpinorParam cp = deviceSpinorParam(QUDA_DOUBLE_PRECISION, QUDA_FULL_SITE_SUBSET, QUDA_TWIST_NO);
  cp.create = QUDA_ZERO_FIELD_CREATE;
  ColorSpinorField v(cp);
  lexToDevice(v, (const double *)h_in, g, true);
  gaussianSmear(v, *U, alpha, nsmear);
  deviceToLex((double *)h_out, v, g, true, 1.0);
  if (gauge_APE) delete U;
}

}  // extern "C"

namespace quda {
// the solve loop itself; every finished propagator (isc = spin * 3 + colour of the source, flavour +1 up / -1 down) is handed to
// `each` as a lexicographic UKQCD host vector of V * 24 doubles that is only valid during the call
void calcMGPropagatorsEach(void **gauge_APE, QudaInvertParam *param, const QudaAmdSourceParam *src, const char *fname,
                           void (*each)(void *ctx, int isc, int flavor, const double *h_prop, size_t nreal), void *ctx) {
  if (!gaugePrecise) errorQuda("%s: Gauge field not allocated", fname);
  if (!cloverPrecise && param->dslash_type == QUDA_TWISTED_CLOVER_DSLASH) errorQuda("%s: Clover field not allocated", fname);
  if (!src) errorQuda("%s: source description is NULL", fname);
  checkCalcParam(param, fname);
  const bool flag_eo = param->matpc_type == QUDA_MATPC_EVEN_EVEN;
  const LatticeGeom &g = residentGeom();
  const CommGrid &cg = commGrid();
  for (int d = 0; d < 4; d++)
    if (src->sourcePosition[d] < 0 || src->sourcePosition[d] >= g.X[d] * cg.dims[d]) errorQuda("%s: source position %d out of range in dimension %d", fname, src->sourcePosition[d], d);
  if (src->nsmearGauss < 0) errorQuda("%s: nsmearGauss = %d", fname, src->nsmearGauss);
  param->secs = 0; param->gflops = 0; param->iter = 0;

  if (src->nsmearGauss > 0 && !gauge_APE && !gaugeSmeared) errorQuda("%s: gauge_APE is NULL and no smeared field is resident (performAPEnStep)", fname);
  GaugeField *Uape = (src->nsmearGauss > 0) ? (gauge_APE ? loadLexGauge(gauge_APE, g) : gaugeSmeared) : nullptr;

  // reference createDirac :6291-6300, once, before the loop: the twist flavour travels with the fields
  const bool pc_solve = true;
  DiracParam dp, dpSloppy, dpPre;
  setDiracParam(dp, param, pc_solve);
  setDiracSloppyParam(dpSloppy, param, pc_solve);
  setDiracPreParam(dpPre, param, pc_solve);
  Dirac *d = Dirac::create(dp), *dSloppy = Dirac::create(dpSloppy), *dPre = Dirac::create(dpPre);
  Dirac &dirac = *d;
  DiracM m(dirac), mSloppy(*dSloppy), mPre(*dPre);

  ColorSpinorParam cp64 = deviceSpinorParam(QUDA_DOUBLE_PRECISION, QUDA_FULL_SITE_SUBSET, QUDA_TWIST_PLUS);
  cp64.create = QUDA_ZERO_FIELD_CREATE;
  ColorSpinorField phi(cp64), source(cp64), result(cp64);
  ColorSpinorParam cp = deviceSpinorParam(param->cuda_prec, QUDA_FULL_SITE_SUBSET, QUDA_TWIST_PLUS);
  cp.create = QUDA_ZERO_FIELD_CREATE;
  ColorSpinorField *b = new ColorSpinorField(cp), *x = new ColorSpinorField(cp);

  // the three colour point sources in spin slots 0..2, on the rank that owns the site (reference :6404-6421), smeared once
  int my_src[4];
  bool mine = true;
  for (int k = 0; k < 4; k++) { my_src[k] = src->sourcePosition[k] - cg.coords[k] * g.X[k]; mine = mine && my_src[k] >= 0 && my_src[k] < g.X[k]; }
  if (mine) {
    const long iv = (((long)my_src[3] * g.X[2] + my_src[2]) * g.X[1] + my_src[1]) * g.X[0] + my_src[0];
    const int parity = (my_src[0] + my_src[1] + my_src[2] + my_src[3]) & 1;
    ColorSpinorField &half = parity ? phi.Odd() : phi.Even();
    hipLaunchKernelGGL(point_slots_kernel, dim3(1), dim3(1), 0, computeStream(), (double *)half.V(), half.Stride(), (int)(iv / 2));
    HIP_CHECK(hipGetLastError());
  }
  if (Uape) gaussianSmear(phi, *Uape, src->alphaGauss, src->nsmearGauss);

  const size_t vec = (size_t)g.V * 24;
  std::vector<double> h_one(vec);
  double secs = 0, gflops = 0;
  int iters = 0;
  const bool rescale = param->mass_normalization == QUDA_MASS_NORMALIZATION || param->mass_normalization == QUDA_ASYMMETRIC_MASS_NORMALIZATION;
  // The twelve spin-colour sources of a flavour through ONE lockstep solve (block_solver.cpp; DESIGN 3 "Several sources in lockstep") when the
  // solver is multigrid-preconditioned GCR without an initial guess and the fields fit: the hierarchy's cycle runs once per iteration for all of
  // them.  QUDA_AMD_QKXTM_LOCKSTEP=0 keeps the reference's order of 24 separate solves.
  bool lockstep = param->inv_type == QUDA_GCR_INVERTER && param->inv_type_precondition == QUDA_MG_INVERTER && param->preconditionerUP && param->preconditionerDN &&
                  param->use_init_guess != QUDA_USE_INIT_GUESS_YES;
  {
    static int env = -1;
    if (env < 0) { const char *e = getenv("QUDA_AMD_QKXTM_LOCKSTEP"); env = e ? atoi(e) : 1; }
    if (!env) lockstep = false;
    if (lockstep) {
      size_t freeB = 0, totalB = 0;
      HIP_CHECK(hipMemGetInfo(&freeB, &totalB));
      const double full = (double)g.V * 24 * (int)param->cuda_prec, parS = 0.5 * (double)g.V * 24 * (int)param->cuda_prec_sloppy;
      const double need = 12.0 * (2.0 * full + 1.0 * full + (2.0 + 2.0 * param->gcrNkrylov) * parS);
      if (need > 0.6 * (double)freeB) lockstep = false;
    }
  }
  if (lockstep) {
    std::vector<ColorSpinorField *> bs(12), xs(12), ins(12), outs(12);
    for (int isc = 0; isc < 12; isc++) { bs[isc] = new ColorSpinorField(cp); xs[isc] = new ColorSpinorField(cp); }
    for (int fl = 0; fl < 2; fl++) {
      const QudaTwistFlavorType flavor = fl == 0 ? QUDA_TWIST_PLUS : QUDA_TWIST_MINUS;   // up, then down (:6401, :6470)
      param->twist_flavor = flavor;
      param->preconditioner = fl == 0 ? param->preconditionerUP : param->preconditionerDN;
      for (int isc = 0; isc < 12; isc++) {
        hipLaunchKernelGGL(spread_slot_kernel, dim3((g.Vh + 255) / 256, 2), dim3(256), 0, computeStream(), (double *)source.V(), (const double *)phi.V(), source.Stride(),
                           parityDoubles(source), g.Vh, isc / 3, isc % 3);
        HIP_CHECK(hipGetLastError());
        bs[isc]->changeTwist(flavor); xs[isc]->changeTwist(flavor);
        *bs[isc] = source;
        blas::zero(*xs[isc]);
        dirac.prepare(ins[isc], outs[isc], *xs[isc], *bs[isc], param->solution_type);
        ins[isc]->changeTwist(flavor); outs[isc]->changeTwist(flavor);
      }
      param->secs = 0; param->gflops = 0; param->iter = 0;
      SolverParam sp(*param);
      MG *K = static_cast<multigrid_solver *>(param->preconditioner)->mg;
      const MultiSrcSolve res = solveMultiSrcGCR(outs, ins, m, mSloppy, K, sp, dSloppy);
      secs += res.secs; iters += 12 * res.iter;
      double worst = 0;
      for (int isc = 0; isc < 12; isc++) worst = std::max(worst, sqrt(res.r2[isc] / res.b2[isc]));
      param->true_res = worst;
      for (int isc = 0; isc < 12; isc++) {
        dirac.reconstruct(*xs[isc], *bs[isc], param->solution_type);
        result = *xs[isc];
        deviceToLex(h_one.data(), result, g, true, rescale ? 2.0 * param->kappa : 1.0);
        each(ctx, isc, fl == 0 ? +1 : -1, h_one.data(), vec);
      }
    }
    for (int isc = 0; isc < 12; isc++) { delete bs[isc]; delete xs[isc]; }
  }
  for (int isc = 0; isc < 12 && !lockstep; isc++) {
    hipLaunchKernelGGL(spread_slot_kernel, dim3((g.Vh + 255) / 256, 2), dim3(256), 0, computeStream(), (double *)source.V(), (const double *)phi.V(), source.Stride(),
                       parityDoubles(source), g.Vh, isc / 3, isc % 3);
    HIP_CHECK(hipGetLastError());
    for (int fl = 0; fl < 2; fl++) {
      const QudaTwistFlavorType flavor = fl == 0 ? QUDA_TWIST_PLUS : QUDA_TWIST_MINUS;   // up, then down (:6401, :6470)
      param->twist_flavor = flavor;
      b->changeTwist(flavor); x->changeTwist(flavor);
      *b = source;
      ColorSpinorField *in = nullptr, *out = nullptr;
      dirac.prepare(in, out, *x, *b, param->solution_type);
      param->preconditioner = fl == 0 ? param->preconditionerUP : param->preconditionerDN;
      // counters: SolverParam starts from the values in *param and updateInvertParam adds them back (reference
      // include/invert_quda.h:262-300), which in the reference's loop doubles the running totals at every solve; summed
      // properly here
      param->secs = 0; param->gflops = 0; param->iter = 0;
      SolverParam sp(*param);
      Solver *solve = Solver::create(sp, m, mSloppy, mPre);
      // the smeared source on the solve parity is the initial guess (:6442-6445; used when use_init_guess says so)
      *out = flag_eo ? source.Even() : source.Odd();
      out->changeTwist(flavor);
      (*solve)(*out, *in);
      sp.updateInvertParam(*param);
      secs += param->secs; gflops += param->gflops; iters += param->iter;
      dirac.reconstruct(*x, *b, param->solution_type);
      delete solve;
      result = *x;
      deviceToLex(h_one.data(), result, g, true, rescale ? 2.0 * param->kappa : 1.0);
      each(ctx, isc, fl == 0 ? +1 : -1, h_one.data(), vec);
    }
  }
  param->secs = secs; param->gflops = gflops; param->iter = iters;
  delete b; delete x;
  delete d; delete dSloppy; delete dPre;
  if (gauge_APE) delete Uape;
}
}  // namespace quda

extern "C" {

void qudaAmdCalcMGPropagators(void *h_prop_up, void *h_prop_dn, void **gauge_APE, QudaInvertParam *param, const QudaAmdSourceParam *src) {
  if (!h_prop_up || !h_prop_dn) errorQuda("qudaAmdCalcMGPropagators: propagator buffers are NULL");
  struct Out { double *up, *dn; } out = {(double *)h_prop_up, (double *)h_prop_dn};
  calcMGPropagatorsEach(gauge_APE, param, src, "qudaAmdCalcMGPropagators",
                        [](void *c, int isc, int flavor, const double *h, size_t n) { Out *o = (Out *)c; memcpy((flavor > 0 ? o->up : o->dn) + (size_t)isc * n, h, n * sizeof(double)); }, &out);
}

}

// ================================================================================================
// the reference's entry-point names (include/qudaQKXTM_Kepler.h): solve loops only, solutions go to the registered sink
// ================================================================================================
#include "qudaQKXTM_Kepler.h"

namespace quda {

static QudaAmdSolutionSink g_sink = nullptr;
static void *g_sinkCtx = nullptr;

static void toSink(const char *kind, int index, int flavor, const double *h_source, const double *h_solution, size_t nreal) {
  if (g_sink) { g_sink(g_sinkCtx, kind, index, flavor, h_source, h_solution, nreal); return; }
  double n2 = 0;
  for (size_t i = 0; i < nreal; i++) n2 += h_solution[i] * h_solution[i];
  comm_allreduce(&n2, 1);
  printfQuda("%s %d (flavour %+d): |solution|^2 = %.12e (no sink registered: dropped)\n", kind, index, flavor, n2);
}

// Z4 noise on every spin-colour component of every local site (reference getStochasticRandomSource<Float>,
// lib/qudaQKXTM_Kepler_utils.cpp:149-180: 1, -1, i, -i with equal probability; UNITY: all ones).  The reference draws from GSL's
// ranlux seeded with seed + rank * seed; GSL is not a dependency here: a counter-based generator keyed by (seed, rank, source
// number, component) — the same stream for a given decomposition, independent of how many sources came before.
static void stochasticSource(double *h, size_t ncomplex, unsigned long seed, int isrc, SOURCE_T type) {
  const unsigned long long key = (unsigned long long)seed * 0x9E3779B97F4A7C15ull + (unsigned long long)commGrid().rank * 0xD1B54A32D192ED03ull + (unsigned long long)isrc * 0x94D049BB133111EBull;
  for (size_t i = 0; i < ncomplex; i++) {
    if (type == UNITY) { h[2 * i] = 1.0; h[2 * i + 1] = 0.0; continue; }
    unsigned long long z = key + (unsigned long long)i * 0xBF58476D1CE4E5B9ull;
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 27; z *= 0x94D049BB133111EBull; z ^= z >> 31;
    const int r = (int)(z >> 62);
    h[2 * i] = r == 0 ? 1.0 : (r == 1 ? -1.0 : 0.0);
    h[2 * i + 1] = r == 2 ? 1.0 : (r == 3 ? -1.0 : 0.0);
  }
}

// the loop of calcMG_loop_wOneD_TSM_EvenOdd (reference lib/interface_quda.cpp:8535-9230) without its contractions
static void loopSolves(QudaInvertParam *param, const qudaQKXTM_loopInfo &loopInfo, const qudaQKXTMinfo_Kepler &info, const char *fname) {
  if (!gaugePrecise) errorQuda("%s: Gauge field not allocated", fname);
  if (!cloverPrecise && param->dslash_type == QUDA_TWISTED_CLOVER_DSLASH) errorQuda("%s: Clover field not allocated", fname);
  if (param->solve_type != QUDA_DIRECT_PC_SOLVE) errorQuda("%s: This function works only with Direct solve and even odd preconditioning", fname);
  if (param->gamma_basis != QUDA_UKQCD_GAMMA_BASIS) errorQuda("%s: This function works only with ukqcd gamma basis", fname);
  if (param->dirac_order != QUDA_DIRAC_ORDER) errorQuda("%s: This function works only with colors inside the spins", fname);
  if (param->solution_type != QUDA_MAT_SOLUTION) errorQuda("%s: solution_type %d not supported (QUDA_MAT_SOLUTION)", fname, param->solution_type);
  if (info.isEven && param->matpc_type != QUDA_MATPC_EVEN_EVEN) errorQuda("%s: Inconsistency between operator types!", fname);        // :8560
  if (!info.isEven && param->matpc_type != QUDA_MATPC_ODD_ODD) errorQuda("%s: Inconsistency between operator types!", fname);

  // truncated-solver-method stopping criterion (:8595-8625): iterations if given, else tolerance; both given -> iterations
  const bool useTSM = loopInfo.useTSM;
  long TSM_maxiter = 0;
  double TSM_tol = 0.0;
  if (loopInfo.TSM_tol == 0 && loopInfo.TSM_maxiter != 0) TSM_maxiter = loopInfo.TSM_maxiter;
  else if (loopInfo.TSM_tol != 0 && loopInfo.TSM_maxiter == 0) TSM_tol = loopInfo.TSM_tol;
  else if (useTSM && loopInfo.TSM_tol != 0 && loopInfo.TSM_maxiter != 0) {
    warningQuda("Both max-iter = %ld and tolerance = %lf defined as criterions for the TSM. Proceeding with max-iter = %ld criterion.", loopInfo.TSM_maxiter, loopInfo.TSM_tol, loopInfo.TSM_maxiter);
    TSM_maxiter = loopInfo.TSM_maxiter;
  } else if (useTSM) errorQuda("%s: the truncated solver method needs TSM_tol or TSM_maxiter", fname);

  const LatticeGeom &g = residentGeom();
  const size_t vec = (size_t)g.V * 24;
  param->secs = 0; param->gflops = 0; param->iter = 0;
  const bool pc_solve = true;
  DiracParam dp, dpSloppy, dpPre;
  setDiracParam(dp, param, pc_solve);
  setDiracSloppyParam(dpSloppy, param, pc_solve);
  setDiracPreParam(dpPre, param, pc_solve);
  Dirac *d = Dirac::create(dp), *dSloppy = Dirac::create(dpSloppy), *dPre = Dirac::create(dpPre);
  Dirac &dirac = *d;
  DiracM m(dirac), mSloppy(*dSloppy), mPre(*dPre);
  ColorSpinorParam cp64 = deviceSpinorParam(QUDA_DOUBLE_PRECISION, QUDA_FULL_SITE_SUBSET, param->twist_flavor);
  cp64.create = QUDA_ZERO_FIELD_CREATE;
  ColorSpinorField stage(cp64);
  ColorSpinorParam cp = deviceSpinorParam(param->cuda_prec, QUDA_FULL_SITE_SUBSET, param->twist_flavor);
  cp.create = QUDA_ZERO_FIELD_CREATE;
  ColorSpinorField b(cp), x(cp);
  std::vector<double> h_src(vec), h_sol(vec);
  const bool rescale = param->mass_normalization == QUDA_MASS_NORMALIZATION || param->mass_normalization == QUDA_ASYMMETRIC_MASS_NORMALIZATION;
  double secs = 0, gflops = 0;
  int iters = 0;

  auto solve = [&](bool lowPrecision, const char *kind, int index) {
    lexToDevice(stage, h_src.data(), g, true);
    b = stage;
    blas::zero(x);
    ColorSpinorField *in = nullptr, *out = nullptr;
    dirac.prepare(in, out, x, b, param->solution_type);
    const double tol0 = param->tol;
    const int maxiter0 = param->maxiter;
    if (lowPrecision) { if (TSM_maxiter == 0) param->tol = TSM_tol; else param->maxiter = (int)TSM_maxiter; }   // :9025-9030
    param->secs = 0; param->gflops = 0; param->iter = 0;
    SolverParam sp(*param);
    Solver *s = Solver::create(sp, m, mSloppy, mPre);
    (*s)(*out, *in);
    sp.updateInvertParam(*param);
    delete s;
    param->tol = tol0; param->maxiter = maxiter0;
    secs += param->secs; gflops += param->gflops; iters += param->iter;
    dirac.reconstruct(x, b, param->solution_type);
    stage = x;
    deviceToLex(h_sol.data(), stage, g, true, rescale ? 2.0 * param->kappa : 1.0);
    toSink(kind, index, (int)param->twist_flavor, h_src.data(), h_sol.data(), vec);
  };

  // Groups of stochastic sources through ONE lockstep solve (block_solver.cpp) when the solver is multigrid-preconditioned GCR without an initial
  // guess: the hierarchy's cycle runs once per iteration for the whole group.  The sources of a group are drawn exactly as one by one (the
  // generator is keyed by the source number); QUDA_AMD_QKXTM_LOCKSTEP=0 keeps the reference's one-by-one order.
  int group = 1;
  if (param->inv_type == QUDA_GCR_INVERTER && param->inv_type_precondition == QUDA_MG_INVERTER && param->preconditioner && param->use_init_guess != QUDA_USE_INIT_GUESS_YES) {
    static int env = -1;
    if (env < 0) { const char *e = getenv("QUDA_AMD_QKXTM_LOCKSTEP"); env = e ? atoi(e) : 1; }
    if (env) {
      size_t freeB = 0, totalB = 0;
      HIP_CHECK(hipMemGetInfo(&freeB, &totalB));
      const double full = (double)g.V * 24 * (int)param->cuda_prec, parS = 0.5 * (double)g.V * 24 * (int)param->cuda_prec_sloppy;
      const double perSource = 3.0 * full + (2.0 + 2.0 * param->gcrNkrylov) * parS;
      group = 12;
      while (group > 1 && group * perSource > 0.6 * (double)freeB) group /= 2;
    }
  }
  auto solveGroup = [&](bool lowPrecision, const char *kind, int first, int n) {
    std::vector<std::vector<double>> h_srcs(n, std::vector<double>(vec));
    std::vector<ColorSpinorField *> bs(n), xs(n), ins(n), outs(n);
    for (int j = 0; j < n; j++) {
      stochasticSource(h_srcs[j].data(), (size_t)g.V * 12, loopInfo.seed, first + j, info.source_type);
      lexToDevice(stage, h_srcs[j].data(), g, true);
      bs[j] = new ColorSpinorField(cp); xs[j] = new ColorSpinorField(cp);
      *bs[j] = stage;
      blas::zero(*xs[j]);
      dirac.prepare(ins[j], outs[j], *xs[j], *bs[j], param->solution_type);
      ins[j]->changeTwist(param->twist_flavor); outs[j]->changeTwist(param->twist_flavor);
    }
    const double tol0 = param->tol;
    const int maxiter0 = param->maxiter;
    if (lowPrecision) { if (TSM_maxiter == 0) param->tol = TSM_tol; else param->maxiter = (int)TSM_maxiter; }
    param->secs = 0; param->gflops = 0; param->iter = 0;
    SolverParam sp(*param);
    MG *K = static_cast<multigrid_solver *>(param->preconditioner)->mg;
    const MultiSrcSolve res = solveMultiSrcGCR(outs, ins, m, mSloppy, K, sp, dSloppy);
    param->tol = tol0; param->maxiter = maxiter0;
    secs += res.secs; iters += n * res.iter;
    for (int j = 0; j < n; j++) {
      dirac.reconstruct(*xs[j], *bs[j], param->solution_type);
      stage = *xs[j];
      deviceToLex(h_sol.data(), stage, g, true, rescale ? 2.0 * param->kappa : 1.0);
      toSink(kind, first + j, (int)param->twist_flavor, h_srcs[j].data(), h_sol.data(), vec);
      delete bs[j]; delete xs[j];
    }
  };

  // production sources: low-precision solves under the truncated solver method, full solves otherwise (:9000-9050)
  const int Nrun = useTSM ? loopInfo.TSM_NLP : loopInfo.Nstoch;
  for (int is = 0; is < Nrun;) {
    const int n = std::min(group, Nrun - is);
    if (n >= 2) { solveGroup(useTSM, useTSM ? "loop_LP" : "loop_stoch", is, n); is += n; continue; }
    stochasticSource(h_src.data(), (size_t)g.V * 12, loopInfo.seed, is, info.source_type);
    solve(useTSM, useTSM ? "loop_LP" : "loop_stoch", is);
    is++;
  }
  // bias correction of the truncated solver method: TSM_NHP fresh sources solved to both precisions (:9170-9230)
  if (useTSM)
    for (int is = 0; is < loopInfo.TSM_NHP; is++) {
      stochasticSource(h_src.data(), (size_t)g.V * 12, loopInfo.seed, Nrun + is, info.source_type);
      solve(false, "loop_HP", is);
      solve(true, "loop_HP_LP", is);
    }
  param->secs = secs; param->gflops = gflops; param->iter = iters;
  delete d; delete dSloppy; delete dPre;
}

}  // namespace quda

extern "C" void qudaAmdSetSolutionSink(QudaAmdSolutionSink sink, void *ctx) { quda::g_sink = sink; quda::g_sinkCtx = ctx; }

void calcMG_threepTwop_EvenOdd(void **gaugeSmeared, void **gauge, QudaGaugeParam *gauge_param, QudaInvertParam *param, quda::qudaQKXTMinfo_Kepler info,
                               char *filename_twop, char *filename_threep, quda::WHICHPARTICLE NUCLEON) {
  (void)gauge; (void)gauge_param; (void)filename_twop; (void)filename_threep; (void)NUCLEON;   // consumed by the contractions / writers only
  if (info.Nsources < 0 || info.Nsources > MAX_NSOURCES) errorQuda("calcMG_threepTwop_EvenOdd: Nsources = %d", info.Nsources);
  double secs = 0, gflops = 0;
  int iters = 0;
  for (int isource = 0; isource < info.Nsources; isource++) {
    QudaAmdSourceParam src;
    for (int k = 0; k < 4; k++) src.sourcePosition[k] = info.sourcePosition[isource][k];
    src.nsmearGauss = info.nsmearGauss;
    src.alphaGauss = info.alphaGauss;
    int base = 12 * isource;
    quda::calcMGPropagatorsEach(gaugeSmeared, param, &src, "calcMG_threepTwop_EvenOdd",
                                [](void *c, int isc, int flavor, const double *h, size_t n) { quda::toSink(flavor > 0 ? "prop_up" : "prop_dn", *(int *)c + isc, flavor, nullptr, h, n); }, &base);
    secs += param->secs; gflops += param->gflops; iters += param->iter;
  }
  param->secs = secs; param->gflops = gflops; param->iter = iters;
}

void calcMG_loop_wOneD_TSM_EvenOdd(void **gaugeToPlaquette, QudaInvertParam *param, QudaGaugeParam *gauge_param, quda::qudaQKXTM_loopInfo loopInfo,
                                   quda::qudaQKXTMinfo_Kepler info) {
  (void)gaugeToPlaquette; (void)gauge_param;   // plaquette check and covariant derivatives of the contraction stage
  quda::loopSolves(param, loopInfo, info, "calcMG_loop_wOneD_TSM_EvenOdd");
}

void calcMG_loop_wOneD_TSM_wExact(void **gaugeToPlaquette, QudaInvertParam *EVparam, QudaInvertParam *param, QudaGaugeParam *gauge_param,
                                  quda::qudaQKXTM_arpackInfo arpackInfo, quda::qudaQKXTM_loopInfo loopInfo, quda::qudaQKXTMinfo_Kepler info) {
  (void)gaugeToPlaquette; (void)gauge_param; (void)EVparam;
  if (arpackInfo.nEv != 0) errorQuda("calcMG_loop_wOneD_TSM_wExact: exact deflation with %d eigenvectors needs ARPACK, which this library does not link; nEv = 0 runs the undeflated loop", arpackInfo.nEv);
  quda::loopSolves(param, loopInfo, info, "calcMG_loop_wOneD_TSM_wExact");
}
