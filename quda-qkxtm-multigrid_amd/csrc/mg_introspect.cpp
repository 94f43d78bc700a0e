// mg_introspect.cpp — read-only access to the pieces of a multigrid hierarchy for parity tests and FFI callers: what a
// C++ test of the reference reaches through multigrid_solver->mg (null vectors B, the packed block-orthonormal V,
// the coarse link fields Y / X, and R, P, M of every level; reference include/multigrid.h:108-330, include/transfer.h).
// Host layouts are the reference's CPU orders: vectors site-major (parity*Vh + x_cb, spin, colour, re/im) fp32
// (QUDA_SPACE_SPIN_COLOR_FIELD_ORDER, DeGrand-Rossi basis on level 0); V as (site, spin, colour, vector);
// coarse links as QDP-ordered Y[dim 0-3 backward | 4-7 forward][site][row][col] with the operator's -kappa folded in.
#include <hip/hip_runtime.h>

#include <cstring>
#include <vector>

#include "interface_internal.h"
#include "block.h"
#include "coarse_cycle.h"
#include "dslash.h"
#include "multigrid.h"
#include "quda_amd_ext.h"

using namespace quda;

static MG *levelOf(void *mg_instance, int level) {
  if (!mg_instance) errorQuda("null multigrid handle");
  MG *m = static_cast<multigrid_solver *>(mg_instance)->mg;
  for (int l = 0; l < level && m; l++) m = m->getCoarse();
  if (!m) errorQuda("multigrid level %d does not exist", level);
  return m;
}

static ColorSpinorField hostView(const ColorSpinorField &dev, void *ptr) {
  ColorSpinorParam p = dev.param();
  p.location = QUDA_CPU_FIELD_LOCATION;
  p.precision = QUDA_SINGLE_PRECISION;
  p.fieldOrder = QUDA_SPACE_SPIN_COLOR_FIELD_ORDER;
  p.gammaBasis = QUDA_DEGRAND_ROSSI_GAMMA_BASIS;
  p.create = QUDA_REFERENCE_FIELD_CREATE;
  p.pad = 0;
  p.v = ptr;
  return ColorSpinorField(p);
}

extern "C" {

void qudaAmdMultigridSetHalfStorage(void *mg_instance, int on) { multigridSetHalfStorage(*static_cast<multigrid_solver *>(mg_instance), on != 0); }

int qudaAmdMultigridLevels(void *mg_instance) {
  int n = 0;
  for (MG *m = static_cast<multigrid_solver *>(mg_instance)->mg; m; m = m->getCoarse()) n++;
  return n;
}

// info[18] = Xf[4], Xc[4], fineSpin, fineColor, Nvec, geo_bs[4], spin_bs, 0, 0  (of the transfer level -> level+1)
void qudaAmdMultigridLevelInfo(void *mg_instance, int level, int info[18]) {
  const MG *lv = levelOf(mg_instance, level);
  const Transfer *T = lv->getTransfer();
  if (!T) errorQuda("level %d is the coarsest level: no transfer", level);
  memset(info, 0, 18 * sizeof(int));
  for (int d = 0; d < 4; d++) { info[d] = T->Xf[d]; info[4 + d] = T->Xc[d]; info[11 + d] = T->geo_bs[d]; }
  info[8] = T->fineSpin; info[9] = T->fineColor; info[10] = T->Nvec; info[15] = T->spin_bs;
  info[16] = lv->nullVectorMethod; info[17] = lv->nullVectorIterations;
}

// set-up refinement (multigrid.h multigrid_solver::refine): `passes` inverse-iteration passes of `cycles` multigrid cycles per null vector,
// the hierarchy rebuilt after each; returns the seconds it took
double qudaAmdMultigridRefine(void *mg_instance, int passes, int cycles) {
  multigrid_solver *s = static_cast<multigrid_solver *>(mg_instance);
  if (!s) errorQuda("null multigrid instance");
  const double before = s->mg_param_copy.secs;
  s->refine(passes, cycles);
  return s->mg_param_copy.secs - before;
}

// (aggregate, chirality) blocks of the level's transfer operator that the fp32 CholeskyQR2 orthonormalisation handed to Gram-Schmidt
int qudaAmdMultigridOrthoFallbackBlocks(void *mg_instance, int level) {
  const Transfer *T = levelOf(mg_instance, level)->getTransfer();
  if (!T) errorQuda("level %d is the coarsest level: no transfer", level);
  return T->lastGsFallbackBlocks;
}

void qudaAmdMultigridGetNullVector(void *mg_instance, int level, int k, float *h_out) {
  const std::vector<ColorSpinorField *> &B = levelOf(mg_instance, level)->nullVectors();
  if (k < 0 || k >= (int)B.size()) errorQuda("null vector %d of %zu", k, B.size());
  ColorSpinorField h = hostView(*B[k], h_out);
  h = *B[k];
}

void qudaAmdMultigridGetV(void *mg_instance, int level, float *h_out) {
  const Transfer *T = levelOf(mg_instance, level)->getTransfer();
  if (!T) errorQuda("level %d is the coarsest level: no transfer", level);
  const int K = T->fineSpin * T->fineColor, N = T->Nvec, bv = T->blockVol;
  std::vector<float> v(T->vBytes() / sizeof(float));
  std::vector<int> b2f(T->fineVol);
  HIP_CHECK(hipStreamSynchronize(computeStream()));
  HIP_CHECK(hipMemcpy(v.data(), T->V, T->vBytes(), hipMemcpyDeviceToHost));
  HIP_CHECK(hipMemcpy(b2f.data(), T->block_to_fine, T->fineVol * sizeof(int), hipMemcpyDeviceToHost));
  for (long A = 0; A < T->nAgg; A++)
    for (int b = 0; b < bv; b++) {
      const long f = b2f[A * bv + b];
      for (int k = 0; k < K; k++)
        for (int j = 0; j < N; j++) {
          const size_t src = ((((size_t)A * K + k) * (N / 2) + j / 2) * bv + b) * 4 + (j & 1) * 2;
          const size_t dst = (((size_t)f * K + k) * N + j) * 2;
          h_out[dst] = v[src]; h_out[dst + 1] = v[src + 1];
        }
    }
}

// Y[8][Vc][n][n], X[Vc][n][n] complex fp32 of the operator on level+1, reference index convention (include/
// gauge_field_order.h QDP order; dims 0-3 = the link applied daggered from the backward neighbour, 4-7 forward)
void qudaAmdMultigridGetCoarseLinks(void *mg_instance, int level, float *h_Y, float *h_X) {
  const DiracCoarse *D = levelOf(mg_instance, level)->getCoarseDirac();
  if (!D) errorQuda("level %d has no coarse operator", level);
  const CoarseGauge &G = D->Links();
  const int n = G.n;
  const long Vc = G.nSites, Vh = Vc / 2;
  std::vector<float> g(G.bytes / sizeof(float));
  HIP_CHECK(hipStreamSynchronize(computeStream()));
  HIP_CHECK(hipMemcpy(g.data(), G.data, G.bytes, hipMemcpyDeviceToHost));
  auto elem = [&](long A, int m, int r, int c) { return &g[((((size_t)A * 9 + m) * (n / 2) + c / 2) * n + r) * 4 + (c & 1) * 2]; };
  for (long A = 0; A < Vc; A++) {
    const int par = A >= Vh;
    long l = A - par * Vh;
    const int Xh = G.Xc[0] / 2;
    const int xh = (int)(l % Xh); l /= Xh;
    const int y = (int)(l % G.Xc[1]); l /= G.Xc[1];
    const int z = (int)(l % G.Xc[2]), t = (int)(l / G.Xc[2]);
    const int c[4] = {2 * xh + ((y + z + t + par) & 1), y, z, t};
    for (int r = 0; r < n; r++)
      for (int q = 0; q < n; q++) {
        const float *x = elem(A, 8, r, q);
        float *o = h_X + (((size_t)A * n + r) * n + q) * 2;
        o[0] = x[0]; o[1] = x[1];
      }
    for (int mu = 0; mu < 4; mu++) {
      // forward link: stored at its own site
      for (int r = 0; r < n; r++)
        for (int q = 0; q < n; q++) {
          const float *f = elem(A, 2 * mu, r, q);
          float *o = h_Y + ((((size_t)(mu + 4) * Vc + A) * n + r) * n + q) * 2;
          o[0] = f[0]; o[1] = f[1];
        }
      // backward link of site A + mu is Y_mu(A)^dagger
      int cn[4] = {c[0], c[1], c[2], c[3]};
      cn[mu] = (c[mu] + 1) % G.Xc[mu];
      const int npar = (cn[0] + cn[1] + cn[2] + cn[3]) & 1;
      const long An = npar * Vh + ((((long)cn[3] * G.Xc[2] + cn[2]) * G.Xc[1] + cn[1]) * G.Xc[0] + cn[0]) / 2;
      for (int r = 0; r < n; r++)
        for (int q = 0; q < n; q++) {
          const float *bk = elem(An, 2 * mu + 1, q, r);
          float *o = h_Y + ((((size_t)mu * Vc + A) * n + r) * n + q) * 2;
          o[0] = bk[0]; o[1] = -bk[1];
        }
    }
  }
}

// op 0: R (level -> level+1), 1: P (level+1 -> level), 2: M of level (the operator the cycle takes residuals with)
void qudaAmdMultigridApply(void *mg_instance, int level, int op, float *h_out, const float *h_in) {
  MG *m = levelOf(mg_instance, level);
  const Transfer *T = m->getTransfer();
  if (op < 2 && !T) errorQuda("level %d is the coarsest level: no transfer", level);
  ColorSpinorField *fin = nullptr, *fout = nullptr;
  const ColorSpinorField &proto = *m->nullVectors()[0];
  auto fine = [&]() { ColorSpinorParam p = proto.param(); p.create = QUDA_ZERO_FIELD_CREATE; return new ColorSpinorField(p); };
  switch (op) {
    case 0: fin = fine(); fout = T->createCoarseField(); break;
    case 1: fin = T->createCoarseField(); fout = fine(); break;
    case 2: case 3: fin = fine(); fout = fine(); break;
    default: errorQuda("unknown op %d", op);
  }
  ColorSpinorField hin = hostView(*fin, const_cast<float *>(h_in)), hout = hostView(*fout, h_out);
  *fin = hin;
  fin->twistFlavor = fout->twistFlavor = proto.twistFlavor;
  if (op == 0) T->R(*fout, *fin);
  else if (op == 1) T->P(*fout, *fin);
  else if (op == 3) (*m)(*fout, *fin);   // one multigrid cycle of this level (MG::operator(): the fused persistent kernel where it qualifies)
  else m->residualMatrix()(*fout, *fin);
  hout = *fout;
  delete fin; delete fout;
}

// the cycle below a coarse level as one persistent kernel (coarse_cycle.h): process-wide switch and the statistics of the last launch of `level`
void qudaAmdMultigridSetFused(int on) { coarseCycleSetEnabled(on); }
int qudaAmdMultigridFusedStats(void *mg_instance, int level, long long out[5]) {
  MG *m = levelOf(mg_instance, level);
  coarseCycleStats(m->fusedCycle(), out);
  return m->fusedCycle() ? 1 : 0;
}

// M of a COARSE level applied to nrhs host vectors at once through the multi-right-hand-side MFMA kernel (block.h): h_in / h_out
// hold nrhs vectors back to back, each in the host layout of qudaAmdMultigridApply.  niter > 0: the application is repeated
// niter times between device events on the compute stream and the seconds per application are returned (else 0).
double qudaAmdMultigridApplyBlock(void *mg_instance, int level, int nrhs, float *h_out, const float *h_in, int niter) {
  MG *m = levelOf(mg_instance, level);
  const Dirac *dg = m->residualMatrix().Expose();
  const DiracCoarse *dc = dynamic_cast<const DiracCoarse *>(dg);
  const QudaDiracType ty = dg->getDiracType();
  const bool fine = ty == QUDA_WILSON_DIRAC || ty == QUDA_TWISTED_MASS_DIRAC || ty == QUDA_TWISTED_CLOVER_DIRAC;
  if (!fine && (!dc || ty != QUDA_COARSE_DIRAC)) errorQuda("level %d does not carry a full fine or coarse operator", level);
  if (!fine && !blockCoarseSupported(dc->Links(), nrhs)) errorQuda("block coarse operator not available for n = %d, nrhs = %d on this lattice", dc->Links().n, nrhs);
  if (fine && !fineBlockSupported(*dg->Gauge(), nrhs)) errorQuda("block fine operator not available for nrhs = %d on this lattice", nrhs);
  const ColorSpinorField &proto = *m->nullVectors()[0];
  std::vector<ColorSpinorField *> f(nrhs);
  const size_t len = (size_t)proto.Volume() * proto.Nspin() * proto.Ncolor() * 2;
  for (int i = 0; i < nrhs; i++) {
    ColorSpinorParam p = proto.param(); p.create = QUDA_ZERO_FIELD_CREATE;
    f[i] = new ColorSpinorField(p);
    ColorSpinorField hin = hostView(*f[i], const_cast<float *>(h_in) + i * len);
    *f[i] = hin;
  }
  const int nSites = fine ? proto.Volume() : dc->Links().nSites, ncomp = fine ? 12 : dc->Links().n;
  const int nGhost = fine ? 2 * blockGhost(dg->Gauge()->geom.X, true).nGhost : blockGhost(dc->Links().Xc, false).nGhost;
  BlockField in(nSites, ncomp, nrhs, nGhost), out(nSites, ncomp, nrhs);
  blockPack(in, f);
  // level 0: the multi-right-hand-side stencil of the null-vector solves (dslash.h applyFineBlockM), twisted clover with its dense
  // site matrices A + i a g5
  float *tmat[2] = {nullptr, nullptr};
  const size_t tmatBytes = (size_t)proto.VolumeCB() * 144 * sizeof(float);
  double a = 0.0;
  if (fine) {
    a = ty == QUDA_WILSON_DIRAC ? 0.0 : 2.0 * dg->Kappa() * (double)proto.TwistFlavor() * dg->Mu();
    if (ty == QUDA_TWISTED_CLOVER_DIRAC)
      for (int p = 0; p < 2; p++) { tmat[p] = (float *)poolDeviceMalloc(tmatBytes); cloverTwistDense(tmat[p], *dg->Clover(), p, a, false); }
  }
  auto apply = [&]() {
    if (fine) applyFineBlockM(out.v, in.v, nrhs, *dg->Gauge(), dg->Kappa(), a, tmat[0] ? tmat : nullptr);
    else applyCoarseBlock(out, in, dc->Links());
  };
  apply();
  double secs = 0;
  if (niter > 0) {
    hipEvent_t e0, e1;
    HIP_CHECK(hipEventCreate(&e0)); HIP_CHECK(hipEventCreate(&e1));
    HIP_CHECK(hipEventRecord(e0, computeStream()));
    for (int k = 0; k < niter; k++) apply();
    HIP_CHECK(hipEventRecord(e1, computeStream()));
    HIP_CHECK(hipEventSynchronize(e1));
    float ms = 0;
    HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    HIP_CHECK(hipEventDestroy(e0)); HIP_CHECK(hipEventDestroy(e1));
    secs = 1e-3 * ms / niter;
  }
  blockUnpack(f, out);
  HIP_CHECK(hipStreamSynchronize(computeStream()));
  for (int p = 0; p < 2; p++) if (tmat[p]) poolDeviceFree(tmat[p], tmatBytes);
  for (int i = 0; i < nrhs; i++) {
    ColorSpinorField hout = hostView(*f[i], h_out + i * len);
    hout = *f[i];
    delete f[i];
  }
  return secs;
}
// seconds per application of the level's restrictor (what = 0) or prolongator (1), HIP events on the compute stream
double qudaAmdMultigridTimeTransfer(void *mg_instance, int level, int what, int niter) {
  MG *m = levelOf(mg_instance, level);
  const Transfer *T = m->getTransfer();
  if (!T) errorQuda("level %d has no transfer operator", level);
  ColorSpinorField *fine = T->createFineField(), *coarse = T->createCoarseField();
  blas::copy(*fine, *m->nullVectors()[0]);
  T->R(*coarse, *fine);
  T->P(*fine, *coarse);
  hipEvent_t e0, e1;
  HIP_CHECK(hipEventCreate(&e0)); HIP_CHECK(hipEventCreate(&e1));
  HIP_CHECK(hipEventRecord(e0, computeStream()));
  for (int k = 0; k < niter; k++) { if (what) T->P(*fine, *coarse); else T->R(*coarse, *fine); }
  HIP_CHECK(hipEventRecord(e1, computeStream()));
  HIP_CHECK(hipEventSynchronize(e1));
  float ms = 0;
  HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
  HIP_CHECK(hipEventDestroy(e0)); HIP_CHECK(hipEventDestroy(e1));
  delete fine; delete coarse;
  return 1e-3 * ms / niter;
}
// the single-right-hand-side coarse operator of the same level, timed the same way (seconds per application)
double qudaAmdMultigridTimeApply(void *mg_instance, int level, int niter) {
  MG *m = levelOf(mg_instance, level);
  const ColorSpinorField &proto = *m->nullVectors()[0];
  ColorSpinorParam p = proto.param(); p.create = QUDA_ZERO_FIELD_CREATE;
  ColorSpinorField a(p), b(p);
  blas::copy(a, proto);
  a.twistFlavor = b.twistFlavor = proto.twistFlavor;
  m->residualMatrix()(b, a);
  hipEvent_t e0, e1;
  HIP_CHECK(hipEventCreate(&e0)); HIP_CHECK(hipEventCreate(&e1));
  HIP_CHECK(hipEventRecord(e0, computeStream()));
  for (int k = 0; k < niter; k++) m->residualMatrix()(b, a);
  HIP_CHECK(hipEventRecord(e1, computeStream()));
  HIP_CHECK(hipEventSynchronize(e1));
  float ms = 0;
  HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
  HIP_CHECK(hipEventDestroy(e0)); HIP_CHECK(hipEventDestroy(e1));
  return 1e-3 * ms / niter;
}

}
