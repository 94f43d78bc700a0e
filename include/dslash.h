// dslash.h — host interface of the fine-grid stencil kernels (Wilson / twisted-mass / twisted-clover).
//
// Replaces the reference's per-action free functions wilsonDslashCuda (include/dslash_quda.h:45),
// twistedMassDslashCuda (lib/dslash_twisted_mass.cu:167), twistedCloverDslashCuda
// (lib/dslash_twisted_clover.cu:257), twistGamma5Cuda (lib/dslash_quda.cu:430) and
// twistCloverGamma5Cuda (:562) with one parameterised launch.
#pragma once

#include "fields.h"

namespace quda {

// what the kernel does with the accumulated hopping term  acc = sum_{8 dirs} U P psi
enum DslashMode {
  DSLASH_PLAIN = 0,            // out = acc                       [xpay: out = x + k acc]             (Wilson)
  DSLASH_TWIST_INV = 1,        // out = b (1 + i a g5) acc        [xpay: out = x + b(..)acc, b incl. k] (QUDA_DEG_DSLASH_TWIST_INV)
  DSLASH_TWIST_INV_DSLASH = 2, // acc built from (1 + i a g5) psi; out = b acc [xpay: x + b acc]       (QUDA_DEG_TWIST_INV_DSLASH)
  DSLASH_TWIST_XPAY = 3,       // out = k acc + (1 + i a g5) x                                         (QUDA_DEG_DSLASH_TWIST_XPAY)
  DSLASH_CLOVER_TWIST_INV = 4, // out = Ainv (A + i a g5) acc     [xpay: out = x + k Ainv(..)acc]      (QUDA_DEG_DSLASH_CLOVER_TWIST_INV)
  DSLASH_CLOVER_TWIST_XPAY = 5 // out = k acc + (A + i a g5) x                                         (QUDA_DEG_DSLASH_CLOVER_TWIST_XPAY)
};

struct DslashParam {
  DslashMode mode = DSLASH_PLAIN;
  int parity = 0;      // parity of the OUTPUT sites
  int dagger = 0;
  double a = 0, b = 1, k = 0;   // already dagger-adjusted by the caller
  const ColorSpinorField *x = nullptr;  // xpay field or nullptr
  const CloverField *clover = nullptr;
  int kernel_type = 0;  // 0 interior(+all if unpartitioned), 1 exterior
};

// out(parity) = stencil(in(other parity)); fields must be device parity fields of equal precision.
void applyDslash(ColorSpinorField &out, const ColorSpinorField &in, const GaugeField &U, const DslashParam &p);

// one direction of the stencil only: out(parity) = coef * U P psi(x + dhat(dir)), dir = 2 mu + (0 fwd, 1 bwd), no dagger
// (building block of the Galerkin coarse-operator construction)
void applyHopDir(ColorSpinorField &out, const ColorSpinorField &in, const GaugeField &U, int parity, int dir, double coef);
// the same hop WITHOUT spin projection, optionally accumulated: out(parity) = xcoef x + coef U psi(x + dhat(dir)); x may be
// nullptr or the output field itself (building block of the Gaussian source smearing, qkxtm.hip)
void applyCovariantShift(ColorSpinorField &out, const ColorSpinorField &in, const GaugeField &U, int parity, int dir, double coef,
                         const ColorSpinorField *x, double xcoef);
// out(x) = in(x + dhat(dir)) for a 24-real fp64 planar site field (out: parity block `parity`, in: the other parity's block), ghost-aware
void applyShift(double *out, const double *in, const LatticeGeom &g, int stride, int parity, int dir);

// multi-right-hand-side FULL operator on block fields ([parity * Vh + x][12 spin-colour][nrhs] float2, see block.h):
// out = (1 + i a g5) in - kappa D in for nrhs vectors per link load (fp32, recon 18, unpartitioned lattice)
bool fineBlockSupported(const GaugeField &U, int nrhs);
// tmat (twisted clover): per parity the dense site matrices A + i a g5 from cloverTwistDense(), which replace (1 + i a g5)
// grid-decomposed lattice: `in` carries the ghost zones of its two halves behind the local panels (2 x blockGhost(X, true).nGhost) and is written there
void applyFineBlockM(float2 *out, float2 *in, int nrhs, const GaugeField &U, double kappa, double a, const float *const tmat[2] = nullptr);
// one parity of the generalised form: out = s0 (1 + i a0 g5) in_same + k1 (1 + i a1 g5) [8 hops of in_other], single-parity panels.
// tmat != nullptr: dense site matrices [Vh][2 chiralities][6 x 6 complex] of the output parity in place of (1 + i a1 g5) on the hop
// sum (tmode 1) or of (1 + i a0 g5) on in_same (tmode 2) — the twisted-clover operators (reference lib/dirac_twisted_clover.cpp:191-330)
// dots != nullptr (8 right-hand sides): inner products per right-hand side in the kernel's epilogue, where `out` and `in_same` are in registers — mode 1:
// (a, out); mode 2: (out, in_same), |out|^2, (a, in_same), (a, out) — finished by fineBlockDotsFinish(sums): sums[k * nrhs + i], k = re / im of
// the products in that order (2 resp. 7 values per right-hand side), global sums on a grid-decomposed lattice
struct FineBlockDots { const float2 *a; int mode; };
void applyFineBlockParity(float2 *out, const float2 *in_same, const float2 *in_other, int nrhs, const GaugeField &U, int parity, double s0, double a0, double k1,
                          double a1, const float *tmat = nullptr, int tmode = 0, float2 *ghost = nullptr, const FineBlockDots *dots = nullptr);
bool fineBlockDotsSupported(int nrhs);
void fineBlockDotsFinish(double *sums, int nrhs, int mode);
// mode 3 (4 or 8 right-hand sides): (out, in_same), |out|^2 only — the sums of a minimal-residual step, no further field read (a unused).
// fineBlockDotsFinishDev: the sums stay in DEVICE memory (rank-local), no host round trip — the update kernel that follows reads them
void fineBlockDotsFinishDev(double *d_sums, int nrhs, int mode);
void freeFineBlockDots();
// ghost: on a grid-decomposed lattice the ghost zone of in_other (blockGhost(X, true).nGhost panels, a whole number of panels away from
// in_other); it is filled here (pack + grouped exchange) before the launch
// out[site][chirality][6][6] complex fp32 = A + i a s (s = +-1 for the upper / lower chirality) of one parity, or its inverse
int haloWireFormat();
// direct Galerkin construction, step 1 (dslash.hip galerkin_uv_kernel): UV(x) = coef U_dir(x) V(x + dhat(dir)) for all columns of the transfer
// matrix at once, V / UV in its aggregate-major order (4^4 aggregates); and the same for the site-diagonal term A + i a g5 of twisted clover
void galerkinUV(float *UV, const float *V, const GaugeField &U, int dir, double coef, const int *block_to_fine, const int *fine_to_block, int aggOffset, int nAgg, int blockVol, int nvec, bool classMajor = false);   // classMajor: sites of a row ordered by (block coordinate along mu, the other three)   // aggregates [aggOffset, aggOffset + nAgg) -> UV[0 .. nAgg)
void galerkinLocalUV(float *L, const float *V, const CloverField &C, double a, const int *block_to_fine, int aggOffset, int nAgg, int blockVol, int nvec);
void cloverTwistDense(float *out, const CloverField &C, int parity, double a, bool inverse);

// site-local kernels
enum SiteOp {
  SITE_TWIST = 0,              // out = b (1 + i a g5) in
  SITE_CLOVER = 1,             // out = A in           (or Ainv in if inverse)
  SITE_CLOVER_TWIST = 2,       // out = (A + i a g5) in
  SITE_CLOVER_TWIST_INV = 3    // out = Ainv (A + i a g5) in
};
void applySite(ColorSpinorField &out, const ColorSpinorField &in, SiteOp op, double a, double b, const CloverField *clover,
               int parity, bool inverse);

// launch geometry of the stencil kernel (block size, XCD mapping, block order); see launchDslash
struct DslashTune {
  int block = 0;       // threads per block; 0: the largest of 256/192/128/64 that divides an (x, y) plane
  int remap = 1;       // XCD-aware block mapping on / off
  int order = 1;       // legacy slab order: 1 = t-interleave over the whole XCD slab, n > 1 = over n slices
  int store_aux = -1;  // cache policy of the output stores: -1 automatic (nt from 2^18 checkerboard sites), 0 default, 2 nt
  int link_aux = -1;   // cache policy of 16-bit link loads: -1 automatic (nt once an application's working set exceeds the 256 MiB Infinity Cache), 0 default, 2 nt
  int tiled = -1;      // plane-tiled order: -1 automatic (on where the lattice allows), 0 off, 1 chunk-major tiles, 2 z-major tiles
  int nxz = 0, tz = 0, tt = 0;   // plane-tiled order: XCDs along z, tile extents in z and t (0: automatic)
  int lds_pad = 0;     // dynamic LDS per block, only to cap the blocks per CU (measurement aid)
  int ygroups = -1;    // plane-tiled order: groups of plane chunks walked one after the other (0 / 1 none, -1 automatic: fp64 fields whose three-slice set per XCD exceeds the L2, n explicit)
  // peer-store halo launch: face packing folded into the site threads (-1: environment QUDA_AMD_P2P_FOLD, default OFF; the path is compiled out unless dslash.hip is built with -DQA_P2P_FOLD=1: it measured slower than pack blocks), start delay of
  // the site blocks and raised issue priority of the pack waves (measurement aids)
  int p2p_fold = -1, site_delay = 0, pack_prio = 0;
  // wire format of the peer-store ghost zones: 0 flag-in-data {word, flag, word, flag} vectors, 1 self-validating 16-byte atoms {3 words, flag} = one
  // 128-byte line per fp64 face site, -1 automatic (atoms between devices, flag-in-data where ranks share one); QUDA_AMD_HALO_FORMAT=ll|atom; must be
  // the same on every rank
  int halo_format = -1;
  int edge_first = 1;  // peer-store launch, plane-tiled order: every XCD starts with its boundary planes
};
DslashTune &dslashTune();
void setDslashTune(const char *key, int value);

// analytic work model per checkerboard site
long long dslashFlopsPerSite(DslashMode mode, bool xpay);
long long dslashBytesPerSite(QudaPrecision prec, int recon, DslashMode mode, bool xpay);

}  // namespace quda
