// transfer.h — inter-grid transfer operators of the adaptive multigrid (reference include/transfer.h,
// lib/transfer.cpp:21-348, lib/transfer_util.cu:15-363, lib/restrictor.cu, lib/prolongator.cu).
//
// V holds the block-orthonormalised null vectors.  R = V^dagger summed over each aggregate and over the fine spins of a
// chirality, P = V broadcast (reference restrictor.cu:51-125, prolongator.cu:42-116).  MI355X layout: V is stored
// AGGREGATE-MAJOR — [aggregate][fine spin-colour k][vector pair][site in aggregate] as float4 — so the work-group
// that owns an aggregate streams one contiguous 2304 B/site slab with 16-byte unit-stride lanes; the (4 %) fine-vector
// traffic goes through a site map.  Both kernels are pure HBM streams of V (V + fine vec + coarse vec bytes).
#pragma once

#include <vector>

#include "fields.h"

namespace quda {

class Transfer {
 public:
  int Nvec;
  int geo_bs[4];
  int spin_bs;
  int fineSpin, fineColor;     // of the level this transfer starts from
  int Xf[4], Xc[4];            // fine / coarse full lattice extents
  int blockVol;                // sites per aggregate
  bool parityMajor = false;    // sites of an aggregate numbered even half first (transfer.hip block_coords), else lexicographically
  int lastGsFallbackBlocks = 0; // (aggregate, chirality) blocks the CholeskyQR2 orthonormalisation handed to Gram-Schmidt (ill-conditioned in fp32)
  int nAgg;                    // aggregates = coarse volume
  long fineVol;
  float *V;                    // device, aggregate-major (see above)
  mutable void *V_h;           // optional fp16 mirror of V streamed by R and P in the solve phase (coarse.h setCoarseHalfStorage)
  void makeHalf() const;
  int *block_to_fine;          // [A*blockVol + b] -> fine full index (parity*Vh + x_cb)
  int *fine_to_block;          // inverse
  mutable unsigned long long flops_;
  QudaSiteSubset site_subset;
  QudaParity subset_parity;

  // B: Nvec null vectors on the fine level (device, fp32, full fields).  geo_bs is adjusted in place with the reference's
  // fallback rule (lib/transfer.cpp:31-44) so callers can read back the block size actually used.
  Transfer(const std::vector<ColorSpinorField *> &B, int Nvec, int *geo_bs, int spin_bs);
  ~Transfer();

  // dir < 0: plain R.  dir in 0..7 with boundary = 0/1: only fine sites whose neighbour in direction dir lies inside /
  // outside the site's own aggregate contribute (used by the Galerkin coarse-operator construction).
  void R(ColorSpinorField &coarse, const ColorSpinorField &fine, int dir = -1, int boundary = 0) const;
  void P(ColorSpinorField &fine, const ColorSpinorField &coarse) const;
  // FOUR sources per pass over V (the multi-source cycle of invertMultiSrcQuda; fine level, fp32 V, aggregates of 64 .. 256 sites)
  bool canQuad() const;
  void R4(ColorSpinorField *const coarse[4], const ColorSpinorField *const fine[4]) const;
  void P4(ColorSpinorField *const fine[4], const ColorSpinorField *const coarse[4]) const;
  // the same with the four fine vectors in columns col0 .. col0 + 3 of a pair-major block field (block.h) of the subset parity: `panel` = its data, nrhs its columns
  void R4Block(ColorSpinorField *const coarse[4], const float2 *panel, int nrhs, int col0) const;
  void P4Block(float2 *panel, int nrhs, int col0, bool accumulate, const ColorSpinorField *const coarse[4]) const;
  // both halves of the Galerkin split in one pass over V: `leaving` = R over the fine sites whose dir-neighbour lies outside
  // their aggregate, `staying` = R over the others (equal to R(.., dir, 1) and R(.., dir, 0))
  void RSplit(ColorSpinorField &leaving, ColorSpinorField &staying, const ColorSpinorField &fine, int dir) const;
  // the same for four fine vectors (each with its own direction) in ONE pass over V; fine level only (canSplit4)
  bool canSplit4() const;
  // direct Galerkin construction of the first coarse level on the matrix cores (transfer.hip galerkin_vuv_kernel; reference ComputeVUV,
  // lib/coarse_op.cuh:487-600): forward link 2 mu and in-aggregate part S (slot 8, accumulated over the directions) of every coarse site
  // from UV = dslash.h galerkinUV(V); 4^4 aggregates of a 4 x 3 fine level, Nvec 8 / 24, unpartitioned
  bool canDirectGalerkin() const;
  void directGalerkinVUV(float *links, const float *UV, int mu, bool accumulateLocal, bool local = false, int aggOffset = 0, int nAggChunk = 0, bool classMajor = false) const;   // local: UV is a chirality-diagonal site term, everything goes to slot 8; UV holds the aggregates [aggOffset, aggOffset + nAggChunk)
  void RSplit4(ColorSpinorField *const leaving[4], ColorSpinorField *const staying[4], ColorSpinorField *const fine[4], const int dir[4]) const;
  // fine = P e_j for the coarse unit vector j (same component at every coarse site): column j of V, without streaming all of V
  void column(ColorSpinorField &fine, int j) const;
  // single-parity fine fields (outer even-odd preconditioned solve: the residual of one parity is injected into the coarse
  // grid, reference Transfer::setSiteSubset lib/transfer.cpp:276-290): the absent parity restricts as zero / is not prolongated
  void setSiteSubset(QudaSiteSubset subset, QudaParity parity);

  ColorSpinorField *createCoarseField() const;   // reference ColorSpinorField::CreateCoarse, lib/color_spinor_field.cpp:737-763
  ColorSpinorField *createFineField() const;
  unsigned long long flops() const { unsigned long long f = flops_; flops_ = 0; return f; }
  size_t vBytes() const { return (size_t)fineVol * fineSpin * fineColor * Nvec * 2 * sizeof(float); }

 private:
  void createGeoMap();
  void fillAndOrthonormalise(const std::vector<ColorSpinorField *> &B);
};

// uniform(0,1) random spinor from a counter-based generator keyed by (seed, global site, component): the same field for
// any process grid (reference uses a private rand48 clone, lib/color_spinor_util.cu:12-22)
void spinorRandom(ColorSpinorField &f, unsigned long long seed);

// half-precision storage switch of the hierarchy (coarse.hip): R, P and the coarse operators stream fp16 mirrors of V / Y when set
void setCoarseHalfStorage(bool on);
bool coarseHalfStorage();

}  // namespace quda
