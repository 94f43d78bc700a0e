// fields.h — lattice fields of the MI355X-native library.
//
// ColorSpinorField keeps the reference's layout contract (include/color_spinor_field.h:75-456,
// lib/color_spinor_field.cpp:129-216): parity fields with x[0] halved, stride = volumeCB + pad, planar
// FLOAT2 (fp64: nSpin*nColor double2 planes) / FLOAT4 (fp32: 6 float4 planes) / 16-bit (6 short4 planes +
// one fp32 norm per site, int16 fixed point as lib/io_spinor.h:49-62) device orders, full fields = even
// half then odd half, twistFlavor / gammaBasis attributes.  One deliberate difference, allowed by
// the device-internal spin basis of nSpin=4 fields is DeGrand-Rossi (chiral), not UKQCD,
// so gamma5, the twist and the clover term are spin-diagonal / chirality-block-diagonal in the kernels and
// the multigrid transfer needs no rotation; host fields in UKQCD basis are rotated on upload/download.
//
// GaugeField is re-designed for CDNA4 (the device link layout is not API-visible): for every parity p and
// checkerboard site x it stores the EIGHT matrices the stencil multiplies with at x,
//      W[p][2mu](x) = U_mu(x)        W[p][2mu+1](x) = U_mu(x - mu)^dagger,
// i.e. backward links are pre-gathered and pre-daggered once at load time.  Every link load of the Dslash
// is then a perfectly coalesced, aligned, unit-stride 16-byte-per-lane stream indexed by the thread's own
// site (no neighbour index arithmetic, no dagger variant, no ghost-link pad); HBM traffic per Dslash is
// unchanged (8 distinct links per site either way) at the price of 2x gauge capacity, which 288 GB absorbs.
#pragma once

#include "qa_core.h"

namespace quda {

// ------------------------------------------------------------------------------------------------
struct ColorSpinorParam {
  QudaFieldLocation location = QUDA_CUDA_FIELD_LOCATION;
  int nColor = 3, nSpin = 4, nDim = 4;
  int x[QUDA_MAX_DIM] = {0, 0, 0, 0, 0, 0};  // x[0] already halved for parity fields (reference convention)
  QudaPrecision precision = QUDA_DOUBLE_PRECISION;
  int pad = 0;
  bool planePad = true;  // device spin-4 fields: add the library's plane padding (fieldPadSites) to pad; false in the param() of an existing field, whose pad already holds it
  QudaTwistFlavorType twistFlavor = QUDA_TWIST_NO;
  QudaSiteSubset siteSubset = QUDA_PARITY_SITE_SUBSET;
  QudaSiteOrder siteOrder = QUDA_EVEN_ODD_SITE_ORDER;
  QudaFieldOrder fieldOrder = QUDA_INVALID_FIELD_ORDER;  // chosen from precision for device fields
  QudaGammaBasis gammaBasis = QUDA_DEGRAND_ROSSI_GAMMA_BASIS;
  QudaFieldCreate create = QUDA_ZERO_FIELD_CREATE;
  void *v = nullptr;     // for QUDA_REFERENCE_FIELD_CREATE
  void *norm = nullptr;
  ColorSpinorParam() {}
  // host field as described by the C ABI (reference include/color_spinor_field.h:118-160)
  ColorSpinorParam(void *V, const QudaInvertParam &inv, const int *X, bool pc_solution);
};

class ColorSpinorField {
 public:
  QudaFieldLocation location;
  int nColor, nSpin, nDim;
  int x[4];            // x[0] halved for parity fields
  int volume, volumeCB, stride, pad;
  QudaPrecision precision;
  QudaSiteSubset siteSubset;
  QudaSiteOrder siteOrder;
  QudaFieldOrder fieldOrder;
  QudaGammaBasis gammaBasis;
  QudaTwistFlavorType twistFlavor;
  size_t bytes, norm_bytes;   // whole allocation (both parities for full fields)
  void *v_;
  void *norm_;
  bool owns;
  ColorSpinorField *even_, *odd_;

  explicit ColorSpinorField(const ColorSpinorParam &param);
  ColorSpinorField(const ColorSpinorField &src);  // deep copy, same location/layout
  virtual ~ColorSpinorField();
  ColorSpinorField &operator=(const ColorSpinorField &src);  // copy with reorder/precision/basis change

  void *V() { return v_; }
  const void *V() const { return v_; }
  void *Norm() { return norm_; }
  const void *Norm() const { return norm_; }
  int Nspin() const { return nSpin; }
  int Ncolor() const { return nColor; }
  int Volume() const { return volume; }
  int VolumeCB() const { return volumeCB; }
  int Stride() const { return stride; }
  int X(int d) const { return x[d]; }
  QudaPrecision Precision() const { return precision; }
  QudaSiteSubset SiteSubset() const { return siteSubset; }
  QudaTwistFlavorType TwistFlavor() const { return twistFlavor; }
  void changeTwist(QudaTwistFlavorType f) { twistFlavor = f; if (even_) even_->twistFlavor = f; if (odd_) odd_->twistFlavor = f; }
  QudaFieldLocation Location() const { return location; }
  size_t Bytes() const { return bytes; }
  long Length() const { return (long)(siteSubset == QUDA_FULL_SITE_SUBSET ? 2 : 1) * stride * nColor * nSpin * 2; }
  long RealLength() const { return (long)volume * nColor * nSpin * 2; }

  ColorSpinorField &Even();
  ColorSpinorField &Odd();
  const ColorSpinorField &Even() const { return const_cast<ColorSpinorField *>(this)->Even(); }
  const ColorSpinorField &Odd() const { return const_cast<ColorSpinorField *>(this)->Odd(); }

  void zero();
  // full local lattice dims (x[0] un-halved)
  void latticeDims(int X[4]) const {
    for (int d = 0; d < 4; d++) X[d] = x[d];
    if (siteSubset == QUDA_PARITY_SITE_SUBSET) X[0] *= 2;
  }
  static ColorSpinorField *Create(const ColorSpinorParam &p) { return new ColorSpinorField(p); }
  ColorSpinorParam param() const;

 private:
  void createViews();
};

// source-compatible names (reference include/color_spinor_field.h:458, :640)
class cudaColorSpinorField : public ColorSpinorField {
 public:
  explicit cudaColorSpinorField(const ColorSpinorParam &p);
  cudaColorSpinorField(const ColorSpinorField &src, const ColorSpinorParam &p);
  cudaColorSpinorField &operator=(const ColorSpinorField &src) { ColorSpinorField::operator=(src); return *this; }
};
class cpuColorSpinorField : public ColorSpinorField {
 public:
  explicit cpuColorSpinorField(const ColorSpinorParam &p);
  cpuColorSpinorField &operator=(const ColorSpinorField &src) { ColorSpinorField::operator=(src); return *this; }
};

void copyColorSpinor(ColorSpinorField &dst, const ColorSpinorField &src);

// ------------------------------------------------------------------------------------------------
// Fine-grid SU(3) gauge field in the bidirectional device layout described at the top of this file.
class GaugeField {
 public:
  LatticeGeom geom;
  QudaPrecision precision;
  QudaReconstructType reconstruct;  // 18 or 12
  QudaTboundary t_boundary;
  double anisotropy;
  int stride;               // = Vh
  size_t link_bytes;        // bytes of one (parity, direction) block
  size_t bytes;
  void *data;               // [parity][dir 0..7][link_bytes]
  bool tbc_folded;          // true: boundary sign is inside the stored links (recon 18)

  GaugeField(const LatticeGeom &g, QudaPrecision prec, QudaReconstructType recon, QudaTboundary tb, double aniso);
  ~GaugeField();
  // host QDP-order links (array of 4 pointers, even then odd, row-major 3x3; reference SURVEY section 9)
  void loadQDP(void *const h_gauge[4], QudaPrecision cpu_prec);
  void copyFrom(const GaugeField &src);   // device copy with precision change (fp32 -> 16-bit), same reconstruct
  const void *block(int parity, int dir) const { return (const char *)data + ((size_t)parity * 8 + dir) * link_bytes; }
  const void *parityBase(int parity) const { return block(parity, 0); }
  double GiB() const { return bytes / (double)(1 << 30); }
};

// Clover term A and its (twisted) inverse, two Hermitian 6x6 chiral blocks per site.  Host packed order
// as reference tests/clover_reference.cpp:25-53; device: planar 16-byte vectors per chiral block
// (fp64 18 double2, fp32 9 float4, 16-bit 9 short4 + one fp32 norm per block).  Values are stored
// un-halved (the reference's native order carries a factor 1/2, include/clover_field_order.h:425-429;
// an API-level replacement may drop it).
class CloverField {
 public:
  LatticeGeom geom;
  QudaPrecision precision;
  int stride;
  size_t parity_bytes, parity_norm_bytes, bytes;
  void *clover;      // [parity] A
  void *cloverInv;   // [parity] (A^2 + mu2)^-1 if twisted, else A^-1
  float *norm;       // [parity][2 blocks][stride] (16-bit only)
  float *invNorm;
  bool twisted;
  double mu2;
  double trlog[2];

  CloverField(const LatticeGeom &g, QudaPrecision prec);
  ~CloverField();
  void loadPacked(const void *h_clover, const void *h_inv, QudaPrecision cpu_prec);
  // A = 1 + i c sum_{mu>nu} sigma_mu_nu F_mu_nu from the resident links (reference createCloverQuda, lib/interface_quda.cpp:3950-4010:
  // computeFmunu + computeClover with c = clover_coeff); fp64 arithmetic, stored in this field's precision
  void computeFromGauge(const GaugeField &U, double coeff);
  void savePacked(void *h_clover, QudaPrecision cpu_prec) const;
  // compute cloverInv = (A^2 + mu2)^-1 (mu2 = 0: plain inverse) on device; reference lib/clover_invert.cu:56-85
  void computeInverse(double mu2);
  void savePackedInverse(void *h_inv, QudaPrecision cpu_prec) const;
  const void *A(int parity) const { return (const char *)clover + (size_t)parity * parity_bytes; }
  const void *Ainv(int parity) const { return (const char *)cloverInv + (size_t)parity * parity_bytes; }
  const float *Anorm(int parity) const { return norm ? norm + (size_t)parity * 2 * stride : nullptr; }
  const float *AinvNorm(int parity) const { return invNorm ? invNorm + (size_t)parity * 2 * stride : nullptr; }
  double GiB() const { return 2.0 * bytes / (double)(1 << 30); }
};

// gauge tools on the device (fields.hip): APE smearing of the spatial links (a new fp64 field), the plaquette averages, and the
// forward links back in host QDP order
GaugeField *apeSmear(const GaugeField &U, unsigned nSteps, double alpha);
void plaquette(const GaugeField &U, double plq[3]);
void saveGaugeQDP(const GaugeField &U, void *const h_gauge[4], QudaPrecision cpu_prec);

}  // namespace quda
