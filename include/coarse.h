// coarse.h — the coarse-grid operator of the multigrid hierarchy: dense (2 Nc x 2 Nc) complex link matrices per
// site and direction plus a dense local term (reference lib/dirac_coarse.cpp, lib/dslash_coarse.cu:50-333,
// lib/coarse_op.cuh:1310-1498).
//
//   M_c = X + sum_{d=0..7} H_d(x) delta_{x + dhat(d)}        d = 2 mu + (0 forward, 1 backward)
//
// Differences from the reference's storage, none visible through the operator API:
//   * like the fine links (fields.h) the coarse links are BIDIRECTIONAL: all 8 matrices a site multiplies with are
//     stored at that site (the reference stores Y_mu(x) and applies Y_mu(x - mu)^dagger from the neighbour), so the
//     apply kernel streams 9 contiguous matrices per site;
//   * the hopping normalisation (-kappa) is folded into H_d and X at construction;
//   * matrices are column-pair major, [site][matrix][column pair][row] float4, so lane = row reads 16-byte unit-stride.
// The Galerkin construction V^dagger (L + sum_d H_d) V is done on the device by probing the parent operator's
// single-direction hops with the 2 Nvec prolongated unit vectors and restricting the part that stays inside /
// leaves each aggregate (reference computes UV and VUV on the CPU, lib/dirac_coarse.cpp:81-91).
#pragma once

#include "dirac.h"
#include "transfer.h"

namespace quda {

struct CoarseGauge {
  int Xc[4];
  int nSites;      // full coarse volume
  int n;           // 2 * Ncolor
  float *data;     // [site][9][n/2][n] float4
  // optional fp16 mirror (same index structure, 8 bytes per column pair) the apply kernel streams instead of `data` when the
  // hierarchy runs with half-precision storage (QUDA_AMD_MG_HALF=1 / qudaAmdMultigridSetHalfStorage): the operator is
  // HBM-bound on exactly these bytes; construction, inversion and introspection stay fp32
  mutable void *data_h;
  void makeHalf() const;
  size_t bytes;
  CoarseGauge(const int Xc[4], int n);
  ~CoarseGauge();
};

class DiracCoarse : public Dirac {
 protected:
  const Transfer *transfer;
  const Dirac *parent;
  CoarseGauge *links;
  bool ownLinks;
  int Nc;
  QudaTwistFlavorType fineFlavor;
  void build();

 public:
  explicit DiracCoarse(const DiracParam &p);              // builds the links from p.dirac through p.transfer
  DiracCoarse(const DiracCoarse &other, const DiracParam &p);  // shares the links
  ~DiracCoarse() override;

  void Dslash(ColorSpinorField &out, const ColorSpinorField &in, const QudaParity parity) const override;
  void DslashXpay(ColorSpinorField &out, const ColorSpinorField &in, const QudaParity parity, const ColorSpinorField &x, const double &k) const override;
  void Clover(ColorSpinorField &out, const ColorSpinorField &in, const QudaParity parity) const;
  void localTermParity(ColorSpinorField &out, const ColorSpinorField &in, int parity) const override { Clover(out, in, parity ? QUDA_ODD_PARITY : QUDA_EVEN_PARITY); }
  void M(ColorSpinorField &out, const ColorSpinorField &in) const override;
  void MdagM(ColorSpinorField &out, const ColorSpinorField &in) const override;
  void prepare(ColorSpinorField *&src, ColorSpinorField *&sol, ColorSpinorField &x, ColorSpinorField &b, const QudaSolutionType) const override;
  void reconstruct(ColorSpinorField &x, const ColorSpinorField &b, const QudaSolutionType) const override;
  void hopDir(ColorSpinorField &out, const ColorSpinorField &in, int dir) const override;
  void localTerm(ColorSpinorField &out, const ColorSpinorField &in) const override;
  const CoarseGauge &Links() const { return *links; }
  int Ncolor() const { return Nc; }
  // preconditioned links: slots 0..7 = Xinv H_d, slot 8 = Xinv (reference Yhat / Xinv, lib/coarse_op.cuh:1217-1278, :1468-1475);
  // built on first use by a dense batched Gauss-Jordan inverse + batched matrix products on the device
  const CoarseGauge &HatLinks() const;
  void CloverInv(ColorSpinorField &out, const ColorSpinorField &in, const QudaParity parity) const;

 protected:
  mutable CoarseGauge *hat;
  mutable bool ownHat;
};

// even-odd preconditioned coarse operator (reference DiracCoarsePC, lib/dirac_coarse.cpp:237-372)
class DiracCoarsePC : public DiracCoarse {
 public:
  DiracCoarsePC(const DiracCoarse &other, const DiracParam &p);
  void Dslash(ColorSpinorField &out, const ColorSpinorField &in, const QudaParity parity) const override;   // A^-1 D
  void DslashXpay(ColorSpinorField &out, const ColorSpinorField &in, const QudaParity parity, const ColorSpinorField &x, const double &k) const override;
  void M(ColorSpinorField &out, const ColorSpinorField &in) const override;
  void prepare(ColorSpinorField *&src, ColorSpinorField *&sol, ColorSpinorField &x, ColorSpinorField &b, const QudaSolutionType) const override;
  void reconstruct(ColorSpinorField &x, const ColorSpinorField &b, const QudaSolutionType) const override;
};

// mmask: bit m set = include matrix m (0..7 hops, 8 local); parity: -1 both, else only that output parity
void applyCoarse(ColorSpinorField &out, const ColorSpinorField &in, const CoarseGauge &G, int mmask, int parity);

}  // namespace quda
