// dirac.h — operator classes behind the reference's C++ surface (include/dirac_quda.h:15-164, :449-617,
// :868-1032): same class names, method names, argument meaning and flop accounting, so in-library callers
// (solvers, multigrid, QKXTM-style drivers) read as they do against the reference.  The implementations
// launch the CDNA4 kernels of dslash.hip / coarse.hip.
#pragma once

#include "dslash.h"

namespace quda {

class Transfer;
struct CoarseGauge;

struct DiracParam {
  QudaDiracType type = QUDA_INVALID_DIRAC;
  double kappa = 0, mass = 0, m5 = 0, mu = 0, epsilon = 0;
  QudaMatPCType matpcType = QUDA_MATPC_INVALID;
  QudaDagType dagger = QUDA_DAG_NO;
  GaugeField *gauge = nullptr;
  CloverField *clover = nullptr;
  ColorSpinorField *tmp1 = nullptr, *tmp2 = nullptr;
  int commDim[QUDA_MAX_DIM] = {1, 1, 1, 1, 1, 1};
  // coarse operators
  const Transfer *transfer = nullptr;
  const class Dirac *dirac = nullptr;   // fine operator the coarse one is built from
  QudaTwistFlavorType twistFlavor = QUDA_TWIST_NO;  // flavour of the fine-level fields the hierarchy is built for
};

class Dirac {
 protected:
  GaugeField *gauge;
  double kappa, mass;
  QudaMatPCType matpcType;
  mutable QudaDagType dagger;
  mutable unsigned long long flops;
  mutable ColorSpinorField *tmp1, *tmp2;
  mutable bool own1, own2;
  QudaDiracType type;
  int commDim[QUDA_MAX_DIM];

  ColorSpinorField *getTmp(ColorSpinorField *&slot, bool &own, const ColorSpinorField &like) const;
  void checkParitySpinor(const ColorSpinorField &a, const ColorSpinorField &b) const;
  void checkFullSpinor(const ColorSpinorField &a, const ColorSpinorField &b) const;

 public:
  explicit Dirac(const DiracParam &p);
  virtual ~Dirac();

  virtual void Dslash(ColorSpinorField &out, const ColorSpinorField &in, const QudaParity parity) const = 0;
  virtual void DslashXpay(ColorSpinorField &out, const ColorSpinorField &in, const QudaParity parity, const ColorSpinorField &x,
                          const double &k) const = 0;
  virtual void M(ColorSpinorField &out, const ColorSpinorField &in) const = 0;
  virtual void MdagM(ColorSpinorField &out, const ColorSpinorField &in) const = 0;
  void Mdag(ColorSpinorField &out, const ColorSpinorField &in) const;   // reference lib/dirac.cpp:71-76
  void MMdag(ColorSpinorField &out, const ColorSpinorField &in) const;

  virtual void prepare(ColorSpinorField *&src, ColorSpinorField *&sol, ColorSpinorField &x, ColorSpinorField &b,
                       const QudaSolutionType solType) const = 0;
  virtual void reconstruct(ColorSpinorField &x, const ColorSpinorField &b, const QudaSolutionType solType) const = 0;

  // Decomposition of the full (unpreconditioned, undaggered) operator M = L + sum_{d=0..7} H_d used by the Galerkin
  // coarse-operator construction (coarse.h): out = H_dir in (full fields, hopping normalisation included), out = L in.
  virtual void hopDir(ColorSpinorField &out, const ColorSpinorField &in, int dir) const;
  virtual void localTerm(ColorSpinorField &out, const ColorSpinorField &in) const;
  // the site-diagonal term A_pp of the operator on ONE parity (1, 1 + i a g5, clover + i a g5): the full-system residual behind an even-odd
  // preconditioned smoother is (A r_hat, 0) for the symmetric preconditioning, see MG::cycleParity; out = in allowed
  virtual void localTermParity(ColorSpinorField &out, const ColorSpinorField &in, int parity) const;

  void setMass(double m) { mass = m; }
  double Kappa() const { return kappa; }
  virtual double Mu() const { return 0.; }
  QudaMatPCType getMatPCType() const { return matpcType; }
  QudaDiracType getDiracType() const { return type; }
  bool isPC() const;
  void Dagger(QudaDagType d) const { dagger = d; }
  void flipDagger() const { dagger = dagger == QUDA_DAG_YES ? QUDA_DAG_NO : QUDA_DAG_YES; }
  unsigned long long Flops() const { unsigned long long r = flops; flops = 0; return r; }
  GaugeField *Gauge() const { return gauge; }
  virtual CloverField *Clover() const { return nullptr; }

  static Dirac *create(const DiracParam &param);  // reference lib/dirac.cpp:140
};

// ---- Wilson ----
class DiracWilson : public Dirac {
 public:
  explicit DiracWilson(const DiracParam &p) : Dirac(p) {}
  void Dslash(ColorSpinorField &out, const ColorSpinorField &in, const QudaParity parity) const override;
  void DslashXpay(ColorSpinorField &out, const ColorSpinorField &in, const QudaParity parity, const ColorSpinorField &x, const double &k) const override;
  void M(ColorSpinorField &out, const ColorSpinorField &in) const override;
  void MdagM(ColorSpinorField &out, const ColorSpinorField &in) const override;
  void prepare(ColorSpinorField *&src, ColorSpinorField *&sol, ColorSpinorField &x, ColorSpinorField &b, const QudaSolutionType) const override;
  void reconstruct(ColorSpinorField &x, const ColorSpinorField &b, const QudaSolutionType) const override;
  void hopDir(ColorSpinorField &out, const ColorSpinorField &in, int dir) const override;
  void localTerm(ColorSpinorField &out, const ColorSpinorField &in) const override;
  void localTermParity(ColorSpinorField &out, const ColorSpinorField &in, int parity) const override;
};

class DiracWilsonPC : public DiracWilson {
 public:
  explicit DiracWilsonPC(const DiracParam &p) : DiracWilson(p) {}
  void M(ColorSpinorField &out, const ColorSpinorField &in) const override;
  void MdagM(ColorSpinorField &out, const ColorSpinorField &in) const override;
  void prepare(ColorSpinorField *&src, ColorSpinorField *&sol, ColorSpinorField &x, ColorSpinorField &b, const QudaSolutionType) const override;
  void reconstruct(ColorSpinorField &x, const ColorSpinorField &b, const QudaSolutionType) const override;
};

// ---- twisted mass (degenerate; the non-degenerate doublet is out of scope) ----
class DiracTwistedMass : public DiracWilson {
 protected:
  double mu, epsilon;
  void twistedApply(ColorSpinorField &out, const ColorSpinorField &in, QudaTwistGamma5Type twistType) const;
  void checkFlavor(const ColorSpinorField &out, const ColorSpinorField &in) const;

 public:
  explicit DiracTwistedMass(const DiracParam &p) : DiracWilson(p), mu(p.mu), epsilon(p.epsilon) {}
  double Mu() const override { return mu; }
  void Twist(ColorSpinorField &out, const ColorSpinorField &in) const;
  void localTerm(ColorSpinorField &out, const ColorSpinorField &in) const override;
  void localTermParity(ColorSpinorField &out, const ColorSpinorField &in, int parity) const override;
  void TwistedDslash(ColorSpinorField &out, const ColorSpinorField &in, QudaParity parity, QudaTwistDslashType t, double a, double b) const;
  void TwistedDslashXpay(ColorSpinorField &out, const ColorSpinorField &in, const ColorSpinorField &x, QudaParity parity,
                         QudaTwistDslashType t, double a, double b) const;
  void M(ColorSpinorField &out, const ColorSpinorField &in) const override;
  void MdagM(ColorSpinorField &out, const ColorSpinorField &in) const override;
  void prepare(ColorSpinorField *&src, ColorSpinorField *&sol, ColorSpinorField &x, ColorSpinorField &b, const QudaSolutionType) const override;
  void reconstruct(ColorSpinorField &x, const ColorSpinorField &b, const QudaSolutionType) const override;
};

class DiracTwistedMassPC : public DiracTwistedMass {
 public:
  explicit DiracTwistedMassPC(const DiracParam &p) : DiracTwistedMass(p) {}
  void TwistInv(ColorSpinorField &out, const ColorSpinorField &in) const;
  void Dslash(ColorSpinorField &out, const ColorSpinorField &in, const QudaParity parity) const override;
  void DslashXpay(ColorSpinorField &out, const ColorSpinorField &in, const QudaParity parity, const ColorSpinorField &x, const double &k) const override;
  void M(ColorSpinorField &out, const ColorSpinorField &in) const override;
  void MdagM(ColorSpinorField &out, const ColorSpinorField &in) const override;
  void prepare(ColorSpinorField *&src, ColorSpinorField *&sol, ColorSpinorField &x, ColorSpinorField &b, const QudaSolutionType) const override;
  void reconstruct(ColorSpinorField &x, const ColorSpinorField &b, const QudaSolutionType) const override;
};

// ---- twisted clover ----
class DiracTwistedClover : public DiracWilson {
 protected:
  double mu, epsilon;
  CloverField &clover;
  void twistedCloverApply(ColorSpinorField &out, const ColorSpinorField &in, QudaTwistGamma5Type twistType, int parity) const;
  void tcDslash(ColorSpinorField &out, const ColorSpinorField &in, int parity, const ColorSpinorField *x, QudaTwistCloverDslashType t,
                double a, double b) const;

 public:
  explicit DiracTwistedClover(const DiracParam &p);
  double Mu() const override { return mu; }
  CloverField *Clover() const override { return &clover; }
  void TwistClover(ColorSpinorField &out, const ColorSpinorField &in, const int parity) const;
  void localTerm(ColorSpinorField &out, const ColorSpinorField &in) const override;
  void localTermParity(ColorSpinorField &out, const ColorSpinorField &in, int parity) const override;
  void M(ColorSpinorField &out, const ColorSpinorField &in) const override;
  void MdagM(ColorSpinorField &out, const ColorSpinorField &in) const override;
  void prepare(ColorSpinorField *&src, ColorSpinorField *&sol, ColorSpinorField &x, ColorSpinorField &b, const QudaSolutionType) const override;
  void reconstruct(ColorSpinorField &x, const ColorSpinorField &b, const QudaSolutionType) const override;
};

class DiracTwistedCloverPC : public DiracTwistedClover {
 public:
  explicit DiracTwistedCloverPC(const DiracParam &p) : DiracTwistedClover(p) {}
  void TwistCloverInv(ColorSpinorField &out, const ColorSpinorField &in, const int parity) const;
  void Dslash(ColorSpinorField &out, const ColorSpinorField &in, const QudaParity parity) const override;
  void DslashXpay(ColorSpinorField &out, const ColorSpinorField &in, const QudaParity parity, const ColorSpinorField &x, const double &k) const override;
  void M(ColorSpinorField &out, const ColorSpinorField &in) const override;
  void MdagM(ColorSpinorField &out, const ColorSpinorField &in) const override;
  void prepare(ColorSpinorField *&src, ColorSpinorField *&sol, ColorSpinorField &x, ColorSpinorField &b, const QudaSolutionType) const override;
  void reconstruct(ColorSpinorField &x, const ColorSpinorField &b, const QudaSolutionType) const override;
};

// ---- matrix functors (reference include/dirac_quda.h:868-1032) ----
class DiracMatrix {
 protected:
  const Dirac *dirac;
 public:
  explicit DiracMatrix(const Dirac *d) : dirac(d) {}
  explicit DiracMatrix(const Dirac &d) : dirac(&d) {}
  virtual ~DiracMatrix() {}
  virtual void operator()(ColorSpinorField &out, const ColorSpinorField &in) const = 0;
  unsigned long long flops() const { return dirac->Flops(); }
  bool isPC() const { return dirac->isPC(); }
  const Dirac *Expose() const { return dirac; }
  virtual bool isM() const { return false; }   // the operator itself (not M^dag M, ...)
};
class DiracM : public DiracMatrix {
 public:
  using DiracMatrix::DiracMatrix;
  bool isM() const override { return true; }
  void operator()(ColorSpinorField &out, const ColorSpinorField &in) const override { dirac->M(out, in); }
};
class DiracMdagM : public DiracMatrix {
 public:
  using DiracMatrix::DiracMatrix;
  void operator()(ColorSpinorField &out, const ColorSpinorField &in) const override { dirac->MdagM(out, in); }
};
class DiracMdag : public DiracMatrix {
 public:
  using DiracMatrix::DiracMatrix;
  void operator()(ColorSpinorField &out, const ColorSpinorField &in) const override { dirac->Mdag(out, in); }
};
class DiracMMdag : public DiracMatrix {
 public:
  using DiracMatrix::DiracMatrix;
  void operator()(ColorSpinorField &out, const ColorSpinorField &in) const override { dirac->MMdag(out, in); }
};

}  // namespace quda
