// dirac.cpp — Wilson / twisted-mass / twisted-clover operator classes.
// Semantics restated from the reference: lib/dirac.cpp:60-140, lib/dirac_wilson.cpp, lib/dirac_twisted_mass.cpp:40-584,
// lib/dirac_twisted_clover.cpp:40-430 (operator algebra, coefficient conventions, flop counters).
#include "dirac.h"

#include "blas.h"
#include "coarse.h"

namespace quda {

Dirac::Dirac(const DiracParam &p)
    : gauge(p.gauge), kappa(p.kappa), mass(p.mass), matpcType(p.matpcType), dagger(p.dagger), flops(0), tmp1(p.tmp1), tmp2(p.tmp2),
      own1(false), own2(false), type(p.type) {
  for (int i = 0; i < QUDA_MAX_DIM; i++) commDim[i] = p.commDim[i];
}

Dirac::~Dirac() {
  if (own1) delete tmp1;
  if (own2) delete tmp2;
}

ColorSpinorField *Dirac::getTmp(ColorSpinorField *&slot, bool &own, const ColorSpinorField &like) const {
  if (slot && (slot->VolumeCB() != like.VolumeCB() || slot->Precision() != like.Precision() || slot->SiteSubset() != like.SiteSubset() ||
               slot->Ncolor() != like.Ncolor() || slot->Nspin() != like.Nspin())) {
    if (own) { delete slot; slot = nullptr; own = false; }
    else errorQuda("caller-supplied temporary does not match the operand geometry");
  }
  if (!slot) {
    ColorSpinorParam p = like.param();
    p.location = QUDA_CUDA_FIELD_LOCATION;
    p.create = QUDA_NULL_FIELD_CREATE;
    slot = new ColorSpinorField(p);
    own = true;
  }
  slot->twistFlavor = like.twistFlavor;
  return slot;
}

void Dirac::checkParitySpinor(const ColorSpinorField &a, const ColorSpinorField &b) const {
  if (a.SiteSubset() != QUDA_PARITY_SITE_SUBSET || b.SiteSubset() != QUDA_PARITY_SITE_SUBSET) errorQuda("parity spinors required");
  if (a.V() == b.V()) errorQuda("aliasing pointers");
}
void Dirac::checkFullSpinor(const ColorSpinorField &a, const ColorSpinorField &b) const {
  if (a.SiteSubset() != QUDA_FULL_SITE_SUBSET || b.SiteSubset() != QUDA_FULL_SITE_SUBSET) errorQuda("full spinors required");
}

void Dirac::Mdag(ColorSpinorField &out, const ColorSpinorField &in) const {
  flipDagger();
  M(out, in);
  flipDagger();
}
void Dirac::MMdag(ColorSpinorField &out, const ColorSpinorField &in) const {
  flipDagger();
  MdagM(out, in);
  flipDagger();
}

void Dirac::hopDir(ColorSpinorField &, const ColorSpinorField &, int) const { errorQuda("hopDir not available for Dirac type %d", type); }
void Dirac::localTerm(ColorSpinorField &, const ColorSpinorField &) const { errorQuda("localTerm not available for Dirac type %d", type); }
void Dirac::localTermParity(ColorSpinorField &, const ColorSpinorField &, int) const { errorQuda("localTermParity not available for Dirac type %d", type); }

bool Dirac::isPC() const {
  return type == QUDA_WILSONPC_DIRAC || type == QUDA_TWISTED_MASSPC_DIRAC || type == QUDA_TWISTED_CLOVERPC_DIRAC || type == QUDA_COARSEPC_DIRAC ||
         type == QUDA_CLOVERPC_DIRAC;
}

Dirac *Dirac::create(const DiracParam &param) {
  switch (param.type) {
    case QUDA_WILSON_DIRAC: return new DiracWilson(param);
    case QUDA_WILSONPC_DIRAC: return new DiracWilsonPC(param);
    case QUDA_TWISTED_MASS_DIRAC: return new DiracTwistedMass(param);
    case QUDA_TWISTED_MASSPC_DIRAC: return new DiracTwistedMassPC(param);
    case QUDA_TWISTED_CLOVER_DIRAC: return new DiracTwistedClover(param);
    case QUDA_TWISTED_CLOVERPC_DIRAC: return new DiracTwistedCloverPC(param);
    case QUDA_COARSE_DIRAC: return new DiracCoarse(param);
    default: errorQuda("Dirac type %d is outside the twisted-mass/multigrid path this library implements", param.type);
  }
  return nullptr;
}

// ================================================================================================
// Wilson
// ================================================================================================
void DiracWilson::Dslash(ColorSpinorField &out, const ColorSpinorField &in, const QudaParity parity) const {
  checkParitySpinor(in, out);
  DslashParam p;
  p.mode = DSLASH_PLAIN; p.parity = parity; p.dagger = dagger == QUDA_DAG_YES;
  applyDslash(out, in, *gauge, p);
  flops += 1320ll * in.Volume();
}

void DiracWilson::DslashXpay(ColorSpinorField &out, const ColorSpinorField &in, const QudaParity parity, const ColorSpinorField &x,
                             const double &k) const {
  checkParitySpinor(in, out);
  DslashParam p;
  p.mode = DSLASH_PLAIN; p.parity = parity; p.dagger = dagger == QUDA_DAG_YES; p.x = &x; p.k = k;
  applyDslash(out, in, *gauge, p);
  flops += 1368ll * in.Volume();
}

void DiracWilson::M(ColorSpinorField &out, const ColorSpinorField &in) const {
  checkFullSpinor(out, in);
  DslashXpay(out.Odd(), in.Even(), QUDA_ODD_PARITY, in.Odd(), -kappa);
  DslashXpay(out.Even(), in.Odd(), QUDA_EVEN_PARITY, in.Even(), -kappa);
}

void DiracWilson::MdagM(ColorSpinorField &out, const ColorSpinorField &in) const {
  ColorSpinorField *t = getTmp(tmp1, own1, in);
  M(*t, in);
  Mdag(out, *t);
}

// M = 1 - kappa D: H_d = -kappa (1 -+ gamma_mu) U, L = 1
void DiracWilson::hopDir(ColorSpinorField &out, const ColorSpinorField &in, int dir) const {
  checkFullSpinor(out, in);
  applyHopDir(out.Odd(), in.Even(), *gauge, QUDA_ODD_PARITY, dir, -kappa);
  applyHopDir(out.Even(), in.Odd(), *gauge, QUDA_EVEN_PARITY, dir, -kappa);
}
void DiracWilson::localTerm(ColorSpinorField &out, const ColorSpinorField &in) const { blas::copy(out, in); }
void DiracWilson::localTermParity(ColorSpinorField &out, const ColorSpinorField &in, int) const { if (out.V() != in.V()) blas::copy(out, in); }

void DiracWilson::prepare(ColorSpinorField *&src, ColorSpinorField *&sol, ColorSpinorField &x, ColorSpinorField &b,
                          const QudaSolutionType solType) const {
  if (solType == QUDA_MATPC_SOLUTION || solType == QUDA_MATPCDAG_MATPC_SOLUTION) errorQuda("Preconditioned solution requires a preconditioned solve_type");
  src = &b;
  sol = &x;
}
void DiracWilson::reconstruct(ColorSpinorField &, const ColorSpinorField &, const QudaSolutionType) const {}

void DiracWilsonPC::M(ColorSpinorField &out, const ColorSpinorField &in) const {
  const double kappa2 = -kappa * kappa;
  ColorSpinorField *t = getTmp(tmp1, own1, in);
  if (matpcType == QUDA_MATPC_EVEN_EVEN) {
    Dslash(*t, in, QUDA_ODD_PARITY);
    DslashXpay(out, *t, QUDA_EVEN_PARITY, in, kappa2);
  } else if (matpcType == QUDA_MATPC_ODD_ODD) {
    Dslash(*t, in, QUDA_EVEN_PARITY);
    DslashXpay(out, *t, QUDA_ODD_PARITY, in, kappa2);
  } else {
    errorQuda("MatPCType %d not valid for DiracWilsonPC", matpcType);
  }
}
void DiracWilsonPC::MdagM(ColorSpinorField &out, const ColorSpinorField &in) const {
  ColorSpinorField *t = getTmp(tmp2, own2, in);
  M(*t, in);
  Mdag(out, *t);
}
void DiracWilsonPC::prepare(ColorSpinorField *&src, ColorSpinorField *&sol, ColorSpinorField &x, ColorSpinorField &b,
                            const QudaSolutionType solType) const {
  if (solType == QUDA_MATPC_SOLUTION || solType == QUDA_MATPCDAG_MATPC_SOLUTION) { src = &b; sol = &x; return; }
  if (matpcType == QUDA_MATPC_EVEN_EVEN) {  // src = b_e + k D_eo b_o
    src = &(x.Odd());
    DiracWilson::DslashXpay(*src, b.Odd(), QUDA_EVEN_PARITY, b.Even(), kappa);
    sol = &(x.Even());
  } else if (matpcType == QUDA_MATPC_ODD_ODD) {
    src = &(x.Even());
    DiracWilson::DslashXpay(*src, b.Even(), QUDA_ODD_PARITY, b.Odd(), kappa);
    sol = &(x.Odd());
  } else {
    errorQuda("MatPCType %d not valid for DiracWilsonPC", matpcType);
  }
}
void DiracWilsonPC::reconstruct(ColorSpinorField &x, const ColorSpinorField &b, const QudaSolutionType solType) const {
  if (solType == QUDA_MATPC_SOLUTION || solType == QUDA_MATPCDAG_MATPC_SOLUTION) return;
  checkFullSpinor(x, b);
  if (matpcType == QUDA_MATPC_EVEN_EVEN) DiracWilson::DslashXpay(x.Odd(), x.Even(), QUDA_ODD_PARITY, b.Odd(), kappa);
  else if (matpcType == QUDA_MATPC_ODD_ODD) DiracWilson::DslashXpay(x.Even(), x.Odd(), QUDA_EVEN_PARITY, b.Even(), kappa);
  else errorQuda("MatPCType %d not valid for DiracWilsonPC", matpcType);
}

// ================================================================================================
// Twisted mass
// ================================================================================================
void DiracTwistedMass::checkFlavor(const ColorSpinorField &out, const ColorSpinorField &in) const {
  if (in.TwistFlavor() != out.TwistFlavor()) errorQuda("Twist flavors %d %d don't match", in.TwistFlavor(), out.TwistFlavor());
  if (in.TwistFlavor() != QUDA_TWIST_PLUS && in.TwistFlavor() != QUDA_TWIST_MINUS)
    errorQuda("Twist flavor %d not supported (degenerate +-1 only)", in.TwistFlavor());
}

// reference lib/dirac_twisted_mass.cpp:47-79 + setTwistParam lib/dslash_constants.h:544-557
void DiracTwistedMass::twistedApply(ColorSpinorField &out, const ColorSpinorField &in, QudaTwistGamma5Type twistType) const {
  checkFlavor(out, in);
  const double fmu = in.TwistFlavor() * mu;
  double a, b;
  if (twistType == QUDA_TWIST_GAMMA5_DIRECT) { a = 2.0 * kappa * fmu; b = 1.0; }
  else { a = -2.0 * kappa * fmu; b = 1.0 / (1.0 + a * a); }
  if (dagger == QUDA_DAG_YES) a = -a;
  if (in.SiteSubset() == QUDA_FULL_SITE_SUBSET) {
    applySite(out.Even(), in.Even(), SITE_TWIST, a, b, nullptr, 0, false);
    applySite(out.Odd(), in.Odd(), SITE_TWIST, a, b, nullptr, 1, false);
  } else {
    applySite(out, in, SITE_TWIST, a, b, nullptr, 0, false);
  }
  flops += 48ll * in.Volume();
}
void DiracTwistedMass::Twist(ColorSpinorField &out, const ColorSpinorField &in) const { twistedApply(out, in, QUDA_TWIST_GAMMA5_DIRECT); }
void DiracTwistedMass::localTerm(ColorSpinorField &out, const ColorSpinorField &in) const { Twist(out, in); }  // L = 1 + i a gamma5
void DiracTwistedMass::localTermParity(ColorSpinorField &out, const ColorSpinorField &in, int) const { Twist(out, in); }

static DslashMode tmMode(QudaTwistDslashType t) {
  switch (t) {
    case QUDA_DEG_TWIST_INV_DSLASH: return DSLASH_TWIST_INV_DSLASH;
    case QUDA_DEG_DSLASH_TWIST_INV: return DSLASH_TWIST_INV;
    case QUDA_DEG_DSLASH_TWIST_XPAY: return DSLASH_TWIST_XPAY;
    default: errorQuda("twist dslash type %d not supported", t);
  }
  return DSLASH_PLAIN;
}

// a is given un-daggered (the kernel wrapper flips it), b multiplies the twisted result (or is k for TWIST_XPAY):
// reference twistedMassDslashCuda lib/dslash_twisted_mass.cu:167-230 and core epilogues
void DiracTwistedMass::TwistedDslash(ColorSpinorField &out, const ColorSpinorField &in, QudaParity parity, QudaTwistDslashType t, double a,
                                     double b) const {
  DslashParam p;
  p.mode = tmMode(t); p.parity = parity; p.dagger = dagger == QUDA_DAG_YES;
  p.a = p.dagger ? -a : a; p.b = b;
  applyDslash(out, in, *gauge, p);
}
void DiracTwistedMass::TwistedDslashXpay(ColorSpinorField &out, const ColorSpinorField &in, const ColorSpinorField &x, QudaParity parity,
                                         QudaTwistDslashType t, double a, double b) const {
  DslashParam p;
  p.mode = tmMode(t); p.parity = parity; p.dagger = dagger == QUDA_DAG_YES;
  p.a = p.dagger ? -a : a; p.b = b; p.k = b; p.x = &x;
  applyDslash(out, in, *gauge, p);
}

void DiracTwistedMass::M(ColorSpinorField &out, const ColorSpinorField &in) const {
  checkFullSpinor(out, in);
  checkFlavor(out, in);
  const double a = 2.0 * kappa * in.TwistFlavor() * mu;
  TwistedDslashXpay(out.Odd(), in.Even(), in.Odd(), QUDA_ODD_PARITY, QUDA_DEG_DSLASH_TWIST_XPAY, a, -kappa);
  TwistedDslashXpay(out.Even(), in.Odd(), in.Even(), QUDA_EVEN_PARITY, QUDA_DEG_DSLASH_TWIST_XPAY, a, -kappa);
  flops += (1320ll + 72ll) * in.Volume();
}
void DiracTwistedMass::MdagM(ColorSpinorField &out, const ColorSpinorField &in) const {
  checkFullSpinor(out, in);
  ColorSpinorField *t = getTmp(tmp1, own1, in);
  M(*t, in);
  Mdag(out, *t);
}
void DiracTwistedMass::prepare(ColorSpinorField *&src, ColorSpinorField *&sol, ColorSpinorField &x, ColorSpinorField &b,
                               const QudaSolutionType solType) const {
  if (solType == QUDA_MATPC_SOLUTION || solType == QUDA_MATPCDAG_MATPC_SOLUTION) errorQuda("Preconditioned solution requires a preconditioned solve_type");
  src = &b;
  sol = &x;
}
void DiracTwistedMass::reconstruct(ColorSpinorField &, const ColorSpinorField &, const QudaSolutionType) const {}

void DiracTwistedMassPC::TwistInv(ColorSpinorField &out, const ColorSpinorField &in) const { twistedApply(out, in, QUDA_TWIST_GAMMA5_INVERSE); }

// (A^-1 D) or, for dagger with symmetric preconditioning, (D^dag A^-1^dag): reference :238-287
void DiracTwistedMassPC::Dslash(ColorSpinorField &out, const ColorSpinorField &in, const QudaParity parity) const {
  checkParitySpinor(in, out);
  checkFlavor(out, in);
  const double a = -2.0 * kappa * in.TwistFlavor() * mu;
  const double b = 1.0 / (1.0 + a * a);
  const bool asym = matpcType == QUDA_MATPC_EVEN_EVEN_ASYMMETRIC || matpcType == QUDA_MATPC_ODD_ODD_ASYMMETRIC;
  if (dagger == QUDA_DAG_NO || asym) TwistedDslash(out, in, parity, QUDA_DEG_DSLASH_TWIST_INV, a, b);
  else TwistedDslash(out, in, parity, QUDA_DEG_TWIST_INV_DSLASH, a, b);
  flops += 1392ll * in.Volume();
}

void DiracTwistedMassPC::DslashXpay(ColorSpinorField &out, const ColorSpinorField &in, const QudaParity parity, const ColorSpinorField &x,
                                    const double &k) const {
  checkParitySpinor(in, out);
  checkFlavor(out, in);
  const double a = -2.0 * kappa * in.TwistFlavor() * mu;
  const double b = k / (1.0 + a * a);
  if (dagger == QUDA_DAG_NO) TwistedDslashXpay(out, in, x, parity, QUDA_DEG_DSLASH_TWIST_INV, a, b);
  else TwistedDslashXpay(out, in, x, parity, QUDA_DEG_TWIST_INV_DSLASH, a, b);
  flops += 1416ll * in.Volume();
}

void DiracTwistedMassPC::M(ColorSpinorField &out, const ColorSpinorField &in) const {
  const double kappa2 = -kappa * kappa;
  checkFlavor(out, in);
  ColorSpinorField *t = getTmp(tmp1, own1, in);
  if (matpcType == QUDA_MATPC_EVEN_EVEN) {
    Dslash(*t, in, QUDA_ODD_PARITY);
    DslashXpay(out, *t, QUDA_EVEN_PARITY, in, kappa2);
  } else if (matpcType == QUDA_MATPC_ODD_ODD) {
    Dslash(*t, in, QUDA_EVEN_PARITY);
    DslashXpay(out, *t, QUDA_ODD_PARITY, in, kappa2);
  } else {
    const double a = 2.0 * kappa * in.TwistFlavor() * mu;
    if (matpcType == QUDA_MATPC_EVEN_EVEN_ASYMMETRIC) {
      Dslash(*t, in, QUDA_ODD_PARITY);
      TwistedDslashXpay(out, *t, in, QUDA_EVEN_PARITY, QUDA_DEG_DSLASH_TWIST_XPAY, a, kappa2);
    } else if (matpcType == QUDA_MATPC_ODD_ODD_ASYMMETRIC) {
      Dslash(*t, in, QUDA_EVEN_PARITY);
      TwistedDslashXpay(out, *t, in, QUDA_ODD_PARITY, QUDA_DEG_DSLASH_TWIST_XPAY, a, kappa2);
    } else {
      errorQuda("Invalid matpcType");
    }
    flops += (1320ll + 96ll) * in.Volume();
  }
}

void DiracTwistedMassPC::MdagM(ColorSpinorField &out, const ColorSpinorField &in) const {
  ColorSpinorField *t = getTmp(tmp2, own2, in);
  M(*t, in);
  Mdag(out, *t);
}

// Schur-complement source / solution: reference :409-578
void DiracTwistedMassPC::prepare(ColorSpinorField *&src, ColorSpinorField *&sol, ColorSpinorField &x, ColorSpinorField &b,
                                 const QudaSolutionType solType) const {
  if (solType == QUDA_MATPC_SOLUTION || solType == QUDA_MATPCDAG_MATPC_SOLUTION) { src = &b; sol = &x; return; }
  checkFullSpinor(x, b);
  ColorSpinorField *t = getTmp(tmp1, own1, b.Even());
  if (matpcType == QUDA_MATPC_EVEN_EVEN) {  // src = A_ee^-1 (b_e + k D_eo A_oo^-1 b_o)
    src = &(x.Odd());
    TwistInv(*src, b.Odd());
    DiracWilson::DslashXpay(*t, *src, QUDA_EVEN_PARITY, b.Even(), kappa);
    TwistInv(*src, *t);
    sol = &(x.Even());
  } else if (matpcType == QUDA_MATPC_ODD_ODD) {
    src = &(x.Even());
    TwistInv(*src, b.Even());
    DiracWilson::DslashXpay(*t, *src, QUDA_ODD_PARITY, b.Odd(), kappa);
    TwistInv(*src, *t);
    sol = &(x.Odd());
  } else if (matpcType == QUDA_MATPC_EVEN_EVEN_ASYMMETRIC) {  // src = b_e + k D_eo A_oo^-1 b_o
    src = &(x.Odd());
    TwistInv(*t, b.Odd());
    DiracWilson::DslashXpay(*src, *t, QUDA_EVEN_PARITY, b.Even(), kappa);
    sol = &(x.Even());
  } else if (matpcType == QUDA_MATPC_ODD_ODD_ASYMMETRIC) {
    src = &(x.Even());
    TwistInv(*t, b.Even());
    DiracWilson::DslashXpay(*src, *t, QUDA_ODD_PARITY, b.Odd(), kappa);
    sol = &(x.Odd());
  } else {
    errorQuda("MatPCType %d not valid for DiracTwistedMassPC", matpcType);
  }
}

void DiracTwistedMassPC::reconstruct(ColorSpinorField &x, const ColorSpinorField &b, const QudaSolutionType solType) const {
  if (solType == QUDA_MATPC_SOLUTION || solType == QUDA_MATPCDAG_MATPC_SOLUTION) return;
  checkFullSpinor(x, b);
  ColorSpinorField *t = getTmp(tmp1, own1, b.Even());
  if (matpcType == QUDA_MATPC_EVEN_EVEN || matpcType == QUDA_MATPC_EVEN_EVEN_ASYMMETRIC) {  // x_o = A_oo^-1 (b_o + k D_oe x_e)
    DiracWilson::DslashXpay(*t, x.Even(), QUDA_ODD_PARITY, b.Odd(), kappa);
    TwistInv(x.Odd(), *t);
  } else if (matpcType == QUDA_MATPC_ODD_ODD || matpcType == QUDA_MATPC_ODD_ODD_ASYMMETRIC) {
    DiracWilson::DslashXpay(*t, x.Odd(), QUDA_EVEN_PARITY, b.Even(), kappa);
    TwistInv(x.Even(), *t);
  } else {
    errorQuda("MatPCType %d not valid for DiracTwistedMassPC", matpcType);
  }
}

// ================================================================================================
// Twisted clover
// ================================================================================================
DiracTwistedClover::DiracTwistedClover(const DiracParam &p) : DiracWilson(p), mu(p.mu), epsilon(p.epsilon), clover(*p.clover) {
  if (!p.clover) errorQuda("twisted-clover operator without a clover field");
}

// reference :50-86 and twistCloverGamma5Cuda lib/dslash_quda.cu:562
void DiracTwistedClover::twistedCloverApply(ColorSpinorField &out, const ColorSpinorField &in, QudaTwistGamma5Type twistType, int parity) const {
  if (in.TwistFlavor() != QUDA_TWIST_PLUS && in.TwistFlavor() != QUDA_TWIST_MINUS) errorQuda("Twist flavor %d not supported", in.TwistFlavor());
  const double fmu = in.TwistFlavor() * mu;
  double a = twistType == QUDA_TWIST_GAMMA5_DIRECT ? 2.0 * kappa * fmu : -2.0 * kappa * fmu;
  if (dagger == QUDA_DAG_YES) a = -a;
  applySite(out, in, twistType == QUDA_TWIST_GAMMA5_DIRECT ? SITE_CLOVER_TWIST : SITE_CLOVER_TWIST_INV, a, 1.0, &clover, parity, false);
  flops += (twistType == QUDA_TWIST_GAMMA5_INVERSE ? 1056ll : 552ll) * in.Volume();
}
void DiracTwistedClover::TwistClover(ColorSpinorField &out, const ColorSpinorField &in, const int parity) const {
  twistedCloverApply(out, in, QUDA_TWIST_GAMMA5_DIRECT, parity);
}

void DiracTwistedClover::localTermParity(ColorSpinorField &out, const ColorSpinorField &in, int parity) const { TwistClover(out, in, parity); }
void DiracTwistedClover::localTerm(ColorSpinorField &out, const ColorSpinorField &in) const {  // L = A + i a gamma5
  checkFullSpinor(out, in);
  TwistClover(out.Even(), in.Even(), QUDA_EVEN_PARITY);
  TwistClover(out.Odd(), in.Odd(), QUDA_ODD_PARITY);
}

void DiracTwistedClover::tcDslash(ColorSpinorField &out, const ColorSpinorField &in, int parity, const ColorSpinorField *x,
                                  QudaTwistCloverDslashType t, double a, double b) const {
  DslashParam p;
  p.parity = parity; p.dagger = dagger == QUDA_DAG_YES; p.a = p.dagger ? -a : a; p.b = b; p.k = b; p.x = x; p.clover = &clover;
  switch (t) {
    case QUDA_DEG_DSLASH_CLOVER_TWIST_INV: p.mode = DSLASH_CLOVER_TWIST_INV; break;
    case QUDA_DEG_DSLASH_CLOVER_TWIST_XPAY: p.mode = DSLASH_CLOVER_TWIST_XPAY; break;
    case QUDA_DEG_CLOVER_TWIST_INV_DSLASH: p.mode = DSLASH_PLAIN; p.clover = nullptr; break;  // clover-twist applied by the caller (reference M :282-287)
    default: errorQuda("bad twisted-clover dslash type %d", t);
  }
  applyDslash(out, in, *gauge, p);
}

void DiracTwistedClover::M(ColorSpinorField &out, const ColorSpinorField &in) const {
  checkFullSpinor(out, in);
  if (in.TwistFlavor() != QUDA_TWIST_PLUS && in.TwistFlavor() != QUDA_TWIST_MINUS) errorQuda("Twist flavor not set %d", in.TwistFlavor());
  const double a = 2.0 * kappa * in.TwistFlavor() * mu;
  tcDslash(out.Odd(), in.Even(), QUDA_ODD_PARITY, &in.Odd(), QUDA_DEG_DSLASH_CLOVER_TWIST_XPAY, a, -kappa);
  tcDslash(out.Even(), in.Odd(), QUDA_EVEN_PARITY, &in.Even(), QUDA_DEG_DSLASH_CLOVER_TWIST_XPAY, a, -kappa);
  flops += (1320ll + 552ll) * in.Volume();
}
void DiracTwistedClover::MdagM(ColorSpinorField &out, const ColorSpinorField &in) const {
  checkFullSpinor(out, in);
  ColorSpinorField *t = getTmp(tmp1, own1, in);
  M(*t, in);
  Mdag(out, *t);
}
void DiracTwistedClover::prepare(ColorSpinorField *&src, ColorSpinorField *&sol, ColorSpinorField &x, ColorSpinorField &b,
                                 const QudaSolutionType solType) const {
  if (solType == QUDA_MATPC_SOLUTION || solType == QUDA_MATPCDAG_MATPC_SOLUTION) errorQuda("Preconditioned solution requires a preconditioned solve_type");
  src = &b;
  sol = &x;
}
void DiracTwistedClover::reconstruct(ColorSpinorField &, const ColorSpinorField &, const QudaSolutionType) const {}

void DiracTwistedCloverPC::TwistCloverInv(ColorSpinorField &out, const ColorSpinorField &in, const int parity) const {
  twistedCloverApply(out, in, QUDA_TWIST_GAMMA5_INVERSE, parity);
}

void DiracTwistedCloverPC::Dslash(ColorSpinorField &out, const ColorSpinorField &in, const QudaParity parity) const {
  checkParitySpinor(in, out);
  const double a = -2.0 * kappa * in.TwistFlavor() * mu;
  const bool asym = matpcType == QUDA_MATPC_EVEN_EVEN_ASYMMETRIC || matpcType == QUDA_MATPC_ODD_ODD_ASYMMETRIC;
  if (dagger == QUDA_DAG_NO || asym) {
    tcDslash(out, in, parity, nullptr, QUDA_DEG_DSLASH_CLOVER_TWIST_INV, a, 1.0);
    flops += 2376ll * in.Volume();
  } else {
    tcDslash(out, in, parity, nullptr, QUDA_DEG_CLOVER_TWIST_INV_DSLASH, a, 1.0);
    flops += 1320ll * in.Volume();
  }
}

void DiracTwistedCloverPC::DslashXpay(ColorSpinorField &out, const ColorSpinorField &in, const QudaParity parity, const ColorSpinorField &x,
                                      const double &k) const {
  checkParitySpinor(in, out);
  const double a = -2.0 * kappa * in.TwistFlavor() * mu;
  if (dagger == QUDA_DAG_NO) {
    tcDslash(out, in, parity, &x, QUDA_DEG_DSLASH_CLOVER_TWIST_INV, a, k);
    flops += 2400ll * in.Volume();
  } else {
    // out = x + k D in  (clover-twist inverse applied by the caller)
    DslashParam p;
    p.mode = DSLASH_PLAIN; p.parity = parity; p.dagger = 1; p.x = &x; p.k = k;
    applyDslash(out, in, *gauge, p);
    flops += 1344ll * in.Volume();
  }
}

void DiracTwistedCloverPC::M(ColorSpinorField &out, const ColorSpinorField &in) const {
  const double kappa2 = -kappa * kappa;
  ColorSpinorField *t = getTmp(tmp1, own1, in);
  const bool sym = matpcType == QUDA_MATPC_EVEN_EVEN || matpcType == QUDA_MATPC_ODD_ODD;
  const QudaParity p0 = (matpcType == QUDA_MATPC_EVEN_EVEN || matpcType == QUDA_MATPC_EVEN_EVEN_ASYMMETRIC) ? QUDA_EVEN_PARITY : QUDA_ODD_PARITY;
  const QudaParity p1 = p0 == QUDA_EVEN_PARITY ? QUDA_ODD_PARITY : QUDA_EVEN_PARITY;
  if (sym) {
    if (dagger == QUDA_DAG_YES) {
      TwistCloverInv(*t, in, p0);
      Dslash(out, *t, p1);
      TwistCloverInv(*t, out, p1);
      DslashXpay(out, *t, p0, in, kappa2);
    } else {
      Dslash(*t, in, p1);
      DslashXpay(out, *t, p0, in, kappa2);
    }
  } else if (matpcType == QUDA_MATPC_EVEN_EVEN_ASYMMETRIC || matpcType == QUDA_MATPC_ODD_ODD_ASYMMETRIC) {
    const double a = 2.0 * kappa * in.TwistFlavor() * mu;
    Dslash(*t, in, p1);
    tcDslash(out, *t, p0, &in, QUDA_DEG_DSLASH_CLOVER_TWIST_XPAY, a, kappa2);
    flops += (1320ll + 96ll) * in.Volume();
  } else {
    errorQuda("Invalid matpcType");
  }
}

void DiracTwistedCloverPC::MdagM(ColorSpinorField &out, const ColorSpinorField &in) const {
  ColorSpinorField *t = getTmp(tmp2, own2, in);
  M(*t, in);
  Mdag(out, *t);
}

void DiracTwistedCloverPC::prepare(ColorSpinorField *&src, ColorSpinorField *&sol, ColorSpinorField &x, ColorSpinorField &b,
                                   const QudaSolutionType solType) const {
  if (solType == QUDA_MATPC_SOLUTION || solType == QUDA_MATPCDAG_MATPC_SOLUTION) { src = &b; sol = &x; return; }
  checkFullSpinor(x, b);
  ColorSpinorField *t = getTmp(tmp1, own1, b.Even());
  if (matpcType == QUDA_MATPC_EVEN_EVEN) {
    src = &(x.Odd());
    TwistCloverInv(*src, b.Odd(), QUDA_ODD_PARITY);
    DiracWilson::DslashXpay(*t, *src, QUDA_EVEN_PARITY, b.Even(), kappa);
    TwistCloverInv(*src, *t, QUDA_EVEN_PARITY);
    sol = &(x.Even());
  } else if (matpcType == QUDA_MATPC_ODD_ODD) {
    src = &(x.Even());
    TwistCloverInv(*src, b.Even(), QUDA_EVEN_PARITY);
    DiracWilson::DslashXpay(*t, *src, QUDA_ODD_PARITY, b.Odd(), kappa);
    TwistCloverInv(*src, *t, QUDA_ODD_PARITY);
    sol = &(x.Odd());
  } else if (matpcType == QUDA_MATPC_EVEN_EVEN_ASYMMETRIC) {
    src = &(x.Odd());
    TwistCloverInv(*t, b.Odd(), QUDA_ODD_PARITY);
    DiracWilson::DslashXpay(*src, *t, QUDA_EVEN_PARITY, b.Even(), kappa);
    sol = &(x.Even());
  } else if (matpcType == QUDA_MATPC_ODD_ODD_ASYMMETRIC) {
    src = &(x.Even());
    TwistCloverInv(*t, b.Even(), QUDA_EVEN_PARITY);
    DiracWilson::DslashXpay(*src, *t, QUDA_ODD_PARITY, b.Odd(), kappa);
    sol = &(x.Odd());
  } else {
    errorQuda("MatPCType %d not valid for DiracTwistedCloverPC", matpcType);
  }
}

void DiracTwistedCloverPC::reconstruct(ColorSpinorField &x, const ColorSpinorField &b, const QudaSolutionType solType) const {
  if (solType == QUDA_MATPC_SOLUTION || solType == QUDA_MATPCDAG_MATPC_SOLUTION) return;
  checkFullSpinor(x, b);
  ColorSpinorField *t = getTmp(tmp1, own1, b.Even());
  if (matpcType == QUDA_MATPC_EVEN_EVEN || matpcType == QUDA_MATPC_EVEN_EVEN_ASYMMETRIC) {
    DiracWilson::DslashXpay(*t, x.Even(), QUDA_ODD_PARITY, b.Odd(), kappa);
    TwistCloverInv(x.Odd(), *t, QUDA_ODD_PARITY);
  } else if (matpcType == QUDA_MATPC_ODD_ODD || matpcType == QUDA_MATPC_ODD_ODD_ASYMMETRIC) {
    DiracWilson::DslashXpay(*t, x.Odd(), QUDA_EVEN_PARITY, b.Even(), kappa);
    TwistCloverInv(x.Even(), *t, QUDA_EVEN_PARITY);
  } else {
    errorQuda("MatPCType %d not valid for DiracTwistedCloverPC", matpcType);
  }
}

}  // namespace quda
