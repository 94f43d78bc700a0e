// interface_internal.h — declarations shared between the C-ABI translation units.
#pragma once

#include "dirac.h"

namespace quda {

// comm.cpp — process grid + transport (RCCL over xGMI; single-rank degenerates to no-ops)
void commInit(const int *dims, QudaCommsMap func, void *fdata);
void commFinalize();
// upload host links into the bidirectional device layout, fetching the backward links that live on the
// -mu neighbour rank (replaces cudaGaugeField::exchangeGhost, reference lib/cuda_gauge_field.cu:160-188)
void loadGaugeWithHalo(GaugeField &U, void *const h_gauge[4], QudaPrecision cpu_prec);

// interface.cpp
extern GaugeField *gaugePrecise, *gaugeSloppy, *gaugePrecondition, *gaugeSmeared;
extern CloverField *cloverPrecise, *cloverSloppy, *cloverPrecondition;
const LatticeGeom &residentGeom();
GaugeField *residentGauge(int which);
CloverField *residentClover(int which);
void setDiracParam(DiracParam &dp, QudaInvertParam *inv, const bool pc);
void setDiracSloppyParam(DiracParam &dp, QudaInvertParam *inv, const bool pc);
void setDiracPreParam(DiracParam &dp, QudaInvertParam *inv, const bool pc);
ColorSpinorParam deviceSpinorParam(QudaPrecision prec, QudaSiteSubset subset, QudaTwistFlavorType flavor);

}  // namespace quda
