// coarse_cycle.h — the multigrid cycle BELOW a given coarse level as ONE persistent kernel.
//
// On the coarse levels of a grid-decomposed lattice (8 x 4 x 4 x 4 and 4 x 2 x 2 x 2 sites per rank in the 8-GPU split of 32^4) the
// V-cycle of the reference (lib/multigrid.cpp:488-604: pre-smooth -> residual -> restrict -> coarse solve -> prolong -> post-smooth,
// with MR smoothers lib/inv_mr_quda.cpp:40-200 and the coarsest-grid GCR lib/inv_gcr_quda.cpp:235-516 on the even-odd preconditioned
// coarse operator lib/dirac_coarse.cpp:237-372, lib/dslash_coarse.cu:68-137) is a chain of ~130 launches of 4-15 us each per cycle:
// pure launch / dependency latency (profiles/r03k_sub8_masked_mg_solve.log).  Here the whole chain — Schur prepare, MR iterations,
// reconstruct, residual, restrictor, coarsest GCR with its orthogonalisation and restarts, prolongator, post-smoothing — runs inside one
// launch: one work-group per coarse site (grid-stride loop), phases separated by a device-wide barrier, scalars (alpha of MR; beta,
// gamma, alpha of GCR; convergence decisions) computed redundantly and identically by every work-group from partial sums reduced in a
// fixed order, halos of partitioned dimensions pushed straight into the neighbours' peer-mapped ghost windows as flag-in-data words and
// polled by the sites that hop across the face, global sums exchanged the same way.  Same arithmetic as the unfused path (fp32 fields,
// fp64 sums), same iteration counts.
#pragma once

#include "multigrid.h"

namespace quda {

class CoarseCycle;

// nullptr: this sub-hierarchy does not qualify (K-cycle, full-operator smoother, aggregates of more than 64 sites, a level too large
// to profit, ranks sharing one device, ...) — the caller keeps the kernel-per-operation path.  `top` is the MG object of the first
// fused level (level >= 1).
CoarseCycle *coarseCycleCreate(MG &top);
void coarseCycleDestroy(CoarseCycle *c);
// x = cycle(b) on the top level of the fused sub-hierarchy; false: not run (caller falls back)
bool coarseCycleApply(CoarseCycle *c, ColorSpinorField &x, ColorSpinorField &b);
// statistics of the last launch: [0] device-wide barriers, [1] coarsest-grid GCR iterations, [2] restarts, [3] halo exchanges, [4] grid size
void coarseCycleStats(const CoarseCycle *c, long long out[5]);
// 0 off, 1 on (default), read once from QUDA_AMD_MG_FUSED
int coarseCycleEnabled();
void coarseCycleSetEnabled(int on);

}  // namespace quda
