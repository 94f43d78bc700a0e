/* invert_quda.h — reference header name (include/invert_quda.h:313-331 Solver::create, SolverParam, GCR, MR, BiCGstab): solver.h */
#ifndef QUDA_AMD_FWD_INVERT_QUDA_H
#define QUDA_AMD_FWD_INVERT_QUDA_H
#include <solver.h>
#endif
