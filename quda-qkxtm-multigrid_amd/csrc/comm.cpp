// comm.cpp — process topology and inter-rank transport.
//
// Reference: lib/comm_common.cpp:94-196 (4-D topology, lexicographic rank map with t fastest,
// lib/interface_quda.cpp:261-270), lib/comm_mpi.cpp:297-326 (allreduce).  Transport here is RCCL over
// xGMI (one process per GPU); a single rank degenerates to no-ops.
#include <cstring>

#include "blas.h"
#include "interface_internal.h"

namespace quda {

void commInit(const int *dims, QudaCommsMap func, void *fdata) {
  CommGrid &g = commGrid();
  int n = 1;
  for (int d = 0; d < 4; d++) { g.dims[d] = dims[d]; n *= dims[d]; }
  if (n != g.size) errorQuda("process grid %d x %d x %d x %d needs %d ranks but the communicator has %d", dims[0], dims[1], dims[2], dims[3], n, g.size);
  // default map: rank = ((x*Ny + y)*Nz + z)*Nt + t  (t fastest)
  int r = g.rank;
  for (int d = 3; d >= 0; d--) { g.coords[d] = r % dims[d]; r /= dims[d]; }
  if (func) {
    // user map: find the coordinates that map to this rank
    int c[4];
    bool found = false;
    for (c[0] = 0; c[0] < dims[0] && !found; c[0]++)
      for (c[1] = 0; c[1] < dims[1] && !found; c[1]++)
        for (c[2] = 0; c[2] < dims[2] && !found; c[2]++)
          for (c[3] = 0; c[3] < dims[3] && !found; c[3]++)
            if (func(c, fdata) == g.rank) { memcpy(g.coords, c, sizeof(c)); found = true; }
    if (!found) errorQuda("rank %d not produced by the user comms map", g.rank);
  }
}

void commFinalize() {}

void comm_allreduce(double *, int) {
  if (commGrid().size == 1) return;
  errorQuda("multi-rank reductions: RCCL transport not initialised");
}
void comm_allreduce_max(double *, int) {
  if (commGrid().size == 1) return;
  errorQuda("multi-rank reductions: RCCL transport not initialised");
}

void loadGaugeWithHalo(GaugeField &U, void *const h_gauge[4], QudaPrecision cpu_prec) {
  if (commGrid().size == 1) { U.loadQDP(h_gauge, cpu_prec); return; }
  errorQuda("multi-rank gauge load not built yet");
}

}  // namespace quda
