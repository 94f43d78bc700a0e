// comm.cpp — process topology and inter-rank transport: RCCL over xGMI, one process per GPU.
//
// What the reference does with MPI persistent point-to-point messages / CUDA-IPC copies and MPI_Allreduce
// (lib/comm_common.cpp:94-196 topology, lib/comm_mpi.cpp:198-326, lib/cuda_color_spinor_field.cu:1212-1765) is
// done here with ONE grouped ncclSend/ncclRecv per Dslash (every face of every partitioned dimension in a
// single group, so on a 2x2x2 grid the <= 6 faces travel over 6 different xGMI links concurrently) and
// ncclAllReduce for the 8-24-byte global sums.  The communicator is bootstrapped by the launcher: rank 0 calls
// qudaAmdCommGetUniqueId, the id is broadcast out of band (torch.distributed / MPI / files), every rank calls
// qudaAmdCommInit, then initCommsGridQuda fixes the 4-D grid (rank = ((x*Ny + y)*Nz + z)*Nt + t, t fastest,
// reference lib/interface_quda.cpp:261-270).  A single rank needs none of this and degenerates to local copies.
#include <rccl/rccl.h>

#include <unistd.h>

#include <cstring>
#include <string>
#include <vector>

#include "blas.h"
#include "halo.h"
#include "interface_internal.h"
#include "p2p.h"
#include "quda_amd_ext.h"

namespace quda {

static ncclComm_t g_nccl = nullptr;

// ---- rehearsal transport (QUDA_AMD_TRANSPORT=shm): messages as files in a shared directory, staged through the host.
// Exists only so that N ranks can be rehearsed on ONE GPU (RCCL refuses duplicate devices); it follows the same posting
// order / per-peer FIFO matching as the RCCL path, so neighbour maps, face ordering and the two-ranks-per-dimension case
// are exercised end to end.  Never used unless the environment variable is set. ----
static bool g_shm = false;
// QUDA_AMD_RCCL_SELFTEST=1 with a one-rank communicator: self-neighbour messages and reductions go through the real
// ncclSend/ncclRecv/ncclAllReduce calls (peer = own rank) instead of the local shortcuts, so the RCCL call sequence of
// the multi-GPU path can be validated on a one-GPU box
static bool g_rccl_self = false;
static std::string g_shm_dir;
static std::vector<unsigned long> g_seq_send, g_seq_recv;
static unsigned long g_seq_red = 0;
// incarnation of the communicator inside this process: sequence numbers restart with every qudaAmdCommInit while the directory
// (and the never-removed files of a communicator's last round) stays, so file names carry the incarnation — a rank that is
// already in the next incarnation must not pick up a file the previous one left behind (seen as rare 2-rank rehearsal time-outs:
// the 2-rank rehearsal re-initialises three times per run)
static int g_shm_epoch = 0;

static void shmWrite(const std::string &name, const void *p, size_t n) {
  const std::string ep = "e" + std::to_string(g_shm_epoch) + "_";
  const std::string tmp = g_shm_dir + "/." + ep + name + ".tmp", fin = g_shm_dir + "/" + ep + name;
  FILE *f = fopen(tmp.c_str(), "wb");
  if (!f || fwrite(p, 1, n, f) != n) errorQuda("shm transport: cannot write %s", tmp.c_str());
  fclose(f);
  if (rename(tmp.c_str(), fin.c_str())) errorQuda("shm transport: rename failed for %s", fin.c_str());
}
static void shmRead(const std::string &name, void *p, size_t n, bool consume) {
  const std::string fin = g_shm_dir + "/e" + std::to_string(g_shm_epoch) + "_" + name;
  for (long spins = 0;; spins++) {
    FILE *f = fopen(fin.c_str(), "rb");
    if (f) {
      const size_t got = fread(p, 1, n, f);
      fclose(f);
      if (got != n) errorQuda("shm transport: short read on %s (%zu of %zu)", fin.c_str(), got, n);
      if (consume) remove(fin.c_str());
      return;
    }
    if (spins > 600000) errorQuda("shm transport: timed out waiting for %s", fin.c_str());
    usleep(100);
  }
}

#define NCCL_CHECK(cmd)                                                                   \
  do {                                                                                    \
    ncclResult_t r_ = (cmd);                                                              \
    if (r_ != ncclSuccess) errorQuda("RCCL call '%s' failed: %s", #cmd, ncclGetErrorString(r_)); \
  } while (0)

void commInit(const int *dims, QudaCommsMap func, void *fdata) {
  CommGrid &g = commGrid();
  int n = 1;
  for (int d = 0; d < 4; d++) { g.dims[d] = dims[d]; n *= dims[d]; }
  if (n != g.size) errorQuda("process grid %d x %d x %d x %d needs %d ranks but the communicator has %d (call qudaAmdCommInit first)", dims[0], dims[1], dims[2], dims[3], n, g.size);
  int r = g.rank;
  for (int d = 3; d >= 0; d--) { g.coords[d] = r % dims[d]; r /= dims[d]; }
  g.user_map = func;
  g.user_data = fdata;
  if (func) {
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

int commRankFromCoords(const int *c) {
  const CommGrid &g = commGrid();
  if (g.user_map) return g.user_map(c, g.user_data);
  return ((c[0] * g.dims[1] + c[1]) * g.dims[2] + c[2]) * g.dims[3] + c[3];
}

int commNeighborRank(int dim, int dir) {
  const CommGrid &g = commGrid();
  int c[4];
  for (int d = 0; d < 4; d++) c[d] = g.coords[d];
  c[dim] = (c[dim] + dir + g.dims[dim]) % g.dims[dim];
  return commRankFromCoords(c);
}

void commFinalize() {
  freeHaloBuffers();
  if (g_nccl) { (void)ncclCommDestroy(g_nccl); g_nccl = nullptr; }
  g_shm = false;
  g_rccl_self = false;
  p2pReset();
  CommGrid &g = commGrid();
  g.rank = 0; g.size = 1;
  for (int d = 0; d < 4; d++) { g.dims[d] = 1; g.coords[d] = 0; g.forced[d] = false; }
}

// ---- reductions ----
static double *d_scratch = nullptr;
static double *h_scratch = nullptr;
static void ensureScratch() {
  if (!d_scratch) HIP_CHECK(qaMalloc((void **)&d_scratch, 64 * sizeof(double)));
  if (!h_scratch) HIP_CHECK(hipHostMalloc((void **)&h_scratch, 64 * sizeof(double), hipHostMallocDefault));
}

static void shmAllreduce(double *data, int n, bool is_max) {
  const CommGrid &g = commGrid();
  const unsigned long seq = g_seq_red++;
  shmWrite("red_" + std::to_string(seq) + "_" + std::to_string(g.rank), data, n * sizeof(double));
  std::vector<double> acc(data, data + n), tmp(n);
  for (int r = 0; r < g.size; r++) {
    if (r == g.rank) continue;
    shmRead("red_" + std::to_string(seq) + "_" + std::to_string(r), tmp.data(), n * sizeof(double), false);
    for (int k = 0; k < n; k++) acc[k] = is_max ? (tmp[k] > acc[k] ? tmp[k] : acc[k]) : acc[k] + tmp[k];
  }
  // everyone has read round seq-1 once it writes round seq: clean up our own file of the previous round
  if (seq > 0) remove((g_shm_dir + "/e" + std::to_string(g_shm_epoch) + "_red_" + std::to_string(seq - 1) + "_" + std::to_string(g.rank)).c_str());
  for (int k = 0; k < n; k++) data[k] = acc[k];
}

bool commReductionsNeeded() { return commGrid().size > 1 || g_rccl_self; }

void commAllreduceDevice(double *d_data, int n, hipStream_t s) {
  if (commGrid().size == 1 && !g_rccl_self) return;
  if (g_shm) {
    double h[64];
    HIP_CHECK(hipMemcpyAsync(h, d_data, n * sizeof(double), hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    shmAllreduce(h, n, false);
    HIP_CHECK(hipMemcpyAsync(d_data, h, n * sizeof(double), hipMemcpyHostToDevice, s));
    HIP_CHECK(hipStreamSynchronize(s));
    return;
  }
  NCCL_CHECK(ncclAllReduce(d_data, d_data, n, ncclDouble, ncclSum, g_nccl, s));
}

static void hostAllreduce(double *data, int n, ncclRedOp_t op) {
  if (commGrid().size == 1 && !g_rccl_self) return;
  if (n > 64) errorQuda("allreduce of %d doubles exceeds the scratch buffer", n);
  if (g_shm) { shmAllreduce(data, n, op == ncclMax); return; }
  ensureScratch();
  hipStream_t s = computeStream();
  memcpy(h_scratch, data, n * sizeof(double));
  HIP_CHECK(hipMemcpyAsync(d_scratch, h_scratch, n * sizeof(double), hipMemcpyHostToDevice, s));
  NCCL_CHECK(ncclAllReduce(d_scratch, d_scratch, n, ncclDouble, op, g_nccl, s));
  HIP_CHECK(hipMemcpyAsync(h_scratch, d_scratch, n * sizeof(double), hipMemcpyDeviceToHost, s));
  HIP_CHECK(hipStreamSynchronize(s));
  memcpy(data, h_scratch, n * sizeof(double));
}
void comm_allreduce(double *data, int n) { hostAllreduce(data, n, ncclSum); }
void comm_allreduce_max(double *data, int n) { hostAllreduce(data, n, ncclMax); }

// every rank contributes n bytes; all[r*n ..] = rank r's blob (IPC handles of the peer-mapped halo windows)
void commAllgatherBytes(const void *mine, void *all, size_t n) {
  const CommGrid &g = commGrid();
  if (g.size == 1) { memcpy(all, mine, n); return; }
  if (g_shm) {
    static unsigned long seq = 0;
    const unsigned long q = seq++;
    shmWrite("ag_" + std::to_string(q) + "_" + std::to_string(g.rank), mine, n);
    for (int r = 0; r < g.size; r++) shmRead("ag_" + std::to_string(q) + "_" + std::to_string(r), (char *)all + (size_t)r * n, n, false);
    return;
  }
  char *d = nullptr;
  HIP_CHECK(qaMalloc((void **)&d, n * (g.size + 1)));
  hipStream_t s = computeStream();
  HIP_CHECK(hipMemcpyAsync(d, mine, n, hipMemcpyHostToDevice, s));
  NCCL_CHECK(ncclAllGather(d, d + n, n, ncclChar, g_nccl, s));
  HIP_CHECK(hipMemcpyAsync(all, d + n, n * g.size, hipMemcpyDeviceToHost, s));
  HIP_CHECK(hipStreamSynchronize(s));
  HIP_CHECK(hipFree(d));
}

// ---- neighbour exchange: all (dim, dir) messages of one halo in a single RCCL group ----
void commExchange(const std::vector<HaloMsg> &msgs, hipStream_t s) {
  const CommGrid &g = commGrid();
  bool remote = false;
  for (const HaloMsg &m : msgs) {
    const int to = commNeighborRank(m.dim, m.dir), from = commNeighborRank(m.dim, -m.dir);
    if (to == g.rank && from == g.rank && !g_rccl_self) {
      // self neighbour (unpartitioned-but-forced dimension): the message lands in this rank's own ghost zone
      HIP_CHECK(hipMemcpyAsync(m.recv, m.send, m.bytes, hipMemcpyDeviceToDevice, s));
    } else {
      remote = true;
    }
  }
  if (!remote) return;
  if (g_shm) {
    // same posting order as the RCCL group below; per-peer FIFO matching through per-pair sequence numbers
    HIP_CHECK(hipStreamSynchronize(s));
    std::vector<char> host;
    for (const HaloMsg &m : msgs) {
      const int to = commNeighborRank(m.dim, m.dir);
      if (to == g.rank) continue;
      host.resize(m.bytes);
      HIP_CHECK(hipMemcpy(host.data(), m.send, m.bytes, hipMemcpyDeviceToHost));
      shmWrite("msg_" + std::to_string(g.rank) + "_" + std::to_string(to) + "_" + std::to_string(g_seq_send[to]++), host.data(), m.bytes);
    }
    for (const HaloMsg &m : msgs) {
      const int from = commNeighborRank(m.dim, -m.dir);
      if (from == g.rank) continue;
      host.resize(m.bytes);
      shmRead("msg_" + std::to_string(from) + "_" + std::to_string(g.rank) + "_" + std::to_string(g_seq_recv[from]++), host.data(), m.bytes, true);
      HIP_CHECK(hipMemcpy(m.recv, host.data(), m.bytes, hipMemcpyHostToDevice));
    }
    return;
  }
  if (!g_nccl) errorQuda("multi-rank halo exchange without an RCCL communicator (qudaAmdCommInit)");
  NCCL_CHECK(ncclGroupStart());
  // order inside a (dim) pair: send forward, send backward, receive from behind, receive from ahead — with only two ranks
  // along a dimension both neighbours are the same peer and RCCL matches messages to one peer in posting order
  for (const HaloMsg &m : msgs) {
    const int to = commNeighborRank(m.dim, m.dir);
    if (to == g.rank && !g_rccl_self) continue;
    NCCL_CHECK(ncclSend(m.send, m.bytes, ncclChar, to, g_nccl, s));
  }
  for (const HaloMsg &m : msgs) {
    const int from = commNeighborRank(m.dim, -m.dir);
    if (from == g.rank && !g_rccl_self) continue;
    NCCL_CHECK(ncclRecv(m.recv, m.bytes, ncclChar, from, g_nccl, s));
  }
  NCCL_CHECK(ncclGroupEnd());
}

void loadGaugeWithHalo(GaugeField &U, void *const h_gauge[4], QudaPrecision cpu_prec) { U.loadQDP(h_gauge, cpu_prec); }

void commBarrier() {
  if (commGrid().size == 1) return;
  double one = 1.0;
  comm_allreduce(&one, 1);
}

}  // namespace quda

using namespace quda;

extern "C" {

void qudaAmdCommGetUniqueId(void *out128) {
  ncclUniqueId id;
  NCCL_CHECK(ncclGetUniqueId(&id));
  static_assert(sizeof(id) == 128, "unique id size");
  memcpy(out128, &id, sizeof(id));
}

void qudaAmdCommInit(const void *id128, int rank, int size) {
  CommGrid &g = commGrid();
  if (size < 1 || rank < 0 || rank >= size) errorQuda("bad rank/size %d/%d", rank, size);
  g.rank = rank;
  g.size = size;
  const char *st = getenv("QUDA_AMD_RCCL_SELFTEST");
  g_rccl_self = size == 1 && st && atoi(st) != 0;
  if (size == 1 && !g_rccl_self) return;
  const char *tr = getenv("QUDA_AMD_TRANSPORT");
  if (tr && !strcmp(tr, "shm")) {
    const char *dir = getenv("QUDA_AMD_SHM_DIR");
    if (!dir) errorQuda("QUDA_AMD_TRANSPORT=shm needs QUDA_AMD_SHM_DIR");
    g_shm = true;
    g_shm_epoch++;
    g_shm_dir = dir;
    g_seq_send.assign(size, 0);
    g_seq_recv.assign(size, 0);
    g_seq_red = 0;
    return;
  }
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  NCCL_CHECK(ncclCommInitRank(&g_nccl, size, id, rank));
}

int qudaAmdCommRank(void) { return commGrid().rank; }
int qudaAmdCommSize(void) { return commGrid().size; }
void qudaAmdCommCoords(int coords[4]) { for (int d = 0; d < 4; d++) coords[d] = commGrid().coords[d]; }
void qudaAmdCommBarrier(void) { commBarrier(); }
void qudaAmdCommAllreduce(double *data, int n) { comm_allreduce(data, n); }
void qudaAmdCommAllreduceMax(double *data, int n) { comm_allreduce_max(data, n); }

}
