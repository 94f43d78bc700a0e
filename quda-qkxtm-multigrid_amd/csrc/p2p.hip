// p2p.hip — see p2p.h
#include "p2p.h"

#include <cstdlib>
#include <cstring>

#include "blas.h"
#include "halo.h"
#include "interface_internal.h"

namespace quda {

void *p2pAlloc(size_t bytes) {
  void *p = nullptr;
  HIP_CHECK(hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained));
  HIP_CHECK(hipMemsetAsync(p, 0, bytes, computeStream()));
  HIP_CHECK(hipStreamSynchronize(computeStream()));
  return p;
}
void p2pFree(void *p) { if (p) (void)hipFree(p); }

bool commMapPeers(void *local, PeerMap &m) {
  const CommGrid &g = commGrid();
  for (int s = 0; s < 8; s++) m.peer[s] = nullptr;
  m.opened.clear();
  double fail = 0;
  hipIpcMemHandle_t mine;
  memset(&mine, 0, sizeof(mine));
  if (g.size > 1 && hipIpcGetMemHandle(&mine, local) != hipSuccess) { (void)hipGetLastError(); fail = 1; }
  std::vector<hipIpcMemHandle_t> all(g.size);
  if (g.size > 1) commAllgatherBytes(&mine, all.data(), sizeof(mine));
  comm_allreduce(&fail, 1);
  if (fail > 0) return false;
  std::vector<void *> byRank(g.size, nullptr);
  byRank[g.rank] = local;
  for (int s = 0; s < 8; s++) {
    const int r = commNeighborRank(s >> 1, (s & 1) ? +1 : -1);
    if (!byRank[r]) {
      void *p = nullptr;
      if (hipIpcOpenMemHandle(&p, all[r], hipIpcMemLazyEnablePeerAccess) != hipSuccess) { (void)hipGetLastError(); fail = 1; break; }
      byRank[r] = p;
      m.opened.push_back(p);
    }
    m.peer[s] = byRank[r];
  }
  comm_allreduce(&fail, 1);
  if (fail > 0) { commUnmapPeers(m); return false; }
  return true;
}
bool commMapAllRanks(void *local, std::vector<void *> &byRank, std::vector<void *> &opened) {
  const CommGrid &g = commGrid();
  byRank.assign(g.size, nullptr);
  opened.clear();
  byRank[g.rank] = local;
  if (g.size == 1) return true;
  double fail = 0;
  hipIpcMemHandle_t mine;
  memset(&mine, 0, sizeof(mine));
  if (hipIpcGetMemHandle(&mine, local) != hipSuccess) { (void)hipGetLastError(); fail = 1; }
  std::vector<hipIpcMemHandle_t> all(g.size);
  commAllgatherBytes(&mine, all.data(), sizeof(mine));
  comm_allreduce(&fail, 1);
  if (fail > 0) return false;
  for (int r = 0; r < g.size && fail == 0; r++) {
    if (r == g.rank) continue;
    void *p = nullptr;
    if (hipIpcOpenMemHandle(&p, all[r], hipIpcMemLazyEnablePeerAccess) != hipSuccess) { (void)hipGetLastError(); fail = 1; break; }
    byRank[r] = p;
    opened.push_back(p);
  }
  comm_allreduce(&fail, 1);
  if (fail > 0) {
    for (void *p : opened) (void)hipIpcCloseMemHandle(p);
    opened.clear();
    return false;
  }
  return true;
}

void commUnmapPeers(PeerMap &m) {
  for (void *p : m.opened) (void)hipIpcCloseMemHandle(p);
  m.opened.clear();
  for (int s = 0; s < 8; s++) m.peer[s] = nullptr;
}

unsigned long long p2pTimeoutTicks() {
  static unsigned long long t = 0;
  if (!t) { const char *e = getenv("QUDA_AMD_P2P_TIMEOUT_S"); t = (unsigned long long)((e ? atof(e) : 10.0) * 1e8); if (!t) t = 1; }
  return t;
}

static int *g_err = nullptr;   // device record (p2p.h): word 0 = 0, or the code of the first wait that timed out (later waits then return at once)
int *p2pErrorWord() {
  if (!g_err) {
    HIP_CHECK(qaMalloc((void **)&g_err, kP2pErrInts * sizeof(int)));
    HIP_CHECK(hipMemset(g_err, 0, kP2pErrInts * sizeof(int)));
    HIP_CHECK(hipDeviceSynchronize());
  }
  return g_err;
}
// What the record says about the three ways an exchange can fail: the expected flag of a (dimension, buffer) zone is this rank's use
// count of the zone, the sender writes ITS use count of the same zone; a word that still carries the flag of the zone's previous use
// (expected - 1; 0 on the first use) was simply never written in this exchange — the sender is late or starved; any other value is
// a sender whose counters disagree with this rank's (different number of exchanges on the two sides, or the other buffer).
static bool describe(const int *h, char *text, size_t n) {
  if (!h[0]) return false;
  const CommGrid &g = commGrid();
  const bool coarse = h[0] >= 17;
  const int hop = coarse ? h[0] - 17 : h[0] - 1, mu = hop >> 1, bwd = hop & 1;
  const unsigned expect = (unsigned)h[2], s0 = (unsigned)h[3], s1 = (unsigned)h[4];
  const int from = commNeighborRank(mu, bwd ? -1 : +1);
  const char *verdict;
  if (coarse) verdict = (int)(s0 - expect) < 0 ? "the arrival counter is short: the neighbour's pack kernel has not (fully) delivered this exchange" : "counter reached after the bound";
  else if ((s0 == expect - 1 || s0 == 0) && (s1 == expect - 1 || s1 == 0)) verdict = "the words still carry the zone's previous use: this exchange's face was never written (sender late, starved or gone)";
  else if (s0 == expect || s1 == expect) verdict = "one half of the vector arrived, the other did not: a store of this exchange is still in flight or was lost";
  else verdict = "the words carry a flag of neither this exchange nor the previous one: the two ranks disagree about the exchange count of this zone (or about the buffer)";
  snprintf(text, n, "%s-grid halo wait ran out on rank %d: %s hop in dimension %d (face from rank %d), %s %d, vector %d, exchange %u of this rank's window, buffer %d: expected %s %u, last seen %u / %u — %s",
           coarse ? "coarse" : "fine", g.rank, bwd ? "backward" : "forward", mu, from, coarse ? "coarse site" : "face site", h[1], h[7], (unsigned)h[5], h[6], coarse ? "arrival count" : "flag",
           expect, s0, s1, verdict);
  return true;
}
bool p2pDescribeError(char *text, size_t n) {
  if (!g_err) return false;
  int h[kP2pErrInts];
  HIP_CHECK(hipMemcpy(h, g_err, sizeof(h), hipMemcpyDeviceToHost));
  return describe(h, text, n);
}
void p2pCheck(const char *where) {
  char text[768];
  if (p2pDescribeError(text, sizeof(text))) errorQuda("%s: %s (QUDA_AMD_P2P_TIMEOUT_S bounds the wait; QUDA_AMD_HALO=rccl selects the staged transport)", where, text);
}

// ---- token round trips through the mapped windows, with the same access types as the production protocols: system-scope
// write-through stores for the payload + a fire-and-forget remote atomic add for the counter + system-scope polling and loads
// (coarse-grid halo, all-reduce), and 16-byte flag-in-data buffer stores / polling buffer loads (fine-grid halo).
// Several rounds over the SAME addresses, so a receiver that could serve a later round from a stale cache line fails here
// and not in production. ----
struct ProbeWindow { unsigned data[8][16]; unsigned flag[8]; unsigned pad[8]; unsigned ll[8][32][4]; unsigned sec[8][32][8]; };   // ll, sec: 16-byte aligned
typedef unsigned int probe_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned probe_token(int rank, int s, int k, int round) { return 0x5eed0000u + (unsigned)round * 4096u + (unsigned)rank * 128u + s * 16u + k; }

__global__ void p2p_probe_send(ProbeWindow *const *peer, int rank, int round) {
  const int s = threadIdx.x >> 4, k = threadIdx.x & 15;   // 8 slots x 16 words
  if (s >= 8 || !peer[s]) return;
  ProbeWindow *w = peer[s];
  __hip_atomic_store(&w->data[s][k], probe_token(rank, s, k, round), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // stores acknowledged before the barrier and the counter (as the production kernels)
  __syncthreads();
  if (k == 0) (void)__hip_atomic_fetch_add(&w->flag[s], 16u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  // the fine-grid halo's flag-in-data vectors {word, flag, word, flag}: 16-byte buffer stores, sc0 sc1, no ordering, no signal
  for (int v = k; v < 32; v += 16) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)w->ll[s], 0, (int)sizeof(w->ll[s]), 0x00020000);
    probe_u32x4 q; q.x = probe_token(rank, s, v, round); q.y = 0xf1a60000u + round; q.z = ~probe_token(rank, s, v, round); q.w = 0xf1a60000u + round;
    __builtin_amdgcn_raw_buffer_store_b128(q, rs, v * 16, 0, 17);
  }
  // the compact format (dslash.hip ghost_atom_store): self-validating 16-byte atoms {3 payload words, flag}, each one 16-byte store of one lane
  for (int v4 = 0; v4 < 4; v4++) {
    const int at = v4 * 16 + k;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)w->sec[s], 0, (int)sizeof(w->sec[s]), 0x00020000);
    const unsigned tk = probe_token(rank, s, at, round);
    probe_u32x4 q; q.x = tk; q.y = ~tk; q.z = tk + 1u; q.w = 0x5ec70000u + round;
    __builtin_amdgcn_raw_buffer_store_b128(q, rs, at * 16, 0, 17);
  }
}
__global__ void p2p_probe_recv(ProbeWindow *mine, const int *fromRank, unsigned long long ticks, int *result, int round) {
  const int s = threadIdx.x;
  if (s >= 8) return;
  const unsigned long long t0 = wall_clock64();
  bool ok = true;
  while ((int)(__hip_atomic_load(&mine->flag[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - 16u * (unsigned)round) < 0) {
    if (wall_clock64() - t0 > ticks) { ok = false; break; }
    __builtin_amdgcn_s_sleep(1);
  }
  if (ok)
    for (int k = 0; k < 16; k++) ok = ok && __hip_atomic_load(&mine->data[s][k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == probe_token(fromRank[s], s, k, round);
  // flag-in-data vectors: poll each one until both halves carry this round's flag, then check the words
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)mine->ll[s], 0, (int)sizeof(mine->ll[s]), 0x00020000);
  for (int v = 0; v < 32 && ok; v++) {
    const unsigned flag = 0xf1a60000u + round;
    for (;;) {
      const probe_u32x4 q = __builtin_amdgcn_raw_buffer_load_b128(rs, v * 16, 0, 17);
      if (q.y == flag && q.w == flag) { ok = q.x == probe_token(fromRank[s], s, v, round) && q.z == ~probe_token(fromRank[s], s, v, round); break; }
      if (wall_clock64() - t0 > ticks) { ok = false; break; }
      __builtin_amdgcn_s_sleep(1);
    }
  }
  const __amdgpu_buffer_rsrc_t rsec = __builtin_amdgcn_make_buffer_rsrc((void *)mine->sec[s], 0, (int)sizeof(mine->sec[s]), 0x00020000);
  for (int v = 0; v < 64 && ok; v++) {
    const unsigned flag = 0x5ec70000u + round, tk = probe_token(fromRank[s], s, v, round);
    for (;;) {
      const probe_u32x4 q = __builtin_amdgcn_raw_buffer_load_b128(rsec, v * 16, 0, 17);
      if (q.w == flag) { ok = q.x == tk && q.y == ~tk && q.z == tk + 1u; break; }
      if (wall_clock64() - t0 > ticks) { ok = false; break; }
      __builtin_amdgcn_s_sleep(1);
    }
  }
  if (!ok) atomicAdd(result, 1);
}

// Can this rank's device reach the devices of all other ranks directly?  PCI bus ids are exchanged (device ordinals mean nothing
// across processes with different visibility masks); a peer that is visible here must be peer-accessible, and access is switched on
// up front rather than left to the lazy path of hipIpcOpenMemHandle — a kernel touching a window that is not reachable would
// fault, and a fault cannot be caught and turned into a fall-back.  Ranks on the same device (rehearsals) need nothing.
static bool g_deviceShared = false;
bool p2pDeviceShared() { return g_deviceShared; }
static bool peersReachable() {
  const CommGrid &g = commGrid();
  g_deviceShared = false;
  if (g.size == 1) return true;
  int dev = 0;
  HIP_CHECK(hipGetDevice(&dev));
  char mine[64];
  memset(mine, 0, sizeof(mine));
  if (hipDeviceGetPCIBusId(mine, (int)sizeof(mine) - 1, dev) != hipSuccess) { (void)hipGetLastError(); mine[0] = 0; }
  std::vector<char> all((size_t)g.size * 64);
  commAllgatherBytes(mine, all.data(), 64);
  double fail = 0;
  for (int r = 0; r < g.size && fail == 0; r++) {
    if (r == g.rank) continue;
    const char *bus = &all[(size_t)r * 64];
    int peer = -1;
    if (!bus[0] || hipDeviceGetByPCIBusId(&peer, bus) != hipSuccess) { (void)hipGetLastError(); continue; }   // not visible here: leave it to the IPC mapping
    if (peer == dev) { g_deviceShared = true; continue; }   // another rank on this very device (rehearsals): see p2pDeviceShared
    int can = 0;
    if (hipDeviceCanAccessPeer(&can, dev, peer) != hipSuccess) { (void)hipGetLastError(); can = 0; }
    if (!can) { fail = 1; break; }
    const hipError_t e = hipDeviceEnablePeerAccess(peer, 0);
    if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) fail = 1;
    (void)hipGetLastError();
  }
  comm_allreduce(&fail, 1);
  { double sh = g_deviceShared ? 1.0 : 0.0; comm_allreduce_max(&sh, 1); g_deviceShared = sh != 0.0; }   // the same answer on every rank
  return fail == 0;
}

static int g_p2p = -1;
void p2pReset() {
  g_p2p = -1;
  if (g_err) { (void)hipFree(g_err); g_err = nullptr; }
}
static long long g_stats[8];
long long *p2pStats() { return g_stats; }
int p2pTransport() { return g_p2p; }
// the production kernel disagreed with the staged transport on its first use (halo.h, verifyPeerStores): staged from now on
void p2pDisable() { g_p2p = 0; g_stats[6]++; }
int p2pTakeError() {   // read and clear the device error record
  if (!g_err) return 0;
  int h[kP2pErrInts];
  HIP_CHECK(hipMemcpy(h, g_err, sizeof(h), hipMemcpyDeviceToHost));
  if (h[0]) {
    char text[768];
    if (describe(h, text, sizeof(text)) && getVerbosity() >= QUDA_SUMMARIZE) warningQuda("%s", text);
    HIP_CHECK(hipMemset(g_err, 0, sizeof(h))); HIP_CHECK(hipDeviceSynchronize());
  }
  return h[0];
}

bool p2pHaloEnabled() {
  if (g_p2p >= 0) return g_p2p != 0;
  const char *e = getenv("QUDA_AMD_HALO");
  if (e && !strcmp(e, "rccl")) { g_p2p = 0; return false; }
  const CommGrid &g = commGrid();
  ProbeWindow *win = (ProbeWindow *)p2pAlloc(sizeof(ProbeWindow));
  PeerMap pm;
  double fail = 0;
  if (!peersReachable() || !commMapPeers(win, pm)) {
    fail = 1;
  } else {
    ProbeWindow **d_peer = nullptr;
    int *d_from = nullptr, *d_res = nullptr, h_from[8], h_res = 0;
    // window s is written by the rank that has me as its slot-s neighbour: its +dim neighbour (s odd) is me -> it is my -dim neighbour
    for (int s = 0; s < 8; s++) h_from[s] = commNeighborRank(s >> 1, (s & 1) ? -1 : +1);
    HIP_CHECK(qaMalloc((void **)&d_peer, 8 * sizeof(void *)));
    HIP_CHECK(qaMalloc((void **)&d_from, 8 * sizeof(int)));
    HIP_CHECK(qaMalloc((void **)&d_res, sizeof(int)));
    HIP_CHECK(hipMemcpy(d_peer, pm.peer, 8 * sizeof(void *), hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(d_from, h_from, 8 * sizeof(int), hipMemcpyHostToDevice));
    HIP_CHECK(hipMemset(d_res, 0, sizeof(int)));
    HIP_CHECK(hipDeviceSynchronize());
    for (int round = 1; round <= 3 && fail == 0; round++) {
      hipLaunchKernelGGL(p2p_probe_send, dim3(1), dim3(128), 0, computeStream(), (ProbeWindow *const *)d_peer, g.rank, round);
      hipLaunchKernelGGL(p2p_probe_recv, dim3(1), dim3(64), 0, computeStream(), win, d_from, (unsigned long long)3e8, d_res, round);
      if (hipStreamSynchronize(computeStream()) != hipSuccess) { (void)hipGetLastError(); h_res = 1; }
      else HIP_CHECK(hipMemcpy(&h_res, d_res, sizeof(int), hipMemcpyDeviceToHost));
      fail = h_res ? 1 : 0;
      comm_allreduce(&fail, 1);   // every rank has verified this round before anyone overwrites the tokens
    }
    (void)hipFree(d_peer); (void)hipFree(d_from); (void)hipFree(d_res);
  }
  comm_allreduce(&fail, 1);   // also keeps every rank's window alive until all neighbours have written it
  commUnmapPeers(pm);
  commBarrier();
  p2pFree(win);
  g_p2p = fail > 0 ? 0 : 1;
  if (g.rank == 0 && getVerbosity() >= QUDA_SUMMARIZE)
    printfQuda("halo transport: %s\n", g_p2p ? "direct peer stores (IPC-mapped ghost zones)" : "RCCL send/recv (peer mapping unavailable)");
  if (!g_p2p && e && !strcmp(e, "p2p")) errorQuda("QUDA_AMD_HALO=p2p requested but the peer windows cannot be mapped / verified");
  return g_p2p != 0;
}

}  // namespace quda
