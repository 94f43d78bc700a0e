// coarse.hip — coarse-grid operator: apply kernel and Galerkin construction (see coarse.h).
//
// Apply: one 256-thread work-group per coarse site; the 9 matrices of the site are split over the 4 waves, lane = output
// row, each lane streams its row in 16-byte column pairs (unit stride across lanes) against the neighbour's vector
// broadcast from LDS; partial rows are combined through LDS.  Per site 9 (2Nc)^2 complex = 166 KB for Nc = 24 are read
// exactly once: HBM-bound at ~1 flop/byte for a single right-hand side, so plain FMAs, not MFMA.
#include "coarse.h"
#include "block.h"
#include "halo.h"

#include "blas.h"

#include <algorithm>
#include <cstring>

namespace quda {

CoarseGauge::CoarseGauge(const int xc[4], int n_) : n(n_), data(nullptr), data_h(nullptr) {
  nSites = 1;
  for (int d = 0; d < 4; d++) { Xc[d] = xc[d]; nSites *= xc[d]; }
  bytes = (size_t)nSites * 9 * n * n * 2 * sizeof(float);
  data = (float *)poolDeviceMalloc(bytes);   // 7 GB at 12^3 x 24, n = 48: reused by the next hierarchy like V (transfer.hip)
  HIP_CHECK(hipMemsetAsync(data, 0, bytes, computeStream()));
}
CoarseGauge::~CoarseGauge() { if (data) poolDeviceFree(data, 0); if (data_h) poolDeviceFree(data_h, 0); }

typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
__global__ void to_half_kernel(half4_t *out, const float4 *in, size_t n) {
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 v = in[i];
  half4_t h;
  h.x = (_Float16)v.x; h.y = (_Float16)v.y; h.z = (_Float16)v.z; h.w = (_Float16)v.w;
  out[i] = h;
}
void CoarseGauge::makeHalf() const {
  if (data_h) return;
  const size_t n4 = bytes / sizeof(float4);
  data_h = poolDeviceMalloc(n4 * sizeof(half4_t));
  hipLaunchKernelGGL(to_half_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, computeStream(), (half4_t *)data_h, (const float4 *)data, n4);
  HIP_CHECK(hipGetLastError());
}

static bool g_mgHalf = false;
void setCoarseHalfStorage(bool on) { g_mgHalf = on; }
bool coarseHalfStorage() { return g_mgHalf; }

struct CVec { float *v[2]; int stride, Vh; };

static CVec cvecFull(ColorSpinorField &f) {
  CVec r;
  r.v[0] = (float *)f.Even().V(); r.v[1] = (float *)f.Odd().V(); r.stride = f.Stride(); r.Vh = f.VolumeCB();
  return r;
}

struct CoarseArg {
  CVec out, in;
  const void *G;     // float4 (fp32) or half4 (fp16 mirror) per (row, column pair)
  int Xc[4];
  int n, mmask, parity, nwork;
  // grid-decomposed lattice: ghost[m] = the neighbour rank's face needed by hop m (full coarse spinors, both or one
  // parity, [q][component][face site] float2); commMask bit mu = dimension mu is partitioned
  const float2 *ghost[8];
  int faceCB[4];
  int commMask, ghostSingle;
  // peer-store transport: counter each crossing hop's face must have reached (0 pointers: data arrived in stream order)
  const unsigned *waitFlag[8];
  unsigned waitCount[8];
  unsigned long long waitTicks;
  int *errWord;              // error record (p2p.h kP2pErrInts)
  unsigned exSeq; int exBuf; // exchange number / buffer of this launch, for that record
};

template <int NMAX, bool HALF>
__global__ void __launch_bounds__(256) coarse_apply_kernel(const CoarseArg arg) {
  __shared__ float2 xin[9][NMAX];
  __shared__ float2 part[4][NMAX];
  const int n = arg.n, Vh = arg.out.Vh;
  // site handled by this work-group (optionally one parity only)
  int A = blockIdx.x;
  if (arg.parity >= 0) A += arg.parity * Vh;
  const int par = A >= Vh, xcb = A - par * Vh;
  // coordinates of the coarse site
  const int Xh = arg.Xc[0] >> 1;
  int l = xcb;
  const int xh = l % Xh; l /= Xh;
  const int y = l % arg.Xc[1]; l /= arg.Xc[1];
  const int z = l % arg.Xc[2]; const int t = l / arg.Xc[2];
  const int c[4] = {2 * xh + ((y + z + t + par) & 1), y, z, t};
  // peer-store transport: a site with a hop across a partitioned face waits for that face (threads 0-7 poll one counter each)
  if (arg.commMask) {
    bool need = false;
    if (threadIdx.x < 8 && ((arg.mmask >> threadIdx.x) & 1) && ((arg.commMask >> (threadIdx.x >> 1)) & 1) && arg.waitFlag[threadIdx.x]) {
      const int mu = threadIdx.x >> 1;
      need = (threadIdx.x & 1) ? c[mu] == 0 : c[mu] == arg.Xc[mu] - 1;
    }
    if (need && !__hip_atomic_load(arg.errWord, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
      const unsigned long long t0 = wall_clock64();
      while ((int)(__hip_atomic_load(arg.waitFlag[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - arg.waitCount[threadIdx.x]) < 0) {
        if (wall_clock64() - t0 > arg.waitTicks) {
          if (atomicCAS(arg.errWord, 0, 17 + (int)threadIdx.x) == 0) {
            const unsigned seen = __hip_atomic_load(arg.waitFlag[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            arg.errWord[1] = A; arg.errWord[2] = (int)arg.waitCount[threadIdx.x]; arg.errWord[3] = (int)seen; arg.errWord[4] = (int)seen;
            arg.errWord[5] = (int)arg.exSeq; arg.errWord[6] = arg.exBuf; arg.errWord[7] = 0;
          }
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");   // once, after the poll: the face words read below are newer than the counter
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
  }
  // stage the 9 input vectors (8 neighbours + self)
  for (int m = 0; m < 9; m++) {
    if (!((arg.mmask >> m) & 1)) continue;
    int cn[4] = {c[0], c[1], c[2], c[3]};
    int npar = par;
    if (m < 8) {
      const int mu = m >> 1, L = arg.Xc[mu];
      cn[mu] = (m & 1) ? (c[mu] == 0 ? L - 1 : c[mu] - 1) : (c[mu] == L - 1 ? 0 : c[mu] + 1);
      npar = (cn[0] + cn[1] + cn[2] + cn[3]) & 1;
    }
    bool cross = false;
    if (m < 8 && ((arg.commMask >> (m >> 1)) & 1)) {
      const int mu = m >> 1;
      cross = (m & 1) ? c[mu] == 0 : c[mu] == arg.Xc[mu] - 1;
    }
    if (cross) {
      const int mu = m >> 1;
      int l = 0, mul = 1;
      for (int k = 0; k < 4; k++) if (k != mu) { l += cn[k] * mul; mul *= arg.Xc[k]; }
      const int f = l >> 1, q = arg.ghostSingle ? 0 : npar;
      const float2 *src = arg.ghost[m] + (size_t)q * n * arg.faceCB[mu] + f;
      // system-scope loads: the zone may have been written by another GPU while this kernel was running
      for (int j = threadIdx.x; j < n; j += blockDim.x)
        xin[m][j] = __builtin_bit_cast(float2, __hip_atomic_load(reinterpret_cast<const unsigned long long *>(src + (size_t)j * arg.faceCB[mu]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
    } else {
      const int nx = (((cn[3] * arg.Xc[2] + cn[2]) * arg.Xc[1] + cn[1]) * arg.Xc[0] + cn[0]) >> 1;
      const float *src = arg.in.v[npar];
      for (int j = threadIdx.x; j < n; j += blockDim.x) {
        const float *p = src + ((size_t)j * arg.in.stride + nx) * 2;
        xin[m][j] = make_float2(p[0], p[1]);
      }
    }
  }
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float re = 0.f, im = 0.f;
  if (lane < n) {
    for (int m = wave; m < 9; m += 4) {
      if (!((arg.mmask >> m) & 1)) continue;
      const size_t mbase = ((size_t)A * 9 + m) * (n / 2) * n + lane;
      for (int jp = 0; jp < n / 2; jp++) {
        float4 w;
        if (HALF) {
          const half4_t h = reinterpret_cast<const half4_t *>(arg.G)[mbase + (size_t)jp * n];
          w = make_float4((float)h.x, (float)h.y, (float)h.z, (float)h.w);
        } else {
          w = reinterpret_cast<const float4 *>(arg.G)[mbase + (size_t)jp * n];
        }
        const float2 a = xin[m][2 * jp], b = xin[m][2 * jp + 1];
        re += w.x * a.x - w.y * a.y + w.z * b.x - w.w * b.y;
        im += w.x * a.y + w.y * a.x + w.z * b.y + w.w * b.x;
      }
    }
    part[wave][lane] = make_float2(re, im);
  }
  __syncthreads();
  if (wave == 0 && lane < n) {
    const float2 s = make_float2(part[0][lane].x + part[1][lane].x + part[2][lane].x + part[3][lane].x,
                                 part[0][lane].y + part[1][lane].y + part[2][lane].y + part[3][lane].y);
    float *o = arg.out.v[par] + ((size_t)lane * arg.out.stride + xcb) * 2;
    o[0] = s.x; o[1] = s.y;
  }
}

// ---- ghost zones of coarse fields (reference: full coarse spinors exchanged blocking, lib/dslash_coarse.cu:68-137, :707) ----
struct CoarseGhost {
  int Xc[4], n;
  int faceCB[4];
  float2 *pool;
  float2 *send[4][2], *ghost[4][2];   // [mu][0: x_mu = 0 face / from the -mu neighbour, 1: x_mu = L-1 face / from the +mu neighbour]
  // peer-store transport (p2p.h, same protocol as the fine stencil's halo): fine-grained window with double-buffered ghost
  // blocks [mu][k][buf] + one cumulative thread counter per block, mapped into the neighbours
  bool p2p;
  char *window;
  float2 *ghostBuf[4][2][2], *peerGhost[4][2][2];   // peerGhost[mu][face 0/1][buf]: where my x_mu = 0 / L-1 face lands
  unsigned *flag[4][2], *peerFlag[4][2];             // + buf
  unsigned seq, uses[4][2][2];
  PeerMap *map;
  int verified;   // 0: the peer-store exchange of this ghost has not been checked against the staged one yet, 1: it has, 3: being checked
};
static std::vector<CoarseGhost> g_cghosts;
void freeCoarseGhosts() {
  bool any = false;
  for (CoarseGhost &c : g_cghosts) any = any || c.window;
  if (any) { HIP_CHECK(hipDeviceSynchronize()); commBarrier(); }
  for (CoarseGhost &c : g_cghosts) {
    if (c.window) { commUnmapPeers(*c.map); delete c.map; }
  }
  if (any) commBarrier();
  for (CoarseGhost &c : g_cghosts) {
    if (c.window) p2pFree(c.window);
    if (c.pool) (void)hipFree(c.pool);
  }
  g_cghosts.clear();
}
static CoarseGhost &coarseGhost(const int Xc[4], int n) {
  for (CoarseGhost &c : g_cghosts)
    if (c.n == n && c.Xc[0] == Xc[0] && c.Xc[1] == Xc[1] && c.Xc[2] == Xc[2] && c.Xc[3] == Xc[3]) return c;
  CoarseGhost c;
  c.n = n;
  size_t total = 0;
  const int Vh = Xc[0] * Xc[1] * Xc[2] * Xc[3] / 2;
  for (int d = 0; d < 4; d++) { c.Xc[d] = Xc[d]; c.faceCB[d] = Vh / Xc[d]; total += (size_t)4 * 2 * n * c.faceCB[d]; }
  HIP_CHECK(qaMalloc((void **)&c.pool, total * sizeof(float2)));
  HIP_CHECK(hipMemsetAsync(c.pool, 0, total * sizeof(float2), computeStream()));
  float2 *p = c.pool;
  for (int d = 0; d < 4; d++)
    for (int k = 0; k < 2; k++) { c.send[d][k] = p; p += (size_t)2 * n * c.faceCB[d]; c.ghost[d][k] = p; p += (size_t)2 * n * c.faceCB[d]; }
  c.p2p = p2pHaloEnabled();
  c.window = nullptr; c.map = nullptr; c.seq = 0; c.verified = 0;
  memset(c.uses, 0, sizeof(c.uses));
  if (c.p2p) {
    size_t wbytes = 0;
    for (int d = 0; d < 4; d++) wbytes += (size_t)4 * 2 * n * c.faceCB[d] * sizeof(float2);   // [k][buf] blocks of both parities
    const size_t flags_off = (wbytes + 255) / 256 * 256;
    c.window = (char *)p2pAlloc(flags_off + 256);
    c.map = new PeerMap;
    if (!commMapPeers(c.window, *c.map)) errorQuda("peer mapping of a coarse halo window failed after the transport probe succeeded");
    size_t off = 0;
    for (int d = 0; d < 4; d++)
      for (int k = 0; k < 2; k++) {
        // zone k = 0: filled by the -d neighbour with its x_d = L-1 face (needed by my backward hops);
        // zone k = 1: filled by the +d neighbour with its x_d = 0 face (needed by my forward hops)
        const int face = k == 0 ? 1 : 0;          // which of MY faces goes into the neighbour's zone k
        const int slot = 2 * d + (face ? 1 : 0);  // face L-1 travels to the +d neighbour, face 0 to the -d neighbour
        for (int buf = 0; buf < 2; buf++) {
          c.ghostBuf[d][k][buf] = (float2 *)(c.window + off);
          c.peerGhost[d][face][buf] = (float2 *)((char *)c.map->peer[slot] + off);
          off += (size_t)2 * n * c.faceCB[d] * sizeof(float2);
        }
        c.flag[d][k] = (unsigned *)(c.window + flags_off) + (2 * d + k) * 2;
        c.peerFlag[d][face] = (unsigned *)((char *)c.map->peer[slot] + flags_off) + (2 * d + k) * 2;
      }
    commBarrier();
  }
  g_cghosts.push_back(c);
  return g_cghosts.back();
}

struct CoarsePackArg {
  CVec in;
  float2 *send[8];   // slot 2 mu + (0: x_mu = 0 face, 1: x_mu = L-1 face); nullptr = not packed
  int start[9];
  int Xc[4], faceCB[4];
  int n, single;     // single >= 0: only that parity of `in` exists
  unsigned *peerFlag[8];   // peer-store transport: thread counters of the faces in the neighbours' windows (send[] then point there too)
  int p2p;
};
constexpr int kCoarsePackChunk = 4;   // divides every supported n (16, 32, 48, 64); the arrival counters count (face site, chunk) pairs
__global__ void __launch_bounds__(256) coarse_pack_kernel(const CoarsePackArg arg) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  if (tid < arg.start[8]) {
  int slot = 0;
  for (int k = 1; k < 8; k++) slot += tid >= arg.start[k];
  const int mu = slot >> 1, nf = arg.faceCB[mu];
  const int local = tid - arg.start[slot];
  const int q = local / nf, f = local - q * nf;
  const int par = arg.single >= 0 ? arg.single : q;
  int c[4], L[3], o[3], k3 = 0;
  for (int k = 0; k < 4; k++) if (k != mu) { L[k3] = arg.Xc[k]; o[k3] = k; k3++; }
  int l = 2 * f;
  const int c0 = l % L[0]; l /= L[0];
  const int c1 = l % L[1]; const int c2 = l / L[1];
  c[mu] = (slot & 1) ? arg.Xc[mu] - 1 : 0;
  c[o[0]] = c0; c[o[1]] = c1; c[o[2]] = c2;
  c[o[0]] += (par + c[0] + c[1] + c[2] + c[3]) & 1;
  const int idx = (((c[3] * arg.Xc[2] + c[2]) * arg.Xc[1] + c[1]) * arg.Xc[0] + c[0]) >> 1;
  const float2 *src = reinterpret_cast<const float2 *>(arg.in.v[par]) + idx;
  float2 *dst = arg.send[slot] + (size_t)q * arg.n * nf + f;
  // kCoarsePackChunk components per thread (blockIdx.y picks the chunk), loads first: one thread per face site walking all n components
  // was n dependent load -> store round trips — 18 us per launch on the 8 x 4 x 4 x 4 coarse lattice of an 8-GPU split, a fifth of the
  // whole MG-GCR solve there (profiles/r03j_sub8_masked_mg_solve.log)
  const int j0 = (int)blockIdx.y * kCoarsePackChunk;
  float2 v[kCoarsePackChunk];
#pragma unroll
  for (int jj = 0; jj < kCoarsePackChunk; jj++) v[jj] = src[(size_t)(j0 + jj) * arg.in.stride];
  if (arg.p2p) {
    // system-scope write-through stores into the neighbour's window (8-byte granules)
#pragma unroll
    for (int jj = 0; jj < kCoarsePackChunk; jj++)
      __hip_atomic_store(reinterpret_cast<unsigned long long *>(dst + (size_t)(j0 + jj) * nf), __builtin_bit_cast(unsigned long long, v[jj]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  } else {
#pragma unroll
    for (int jj = 0; jj < kCoarsePackChunk; jj++) dst[(size_t)(j0 + jj) * nf] = v[jj];
  }
  }
  if (arg.p2p) {
    // every storing wave waits until its write-through stores have been acknowledged, THEN the barrier, THEN one lane per
    // face adds the threads this block contributed to the neighbour's arrival counter (release, system scope)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x < 8 && arg.peerFlag[threadIdx.x]) {
      const int beg = blockIdx.x * (int)blockDim.x, end = beg + (int)blockDim.x;
      const int lo = beg > arg.start[threadIdx.x] ? beg : arg.start[threadIdx.x];
      const int hi = end < arg.start[threadIdx.x + 1] ? end : arg.start[threadIdx.x + 1];
      if (hi > lo) (void)__hip_atomic_fetch_add(arg.peerFlag[threadIdx.x], (unsigned)(hi - lo), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// exchange the faces of `in` the masked hops need; fills arg.ghost / commMask
static void exchangeCoarseGhost(CoarseArg &arg, const CoarseGauge &G, int single) {
  arg.commMask = 0; arg.ghostSingle = single >= 0;
  arg.waitTicks = 0; arg.errWord = nullptr; arg.exSeq = 0; arg.exBuf = 0;
  for (int m = 0; m < 8; m++) { arg.ghost[m] = nullptr; arg.waitFlag[m] = nullptr; arg.waitCount[m] = 0; }
  int mask = 0;
  for (int d = 0; d < 4; d++) if (commGrid().partitioned(d) && ((arg.mmask >> (2 * d)) & 3)) mask |= 1 << d;
  for (int d = 0; d < 4; d++) arg.faceCB[d] = G.nSites / 2 / G.Xc[d];
  if (!mask) return;
  CoarseGhost &cg = coarseGhost(G.Xc, G.n);
  CoarsePackArg pa;
  pa.in = arg.in; pa.n = G.n; pa.single = single; pa.p2p = cg.p2p ? 1 : 0;
  for (int k = 0; k < 8; k++) pa.peerFlag[k] = nullptr;
  const int nq = single >= 0 ? 1 : 2;
  int nt = 0;
  std::vector<HaloMsg> msgs;
  const int buf = cg.p2p ? (int)(++cg.seq & 1) : 0;
  arg.exSeq = cg.seq; arg.exBuf = buf;
  p2pStats()[cg.p2p ? 2 : 3]++;
  for (int d = 0; d < 4; d++) {
    pa.Xc[d] = G.Xc[d]; pa.faceCB[d] = cg.faceCB[d];
    const size_t bytes = (size_t)nq * G.n * cg.faceCB[d] * sizeof(float2);
    // hop 2d (forward) reads the +d neighbour's x_d = 0 face: every rank sends that face backward
    const bool needFwd = ((mask >> d) & 1) && ((arg.mmask >> (2 * d)) & 1), needBwd = ((mask >> d) & 1) && ((arg.mmask >> (2 * d + 1)) & 1);
    pa.send[2 * d] = cg.p2p ? cg.peerGhost[d][0][buf] : cg.send[d][0]; pa.start[2 * d] = nt; if (needFwd) nt += nq * cg.faceCB[d];
    pa.send[2 * d + 1] = cg.p2p ? cg.peerGhost[d][1][buf] : cg.send[d][1]; pa.start[2 * d + 1] = nt; if (needBwd) nt += nq * cg.faceCB[d];
    if (cg.p2p) {
      // my x_d = 0 face (needed by the -d neighbour's forward hops) lands in its zone 1, my x_d = L-1 face in the +d neighbour's zone 0
      if (needFwd) {
        pa.peerFlag[2 * d] = cg.peerFlag[d][0] + buf;
        arg.ghost[2 * d] = cg.ghostBuf[d][1][buf];
        arg.waitFlag[2 * d] = cg.flag[d][1] + buf;
        arg.waitCount[2 * d] = (cg.uses[d][1][buf] += (unsigned)(nq * cg.faceCB[d] * (G.n / kCoarsePackChunk)));
      }
      if (needBwd) {
        pa.peerFlag[2 * d + 1] = cg.peerFlag[d][1] + buf;
        arg.ghost[2 * d + 1] = cg.ghostBuf[d][0][buf];
        arg.waitFlag[2 * d + 1] = cg.flag[d][0] + buf;
        arg.waitCount[2 * d + 1] = (cg.uses[d][0][buf] += (unsigned)(nq * cg.faceCB[d] * (G.n / kCoarsePackChunk)));
      }
    } else {
      if (needBwd) { msgs.push_back({d, +1, cg.send[d][1], cg.ghost[d][0], bytes}); arg.ghost[2 * d + 1] = cg.ghost[d][0]; }
      if (needFwd) { msgs.push_back({d, -1, cg.send[d][0], cg.ghost[d][1], bytes}); arg.ghost[2 * d] = cg.ghost[d][1]; }
    }
  }
  pa.start[8] = nt;
  if (G.n % kCoarsePackChunk) errorQuda("coarse halo pack: n = %d is not a multiple of %d", G.n, kCoarsePackChunk);
  hipLaunchKernelGGL(coarse_pack_kernel, dim3((nt + 255) / 256, G.n / kCoarsePackChunk), dim3(256), 0, computeStream(), pa);
  HIP_CHECK(hipGetLastError());
  if (cg.p2p) { arg.waitTicks = p2pTimeoutTicks(); arg.errWord = p2pErrorWord(); }
  else commExchange(msgs, computeStream());
  arg.commMask = mask;
}

void applyCoarse(ColorSpinorField &out, const ColorSpinorField &in, const CoarseGauge &G, int mmask, int parity) {
  if (out.Precision() != QUDA_SINGLE_PRECISION || in.Precision() != QUDA_SINGLE_PRECISION) errorQuda("coarse operator is fp32");
  {
    // first partitioned application through a peer-store ghost: once staged, once with peer stores, and every rank must see the
    // same field — otherwise all ranks keep the staged exchange (the fine stencil does the same, dslash.hip)
    int mask = 0;
    for (int d = 0; d < 4; d++) if (commGrid().partitioned(d) && ((mmask >> (2 * d)) & 3)) mask |= 1 << d;
    if (mask) {
      CoarseGhost &cg = coarseGhost(G.Xc, G.n);
      if (cg.p2p && cg.verified == 0) {
        cg.verified = 3;
        cg.p2p = false;
        applyCoarse(out, in, G, mmask, parity);
        HIP_CHECK(hipStreamSynchronize(computeStream()));
        ColorSpinorField ref(out);
        CoarseGhost &cg2 = coarseGhost(G.Xc, G.n);   // (the vector may have grown in between)
        cg2.p2p = true;
        applyCoarse(out, in, G, mmask, parity);
        HIP_CHECK(hipStreamSynchronize(computeStream()));
        double fail = p2pTakeError() ? 1.0 : 0.0;
        const bool wasGlobal = blas::globalReduction();
        blas::setGlobalReduction(false);
        const double n2 = blas::norm2(out), d2 = blas::xmyNorm(out, ref);
        blas::setGlobalReduction(wasGlobal);
        if (!(d2 <= 1e-8 * n2)) fail = 1.0;
        if (getenv("QUDA_AMD_P2P_VERIFY_FAIL")) fail = 1.0;
        comm_allreduce(&fail, 1);
        CoarseGhost &cg3 = coarseGhost(G.Xc, G.n);
        if (fail != 0.0) {
          if (commGrid().rank == 0) warningQuda("peer-store coarse halo disagrees with the staged exchange on its first use: staying with RCCL send/recv");
          for (CoarseGhost &c : g_cghosts) { c.p2p = false; c.verified = 1; }
          applyCoarse(out, in, G, mmask, parity);
          return;
        }
        cg3.verified = 1;
        return;
      }
    }
  }
  if (in.Nspin() != 2 || 2 * in.Ncolor() != G.n) errorQuda("coarse field (%d spins, %d colours) does not match the operator (n = %d)", in.Nspin(), in.Ncolor(), G.n);
  if (G.n > 64) errorQuda("2 Nc = %d exceeds one wavefront", G.n);
  if (in.V() == out.V()) errorQuda("in and out must not alias");
  CoarseArg arg;
  ColorSpinorField &o = out, &i = const_cast<ColorSpinorField &>(in);
  if (parity < 0) {
    if (out.SiteSubset() != QUDA_FULL_SITE_SUBSET || in.SiteSubset() != QUDA_FULL_SITE_SUBSET) errorQuda("full coarse fields required");
    arg.out = cvecFull(o); arg.in = cvecFull(i);
    arg.nwork = G.nSites;
  } else {
    // parity fields: `out` lives on `parity`; hops read the other parity, the local term the same parity
    if (out.SiteSubset() != QUDA_PARITY_SITE_SUBSET || in.SiteSubset() != QUDA_PARITY_SITE_SUBSET) errorQuda("parity coarse fields required");
    const bool local_only = mmask == (1 << 8);
    arg.out.v[parity] = (float *)o.V(); arg.out.v[1 - parity] = nullptr; arg.out.stride = o.Stride(); arg.out.Vh = o.VolumeCB();
    arg.in.v[local_only ? parity : 1 - parity] = (float *)i.V(); arg.in.v[local_only ? 1 - parity : parity] = nullptr;
    arg.in.stride = i.Stride(); arg.in.Vh = i.VolumeCB();
    if (!local_only && (mmask & (1 << 8))) errorQuda("hop + local on parity fields needs both parities of the input");
    arg.nwork = G.nSites / 2;
  }
  const bool half = g_mgHalf && G.data_h != nullptr;
  arg.G = half ? (const void *)G.data_h : (const void *)G.data;
  for (int d = 0; d < 4; d++) arg.Xc[d] = G.Xc[d];
  arg.n = G.n; arg.mmask = mmask; arg.parity = parity;
  if (g_acctOn) {
    // links of the matrices in the mask (once per work site) + one input vector per matrix (ideal re-use: each vector once) + output
    int nm = 0;
    for (int m = 0; m < 9; m++) nm += (mmask >> m) & 1;
    char tag[64];
    snprintf(tag, sizeof(tag), "coarse %dx%dx%dx%d n %d mask 0x%x%s", G.Xc[0], G.Xc[1], G.Xc[2], G.Xc[3], G.n, mmask, half ? " fp16 links" : "");
    acct("coarse_apply_kernel", (double)arg.nwork * ((double)nm * G.n * G.n * (half ? 4 : 8) + 2.0 * G.n * 8), tag);
  }
  exchangeCoarseGhost(arg, G, parity < 0 ? -1 : 1 - parity);
  if (half) hipLaunchKernelGGL((coarse_apply_kernel<64, true>), dim3(arg.nwork), dim3(256), 0, computeStream(), arg);
  else hipLaunchKernelGGL((coarse_apply_kernel<64, false>), dim3(arg.nwork), dim3(256), 0, computeStream(), arg);
  HIP_CHECK(hipGetLastError());
}

// ---- construction helpers ----
// column j of matrix m at every site (+)= coarse vector c
__global__ void insert_column_kernel(float *G, CVec c, int n, int m, int j, int accumulate, int nSites) {
  const long t = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (t >= (long)nSites * n) return;
  const int A = (int)(t / n), i = (int)(t - (long)A * n);
  const int par = A >= c.Vh, x = A - par * c.Vh;
  const float *p = c.v[par] + ((size_t)i * c.stride + x) * 2;
  float *g = G + ((((size_t)A * 9 + m) * (n / 2) + j / 2) * n + i) * 4 + (j & 1) * 2;
  if (accumulate) { g[0] += p[0]; g[1] += p[1]; }
  else { g[0] = p[0]; g[1] = p[1]; }
}

// ---- Galerkin construction from the forward hops only (reference: calculateY builds the backward links from the forward ones
// unless bidirectional links are asked for, lib/coarse_op.cuh:1310-1420).  With the fine hop term gamma5-hermitian,
// H_{-mu}(x + mu, x) = g5 H_{+mu}(x, x + mu)^dagger g5, and V made of vectors of definite chirality (g5 V_j = s_j V_j, s = +1 / -1
// for the upper / lower coarse spin):
//     Y_{2mu+1}(X + mu)[i][j] = s_i s_j conj( Y_{2mu}(X)[j][i] )
//     X_hop(X)               = S(X) + G5 S(X)^dagger G5,   S = sum over the forward hops that stay inside the aggregate
// so a probe needs the four forward hops and ONE pass over V (restrict4_kernel) instead of eight hops and three passes. ----
// backward links from the forward links of the neighbour behind
__global__ void galerkin_backward_kernel(float *G, const int *nbr, int n, int nvec, long total) {
  const long t = blockIdx.x * (long)blockDim.x + threadIdx.x;   // (site B, mu, column pair jp, row i)
  if (t >= total) return;
  const int i = (int)(t % n);
  long r = t / n;
  const int jp = (int)(r % (n / 2)); r /= (n / 2);
  const int mu = (int)(r & 3);
  const long B = r >> 2;
  const long A = nbr[9 * B + 2 * mu + 1];   // B - mu
  const float4 *src = reinterpret_cast<const float4 *>(G) + ((size_t)A * 9 + 2 * mu) * (n / 2) * n;
  const float si = (i < nvec) ? 1.f : -1.f;
  float o[4];
#pragma unroll
  for (int h = 0; h < 2; h++) {
    const int j = 2 * jp + h;
    const float sj = (j < nvec) ? 1.f : -1.f;
    // Y_fwd(A)[j][i]: row j, column i -> float4 at (column pair i / 2, row j), component pair i & 1
    const float4 v = src[(size_t)(i >> 1) * n + j];
    const float re = (i & 1) ? v.z : v.x, im = (i & 1) ? v.w : v.y;
    o[2 * h] = si * sj * re; o[2 * h + 1] = -si * sj * im;
  }
  reinterpret_cast<float4 *>(G)[(((size_t)B * 9 + 2 * mu + 1) * (n / 2) + jp) * n + i] = make_float4(o[0], o[1], o[2], o[3]);
}
// local matrix: X = S + G5 S^dagger G5 + diag(d_up, d_down)   (S read from a copy: the transpose reads what other threads write)
__global__ void galerkin_local_kernel(float *G, const float4 *S, int n, int nvec, float2 dUp, float2 dDown, long total) {
  const long t = blockIdx.x * (long)blockDim.x + threadIdx.x;   // (site A, column pair jp, row i)
  if (t >= total) return;
  const int i = (int)(t % n);
  long r = t / n;
  const int jp = (int)(r % (n / 2));
  const long A = r / (n / 2);
  const float4 *Sa = S + (size_t)A * (n / 2) * n;
  const float si = (i < nvec) ? 1.f : -1.f;
  const float4 s = Sa[(size_t)jp * n + i];
  float o[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
  for (int h = 0; h < 2; h++) {
    const int j = 2 * jp + h;
    const float sj = (j < nvec) ? 1.f : -1.f;
    const float4 v = Sa[(size_t)(i >> 1) * n + j];
    const float re = (i & 1) ? v.z : v.x, im = (i & 1) ? v.w : v.y;
    o[2 * h] += si * sj * re; o[2 * h + 1] -= si * sj * im;
    if (i == j) { const float2 d = (i < nvec) ? dUp : dDown; o[2 * h] += d.x; o[2 * h + 1] += d.y; }
  }
  reinterpret_cast<float4 *>(G)[(((size_t)A * 9 + 8) * (n / 2) + jp) * n + i] = make_float4(o[0], o[1], o[2], o[3]);
}
__global__ void galerkin_copy_local_kernel(float4 *S, const float *G, int n, long total) {
  const long t = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (t >= total) return;
  const long per = (long)(n / 2) * n, A = t / per, e = t - A * per;
  S[t] = reinterpret_cast<const float4 *>(G)[((size_t)A * 9 + 8) * per + e];
}

void DiracCoarse::build() {
  const Transfer &T = *transfer;
  const int n = 2 * T.Nvec;
  links = new CoarseGauge(T.Xc, n);
  ownLinks = true;
  ColorSpinorField *c = T.createCoarseField(), *c2 = T.createCoarseField();
  ColorSpinorField *phi = T.createFineField(), *w = T.createFineField();
  phi->twistFlavor = w->twistFlavor = fineFlavor;
  const int bs = 256;
  const unsigned nins = (unsigned)(((long)links->nSites * n + bs - 1) / bs);
  // first coarse level: the 8 hop outputs of a probe are restricted four at a time (V streamed 3 instead of 9 times per probe)
  const bool four = T.canSplit4();
  ColorSpinorField *w8[8] = {}, *cl[4] = {}, *cs[4] = {};
  if (four) {
    for (int d = 0; d < 8; d++) { w8[d] = T.createFineField(); w8[d]->twistFlavor = fineFlavor; }
    for (int q = 0; q < 4; q++) { cl[q] = T.createCoarseField(); cs[q] = T.createCoarseField(); }
  }
  // forward hops only + hermitian completion where the fine operator allows it: Wilson / twisted mass (local term (1 + i a g5):
  // its coarse image is (1 +- i a) V^dagger V = a multiple of the identity per chirality, read off two probes), unpartitioned
  // A coarse parent (second coarsening) has the same structure in its hop part — its backward links ARE the g5-conjugates of its
  // forward ones — but a dense local matrix: that one is probed as before, after the completion.
  bool herm = false, analyticLocal = false;
  {
    static int full = -1;
    if (full < 0) { const char *e = getenv("QUDA_AMD_GALERKIN_FULL"); full = e ? atoi(e) : 0; }
    const QudaDiracType pt = parent->getDiracType();
    analyticLocal = four && (pt == QUDA_WILSON_DIRAC || pt == QUDA_TWISTED_MASS_DIRAC);
    herm = !full && (analyticLocal || (!four && pt == QUDA_COARSE_DIRAC));
    for (int d = 0; d < 4; d++) if (commGrid().partitioned(d)) herm = false;
  }
  // fine parent with plain hop term (Wilson / twisted mass / twisted clover): the forward links and S as ONE batched product per direction
  // on the matrix cores instead of 2 Nvec probes (Transfer::directGalerkinVUV); twisted clover keeps probing for its local term only
  bool direct = false;
  {
    const QudaDiracType pt = parent->getDiracType();
    direct = T.canDirectGalerkin() && (pt == QUDA_WILSON_DIRAC || pt == QUDA_TWISTED_MASS_DIRAC || pt == QUDA_TWISTED_CLOVER_DIRAC) && parent->Gauge() &&
             parent->Gauge()->precision == QUDA_SINGLE_PRECISION && (parent->Gauge()->reconstruct == QUDA_RECONSTRUCT_NO || parent->Gauge()->reconstruct == QUDA_RECONSTRUCT_12);
    static int full = -1;
    if (full < 0) { const char *e = getenv("QUDA_AMD_GALERKIN_FULL"); full = e ? atoi(e) : 0; }
    if (full) direct = false;
    if (direct && pt == QUDA_TWISTED_CLOVER_DIRAC) herm = true;   // hop part by completion, local term probed below (analyticLocal is false)
  }
  if (herm && direct) {
    // in chunks of aggregates (the product of an aggregate needs UV on its own sites only): the temporary is ~3 GB instead of a second V
    // (24.5 GB at 48^3 x 96, whose allocation alone cost more than the products) and fits a buffer the null-vector stage left in the pool
    const int chunk = (int)std::min<long>(T.nAgg, std::max<long>(512, (T.nAgg + 7) / 8));
    const size_t wbytes = (size_t)chunk * 12 * T.Nvec * T.blockVol * 2 * sizeof(float);
    static int cmEnv = -1;
    if (cmEnv < 0) { const char *e = getenv("QUDA_AMD_GALERKIN_CLASS_MAJOR"); cmEnv = e ? atoi(e) : 0; }
    // UV (this routine's own temporary) with the sites of a row ordered by the class of galerkin_vuv_kernel's waves.  MEASURED at 48^3 x 96: coarse operator
    // 0.172 s with it, 0.166 s without — what helped was walking a wave's sites inside one spin-colour row (0.216 -> 0.166 s, transfer.hip
    // galerkin_vuv_accumulate); the scattered stores of the class-major UV cost what its reads gain.  Off by default.
    const bool classMajor = cmEnv != 0;
    float *UV = (float *)poolDeviceMalloc(wbytes);
    for (long a0 = 0; a0 < T.nAgg; a0 += chunk) {
      const int na = (int)std::min<long>(chunk, T.nAgg - a0);
      for (int mu = 0; mu < 4; mu++) {
        galerkinUV(UV, T.V, *parent->Gauge(), 2 * mu, -parent->Kappa(), T.block_to_fine, T.fine_to_block, (int)a0, na, T.blockVol, T.Nvec, classMajor);
        T.directGalerkinVUV(links->data, UV, mu, mu > 0, false, (int)a0, na, classMajor);
      }
    }
    HIP_CHECK(hipStreamSynchronize(computeStream()));
    poolDeviceFree(UV, wbytes);
  }
  if (herm) {
    const int fdirs[4] = {0, 2, 4, 6};
    for (int j = 0; j < (direct ? 0 : n); j++) {
      T.column(*phi, j);
      if (four) {
        for (int q = 0; q < 4; q++) parent->hopDir(*w8[q], *phi, fdirs[q]);
        ColorSpinorField *in4[4] = {w8[0], w8[1], w8[2], w8[3]};
        T.RSplit4(cl, cs, in4, fdirs);
        for (int q = 0; q < 4; q++) {
          hipLaunchKernelGGL(insert_column_kernel, dim3(nins), dim3(bs), 0, computeStream(), links->data, cvecFull(*cl[q]), n, fdirs[q], j, 0, links->nSites);
          hipLaunchKernelGGL(insert_column_kernel, dim3(nins), dim3(bs), 0, computeStream(), links->data, cvecFull(*cs[q]), n, 8, j, 1, links->nSites);
        }
      } else {
        for (int q = 0; q < 4; q++) {
          parent->hopDir(*w, *phi, fdirs[q]);
          T.RSplit(*c, *c2, *w, fdirs[q]);
          hipLaunchKernelGGL(insert_column_kernel, dim3(nins), dim3(bs), 0, computeStream(), links->data, cvecFull(*c), n, fdirs[q], j, 0, links->nSites);
          hipLaunchKernelGGL(insert_column_kernel, dim3(nins), dim3(bs), 0, computeStream(), links->data, cvecFull(*c2), n, 8, j, 1, links->nSites);
        }
      }
    }
    // the local term's diagonal per chirality from one probe each (robust against the sign conventions of flavour and dagger)
    float2 dloc[2] = {make_float2(0.f, 0.f), make_float2(0.f, 0.f)};
    for (int chi = 0; analyticLocal && chi < 2; chi++) {
      const int j = chi * T.Nvec;
      T.column(*phi, j);
      parent->localTerm(*w, *phi);
      T.R(*c, *w);
      HIP_CHECK(hipStreamSynchronize(computeStream()));
      // component j of the coarse vector at coarse site 0 (even parity, x_cb = 0)
      const CVec cv = cvecFull(*c);
      float h[2];
      HIP_CHECK(hipMemcpy(h, cv.v[0] + ((size_t)j * cv.stride + 0) * 2, 2 * sizeof(float), hipMemcpyDeviceToHost));
      dloc[chi] = make_float2(h[0], h[1]);
    }
    const long per = (long)(n / 2) * n;
    float4 *S = (float4 *)poolDeviceMalloc((size_t)links->nSites * per * sizeof(float4));
    const long nS = (long)links->nSites * per;
    hipLaunchKernelGGL(galerkin_copy_local_kernel, dim3((unsigned)((nS + 255) / 256)), dim3(256), 0, computeStream(), S, links->data, n, nS);
    hipLaunchKernelGGL(galerkin_local_kernel, dim3((unsigned)((nS + 255) / 256)), dim3(256), 0, computeStream(), links->data, S, n, T.Nvec, dloc[0], dloc[1], nS);
    const long nB = 4 * nS;
    hipLaunchKernelGGL(galerkin_backward_kernel, dim3((unsigned)((nB + 255) / 256)), dim3(256), 0, computeStream(), links->data, coarseNeighbourTable(links->Xc), n, T.Nvec, nB);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipStreamSynchronize(computeStream()));
    poolDeviceFree(S, (size_t)links->nSites * per * sizeof(float4));
    if (!analyticLocal && direct && parent->Clover() && parent->Clover()->precision == QUDA_SINGLE_PRECISION) {
      // twisted clover: V^dagger (A + i a g5) V as one more batched product, added to the completed hop part
      const int chunk = (int)std::min<long>(T.nAgg, std::max<long>(512, (T.nAgg + 7) / 8));
      const size_t wbytes = (size_t)chunk * 12 * T.Nvec * T.blockVol * 2 * sizeof(float);
      float *L = (float *)poolDeviceMalloc(wbytes);
      for (long a0 = 0; a0 < T.nAgg; a0 += chunk) {
        const int na = (int)std::min<long>(chunk, T.nAgg - a0);
        galerkinLocalUV(L, T.V, *parent->Clover(), 2.0 * parent->Kappa() * (double)fineFlavor * parent->Mu(), T.block_to_fine, (int)a0, na, T.blockVol, T.Nvec);
        T.directGalerkinVUV(links->data, L, 0, true, true, (int)a0, na);
      }
      HIP_CHECK(hipStreamSynchronize(computeStream()));
      poolDeviceFree(L, wbytes);
    } else if (!analyticLocal)
      for (int j = 0; j < n; j++) {   // the dense local matrix of a coarse parent, on top of the completed hop part
        T.column(*phi, j);
        parent->localTerm(*w, *phi);
        T.R(*c, *w);
        hipLaunchKernelGGL(insert_column_kernel, dim3(nins), dim3(bs), 0, computeStream(), links->data, cvecFull(*c), n, 8, j, 1, links->nSites);
      }
  } else
  for (int j = 0; j < n; j++) {
    T.column(*phi, j);   // = P e_j for the unit vector j at every coarse site
    if (four) {
      for (int d = 0; d < 8; d++) parent->hopDir(*w8[d], *phi, d);
      for (int half = 0; half < 2; half++) {
        const int dirs[4] = {4 * half, 4 * half + 1, 4 * half + 2, 4 * half + 3};
        ColorSpinorField *in4[4] = {w8[dirs[0]], w8[dirs[1]], w8[dirs[2]], w8[dirs[3]]};
        T.RSplit4(cl, cs, in4, dirs);
        for (int q = 0; q < 4; q++) {
          hipLaunchKernelGGL(insert_column_kernel, dim3(nins), dim3(bs), 0, computeStream(), links->data, cvecFull(*cl[q]), n, dirs[q], j, 0, links->nSites);
          hipLaunchKernelGGL(insert_column_kernel, dim3(nins), dim3(bs), 0, computeStream(), links->data, cvecFull(*cs[q]), n, 8, j, 1, links->nSites);
        }
      }
    } else {
      for (int d = 0; d < 8; d++) {
        parent->hopDir(*w, *phi, d);
        T.RSplit(*c, *c2, *w, d);   // one pass over V: the part of the hop that leaves the aggregate -> link d, the rest -> local term
        hipLaunchKernelGGL(insert_column_kernel, dim3(nins), dim3(bs), 0, computeStream(), links->data, cvecFull(*c), n, d, j, 0, links->nSites);
        hipLaunchKernelGGL(insert_column_kernel, dim3(nins), dim3(bs), 0, computeStream(), links->data, cvecFull(*c2), n, 8, j, 1, links->nSites);
      }
    }
    parent->localTerm(*w, *phi);
    T.R(*c, *w);
    hipLaunchKernelGGL(insert_column_kernel, dim3(nins), dim3(bs), 0, computeStream(), links->data, cvecFull(*c), n, 8, j, 1, links->nSites);
  }
  HIP_CHECK(hipGetLastError());
  HIP_CHECK(hipStreamSynchronize(computeStream()));
  for (int d = 0; d < 8; d++) delete w8[d];
  for (int q = 0; q < 4; q++) { delete cl[q]; delete cs[q]; }
  delete c; delete c2; delete phi; delete w;
}

// ---- preconditioned links: batched dense inverse and products ----
// one work-group per coarse site: Gauss-Jordan with partial pivoting on [X | 1] held in LDS (n <= 64: 64 x 128 complex fp32 = 64 KiB)
__global__ void __launch_bounds__(256) coarse_invert_kernel(float *hat, const float *G, int n, int *fail) {
  extern __shared__ float2 aug[];  // [n][2n] row-major
  __shared__ int piv;
  __shared__ float2 pinv;
  const size_t A = blockIdx.x;
  const int W = 2 * n;
  const float4 *X4 = reinterpret_cast<const float4 *>(G) + (A * 9 + 8) * (size_t)(n / 2) * n;
  for (int e = threadIdx.x; e < n * (n / 2); e += blockDim.x) {
    const int jp = e / n, i = e - jp * n;
    const float4 w = X4[(size_t)jp * n + i];
    aug[i * W + 2 * jp] = make_float2(w.x, w.y);
    aug[i * W + 2 * jp + 1] = make_float2(w.z, w.w);
  }
  for (int e = threadIdx.x; e < n * n; e += blockDim.x) { const int i = e / n, j = e - i * n; aug[i * W + n + j] = make_float2(i == j ? 1.f : 0.f, 0.f); }
  __syncthreads();
  for (int p = 0; p < n; p++) {
    if (threadIdx.x == 0) {
      int best = p; float bm = -1.f;
      for (int r = p; r < n; r++) { const float2 v = aug[r * W + p]; const float m = v.x * v.x + v.y * v.y; if (m > bm) { bm = m; best = r; } }
      piv = best;
      const float2 v = aug[best * W + p];
      if (bm <= 0.f) { *fail = 1; pinv = make_float2(0.f, 0.f); }
      else pinv = make_float2(v.x / bm, -v.y / bm);
    }
    __syncthreads();
    const int pr = piv;
    if (pr != p) for (int c = threadIdx.x; c < W; c += blockDim.x) { const float2 t = aug[p * W + c]; aug[p * W + c] = aug[pr * W + c]; aug[pr * W + c] = t; }
    __syncthreads();
    const float2 ip = pinv;
    for (int c = threadIdx.x; c < W; c += blockDim.x) { const float2 v = aug[p * W + c]; aug[p * W + c] = make_float2(v.x * ip.x - v.y * ip.y, v.x * ip.y + v.y * ip.x); }
    __syncthreads();
    // eliminate column p from every other row: thread -> (row r, column c)
    for (int e = threadIdx.x; e < n * W; e += blockDim.x) {
      const int r = e / W, c = e - r * W;
      if (r == p || c == p) continue;
      const float2 f = aug[r * W + p], v = aug[p * W + c];
      float2 &t = aug[r * W + c];
      t.x -= f.x * v.x - f.y * v.y; t.y -= f.x * v.y + f.y * v.x;
    }
    __syncthreads();
    for (int r = threadIdx.x; r < n; r += blockDim.x) if (r != p) aug[r * W + p] = make_float2(0.f, 0.f);
    __syncthreads();
  }
  float4 *O4 = reinterpret_cast<float4 *>(hat) + (A * 9 + 8) * (size_t)(n / 2) * n;
  for (int e = threadIdx.x; e < n * (n / 2); e += blockDim.x) {
    const int jp = e / n, i = e - jp * n;
    const float2 a = aug[i * W + n + 2 * jp], b = aug[i * W + n + 2 * jp + 1];
    O4[(size_t)jp * n + i] = make_float4(a.x, a.y, b.x, b.y);
  }
}

// hat[d] = Xinv * H_d for d = 0..7 (one work-group per (site, d))
__global__ void __launch_bounds__(256) coarse_hat_kernel(float *hat, const float *G, int n) {
  extern __shared__ float2 sm[];  // Xinv [n][n] then H [n][n], row-major
  float2 *Xi = sm, *H = sm + n * n;
  const size_t A = blockIdx.x >> 3;
  const int d = blockIdx.x & 7;
  const float4 *X4 = reinterpret_cast<const float4 *>(hat) + (A * 9 + 8) * (size_t)(n / 2) * n;
  const float4 *H4 = reinterpret_cast<const float4 *>(G) + (A * 9 + d) * (size_t)(n / 2) * n;
  for (int e = threadIdx.x; e < n * (n / 2); e += blockDim.x) {
    const int jp = e / n, i = e - jp * n;
    const float4 x = X4[(size_t)jp * n + i], h = H4[(size_t)jp * n + i];
    Xi[i * n + 2 * jp] = make_float2(x.x, x.y); Xi[i * n + 2 * jp + 1] = make_float2(x.z, x.w);
    H[i * n + 2 * jp] = make_float2(h.x, h.y); H[i * n + 2 * jp + 1] = make_float2(h.z, h.w);
  }
  __syncthreads();
  float4 *O4 = reinterpret_cast<float4 *>(hat) + (A * 9 + d) * (size_t)(n / 2) * n;
  for (int e = threadIdx.x; e < n * (n / 2); e += blockDim.x) {
    const int jp = e / n, i = e - jp * n;
    float2 a = make_float2(0.f, 0.f), b = make_float2(0.f, 0.f);
    for (int k = 0; k < n; k++) {
      const float2 x = Xi[i * n + k], h0 = H[k * n + 2 * jp], h1 = H[k * n + 2 * jp + 1];
      a.x += x.x * h0.x - x.y * h0.y; a.y += x.x * h0.y + x.y * h0.x;
      b.x += x.x * h1.x - x.y * h1.y; b.y += x.x * h1.y + x.y * h1.x;
    }
    O4[(size_t)jp * n + i] = make_float4(a.x, a.y, b.x, b.y);
  }
}

const CoarseGauge &DiracCoarse::HatLinks() const {
  if (hat) return *hat;
  const int n = links->n;
  hat = new CoarseGauge(links->Xc, n);
  ownHat = true;
  int *d_fail = nullptr, h_fail = 0;
  HIP_CHECK(qaMalloc((void **)&d_fail, sizeof(int)));
  HIP_CHECK(hipMemsetAsync(d_fail, 0, sizeof(int), computeStream()));
  HIP_CHECK(hipFuncSetAttribute((const void *)coarse_invert_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
  HIP_CHECK(hipFuncSetAttribute((const void *)coarse_hat_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
  hipLaunchKernelGGL(coarse_invert_kernel, dim3(links->nSites), dim3(256), (size_t)n * 2 * n * sizeof(float2), computeStream(), hat->data, links->data, n, d_fail);
  hipLaunchKernelGGL(coarse_hat_kernel, dim3(links->nSites * 8), dim3(256), (size_t)2 * n * n * sizeof(float2), computeStream(), hat->data, links->data, n);
  HIP_CHECK(hipGetLastError());
  HIP_CHECK(hipMemcpyAsync(&h_fail, d_fail, sizeof(int), hipMemcpyDeviceToHost, computeStream()));
  HIP_CHECK(hipStreamSynchronize(computeStream()));
  HIP_CHECK(hipFree(d_fail));
  if (h_fail) errorQuda("singular coarse local matrix X: cannot build the preconditioned coarse operator");
  return *hat;
}

DiracCoarse::DiracCoarse(const DiracParam &p) : Dirac(p), transfer(p.transfer), parent(p.dirac), links(nullptr), ownLinks(false), fineFlavor(p.twistFlavor), hat(nullptr), ownHat(false) {
  if (!transfer || !parent) errorQuda("coarse operator needs a transfer operator and a parent operator");
  Nc = transfer->Nvec;
  type = QUDA_COARSE_DIRAC;
  build();
}
DiracCoarse::DiracCoarse(const DiracCoarse &o, const DiracParam &p)
    : Dirac(p), transfer(o.transfer), parent(o.parent), links(o.links), ownLinks(false), Nc(o.Nc), fineFlavor(o.fineFlavor), hat(nullptr), ownHat(false) {
  type = QUDA_COARSE_DIRAC;
  hat = const_cast<CoarseGauge *>(&o.HatLinks());  // shared with (and owned by) the operator the links came from
}
DiracCoarse::~DiracCoarse() { if (ownLinks) delete links; if (ownHat) delete hat; }

void DiracCoarse::CloverInv(ColorSpinorField &out, const ColorSpinorField &in, const QudaParity parity) const {
  applyCoarse(out, in, HatLinks(), 1 << 8, in.SiteSubset() == QUDA_FULL_SITE_SUBSET ? -1 : (int)parity);
  const long n = 2 * Nc;
  flops += (8 * n * n - 2 * n) * (unsigned long long)out.Volume();
}

// ---- even-odd preconditioned coarse operator: M = 1 - (A^-1 D)_{p pbar} (A^-1 D)_{pbar p}  (reference :237-372) ----
DiracCoarsePC::DiracCoarsePC(const DiracCoarse &o, const DiracParam &p) : DiracCoarse(o, p) { type = QUDA_COARSEPC_DIRAC; }

void DiracCoarsePC::Dslash(ColorSpinorField &out, const ColorSpinorField &in, const QudaParity parity) const {
  applyCoarse(out, in, HatLinks(), 0xff, (int)parity);
  const long n = 2 * Nc;
  flops += (8 * (8 * n * n) - 2 * n) * (unsigned long long)out.Volume();
}
void DiracCoarsePC::DslashXpay(ColorSpinorField &out, const ColorSpinorField &in, const QudaParity parity, const ColorSpinorField &x, const double &k) const {
  Dslash(out, in, parity);
  blas::xpay(x, k, out);
}
void DiracCoarsePC::M(ColorSpinorField &out, const ColorSpinorField &in) const {
  if (in.SiteSubset() == QUDA_FULL_SITE_SUBSET || out.SiteSubset() == QUDA_FULL_SITE_SUBSET) errorQuda("Cannot apply preconditioned operator to full field");
  if (dagger != QUDA_DAG_NO) errorQuda("Dagger operator not implemented");
  ColorSpinorField *t = getTmp(tmp1, own1, in);
  if (matpcType == QUDA_MATPC_EVEN_EVEN) {
    Dslash(*t, in, QUDA_ODD_PARITY);
    DslashXpay(out, *t, QUDA_EVEN_PARITY, in, -1.0);
  } else if (matpcType == QUDA_MATPC_ODD_ODD) {
    Dslash(*t, in, QUDA_EVEN_PARITY);
    DslashXpay(out, *t, QUDA_ODD_PARITY, in, -1.0);
  } else {
    errorQuda("matpcType %d: the coarse preconditioned operator is built for the symmetric types", matpcType);
  }
}
void DiracCoarsePC::prepare(ColorSpinorField *&src, ColorSpinorField *&sol, ColorSpinorField &x, ColorSpinorField &b, const QudaSolutionType solType) const {
  if (solType == QUDA_MATPC_SOLUTION || solType == QUDA_MATPCDAG_MATPC_SOLUTION) { src = &b; sol = &x; return; }
  ColorSpinorField *t = getTmp(tmp1, own1, b.Even());
  if (matpcType == QUDA_MATPC_EVEN_EVEN) {  // src = A_ee^-1 (b_e - D_eo A_oo^-1 b_o)
    src = &(x.Odd());
    CloverInv(*src, b.Odd(), QUDA_ODD_PARITY);
    DiracCoarse::Dslash(*t, *src, QUDA_EVEN_PARITY);
    blas::xpay(b.Even(), -1.0, *t);
    CloverInv(*src, *t, QUDA_EVEN_PARITY);
    sol = &(x.Even());
  } else if (matpcType == QUDA_MATPC_ODD_ODD) {
    src = &(x.Even());
    CloverInv(*src, b.Even(), QUDA_EVEN_PARITY);
    DiracCoarse::Dslash(*t, *src, QUDA_ODD_PARITY);
    blas::xpay(b.Odd(), -1.0, *t);
    CloverInv(*src, *t, QUDA_ODD_PARITY);
    sol = &(x.Odd());
  } else {
    errorQuda("MatPCType %d not valid for DiracCoarsePC", matpcType);
  }
}
void DiracCoarsePC::reconstruct(ColorSpinorField &x, const ColorSpinorField &b, const QudaSolutionType solType) const {
  if (solType == QUDA_MATPC_SOLUTION || solType == QUDA_MATPCDAG_MATPC_SOLUTION) return;
  checkFullSpinor(x, b);
  ColorSpinorField *t = getTmp(tmp1, own1, b.Even());
  if (matpcType == QUDA_MATPC_EVEN_EVEN) {  // x_o = A_oo^-1 (b_o - D_oe x_e)
    DiracCoarse::Dslash(*t, x.Even(), QUDA_ODD_PARITY);
    blas::xpay(b.Odd(), -1.0, *t);
    CloverInv(x.Odd(), *t, QUDA_ODD_PARITY);
  } else if (matpcType == QUDA_MATPC_ODD_ODD) {
    DiracCoarse::Dslash(*t, x.Odd(), QUDA_EVEN_PARITY);
    blas::xpay(b.Even(), -1.0, *t);
    CloverInv(x.Even(), *t, QUDA_EVEN_PARITY);
  } else {
    errorQuda("MatPCType %d not valid for DiracCoarsePC", matpcType);
  }
}

void DiracCoarse::M(ColorSpinorField &out, const ColorSpinorField &in) const {
  if (dagger == QUDA_DAG_YES) errorQuda("coarse dagger operator not implemented (as in the reference)");
  applyCoarse(out, in, *links, 0x1ff, -1);
  const long n = 2 * Nc;
  flops += (9 * (8 * n * n) - 2 * n) * (unsigned long long)in.Volume();
}
void DiracCoarse::Dslash(ColorSpinorField &out, const ColorSpinorField &in, const QudaParity parity) const {
  applyCoarse(out, in, *links, 0xff, in.SiteSubset() == QUDA_FULL_SITE_SUBSET ? -1 : (int)parity);
  const long n = 2 * Nc;
  flops += (8 * (8 * n * n) - 2 * n) * (unsigned long long)out.Volume();
}
void DiracCoarse::Clover(ColorSpinorField &out, const ColorSpinorField &in, const QudaParity parity) const {
  applyCoarse(out, in, *links, 1 << 8, in.SiteSubset() == QUDA_FULL_SITE_SUBSET ? -1 : (int)parity);
  const long n = 2 * Nc;
  flops += (8 * n * n - 2 * n) * (unsigned long long)out.Volume();
}
void DiracCoarse::DslashXpay(ColorSpinorField &out, const ColorSpinorField &in, const QudaParity parity, const ColorSpinorField &x, const double &k) const {
  Dslash(out, in, parity);
  blas::xpay(x, k, out);
}
void DiracCoarse::MdagM(ColorSpinorField &, const ColorSpinorField &) const { errorQuda("Not implemented"); }
void DiracCoarse::prepare(ColorSpinorField *&src, ColorSpinorField *&sol, ColorSpinorField &x, ColorSpinorField &b, const QudaSolutionType solType) const {
  if (solType == QUDA_MATPC_SOLUTION || solType == QUDA_MATPCDAG_MATPC_SOLUTION) errorQuda("Preconditioned solution requires a preconditioned solve_type");
  src = &b;
  sol = &x;
}
void DiracCoarse::reconstruct(ColorSpinorField &, const ColorSpinorField &, const QudaSolutionType) const {}
void DiracCoarse::hopDir(ColorSpinorField &out, const ColorSpinorField &in, int dir) const { applyCoarse(out, in, *links, 1 << dir, -1); }
void DiracCoarse::localTerm(ColorSpinorField &out, const ColorSpinorField &in) const { applyCoarse(out, in, *links, 1 << 8, -1); }

}  // namespace quda
