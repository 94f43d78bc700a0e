// coarse_cycle.hip — the multigrid V-cycle below a coarse level as ONE persistent kernel (see coarse_cycle.h).
//
// Structure of the launch.  G work-groups of 256 threads, all resident (G <= occupancy x CUs, checked on the host).  The cycle is a fixed
// sequence of PHASES; inside a phase every work-group walks its share of the phase's tasks (a task = one coarse site, one aggregate, or a
// chunk of vector elements) and between two phases all work-groups meet at a device-wide barrier.  The barrier is arrival slots (one
// 64-byte line per work-group) gathered by work-group 0, which then raises a release word: 1.9 us for 512 work-groups, where one atomic
// counter bracketed by agent-scope release / acquire fences took 30 us (tools/ubench_grid_barrier.hip, profiles/r04_grid_barrier_variants.log:
// the fences write back and invalidate the L2 of every XCD, and 512 pollers fight the 512 adds for one address).  There are NO cache
// fences: every datum one work-group writes and another reads — work vectors, partial sums — moves through agent-scope (sc1) loads and
// stores, which are coherent across the XCDs' L2s by themselves; a work-group's stores are acknowledged (s_waitcnt vmcnt(0), part of the
// work-group barrier) before its arrival slot is written.  A site task is the dense product of coarse_apply_kernel (coarse.hip; reference lib/dslash_coarse.cu:50-203):
// the 9 input vectors of the site staged in LDS, the (matrix, column pair) range dealt to the four waves, rows on lanes, 16-byte link loads.
// Sums (MR: (Ar, r), |Ar|^2; GCR: all (Ap_i, Ap_k), (Ap_k, r), |Ap_k|^2 of an iteration, |r|^2) are accumulated per work-group in fp64,
// written to a double-buffered partial array, and after the barrier EVERY work-group adds the partials in the same fixed order — so all of
// them (and, after the rank exchange, all ranks) hold bit-identical scalars and take identical control-flow decisions without a broadcast.
// Grid-decomposed lattices: a phase that hops reads faces of its input vector from the neighbour rank; the work-groups first PUSH the face
// sites they own into the neighbours' peer-mapped windows as flag-in-data 16-byte words {re, flag, im, flag} (flag = exchange number; the
// protocol of the fine stencil's halo, dslash.hip GhostLL) and the site tasks that hop across the face poll exactly the words they need.
// Global sums travel the same way ({lo, flag, hi, flag} per double into a slot per rank on every rank, added in rank order).  Zones are
// double-buffered by the parity of the exchange number: a rank can be at most one exchange ahead of a neighbour, because it needs that
// neighbour's face of exchange k to finish exchange k and there is a device-wide barrier between two exchanges of one rank.
#include "coarse_cycle.h"

#include <cmath>
#include <cstring>

#include "coarse.h"
#include "device_io.h"
#include "p2p.h"

namespace quda {

bool commReductionsNeeded();
int commNeighborRank(int dim, int dir);
void commBarrier();

namespace {

constexpr int kMaxLevels = QUDA_MAX_MG_LEVEL - 1;
constexpr int kKrylovMax = 20;
constexpr int kRedMax = 2 * kKrylovMax + 4;   // sums of one grid reduction
constexpr int kMaxRanks = 16;
constexpr int kThreads = 256;
// Halo zones are buffered FOUR deep (exchange number mod 4).  A face is pushed either by the phase that reads it (after the barrier in front of
// that phase) or — matpc's intermediate vector, the residual of MR / GCR — already by the phase that PRODUCES it, so that it is on its way
// while the barrier is still being crossed (3.5 us per hop phase).  With producer-side pushes a rank in the phase that consumes exchange k + 1
// may push exchange k + 2 while a neighbour is still reading exchange k - 1: live exchanges span at most 4 numbers.
constexpr int kHaloBufs = 4;

struct CcVec { float2 *p[2]; int stride; };

struct CcLevel {
  int Xc[4];
  int Vh, n;
  const float4 *links, *hat;   // [site][9][n/2][n] float4: Y (slot 8 = X) and Xinv Y (slot 8 = Xinv)
  int solvePar;                // parity of the even-odd preconditioned system (reference matpc_type of DiracCoarsePC)
  int nuPre, nuPost, mrGlobal;
  int ntLinks;                 // links streamed with non-temporal loads (levels whose links exceed what the L2s hold between two phases)
  float omega;
  // transfer to the next coarser level (absent on the coarsest one)
  const float4 *V; const int *b2f; int blockVol, GS, nAgg;
  CcVec b, x, rf;              // full fields
  int bExternal;               // b is the caller's field (top level)
  float2 *bt, *r, *Ar, *t;     // parity fields, [component][Vh]: bt, r, Ar on solvePar, t on the other parity
  // halo of partitioned dimensions
  int commMask, faceCB[4];
  u32x4_t *ghost[4][2][kHaloBufs];   // [dim][0: from the -dim neighbour (its x = L-1 face), 1: from the +dim neighbour (its x = 0 face)][buffer]
  u32x4_t *peer[4][2][kHaloBufs];    // [dim][my face 0 / L-1][buffer]: where that face lands in the neighbour's window
};

struct CcArg {
  int nl;
  CcLevel L[kMaxLevels];
  // coarsest-grid GCR (reference lib/inv_gcr_quda.cpp:235-516, K = none, one precision)
  float2 *P, *AP, *y;          // [k][component][Vh] search directions, their images; y: accumulated solution
  int nKrylov, maxiter, maxResInc, maxResIncTotal;
  double tol, delta;
  // synchronisation and sums
  unsigned *bar;               // [0] release word, [16 (1 + w)] arrival slot of work-group w
  double *partial;             // [2 buffers][G][kRedMax] (inside the slab)
  char *slab; unsigned slabBytes;   // everything one work-group writes and another reads
  CcVec xOut;                  // the caller's solution field: copied out of the slab at the end
  unsigned long long *timeline;   // optional: wall clock at every barrier (work-group 0)
  unsigned *state;             // [0] exchange number of the halo windows, [1] of the sum windows, [2..6] statistics of the launch, [7] barrier epoch
  int world, rank;
  u32x4_t *redOwn;             // [2 buffers][world][kRedMax]
  u32x4_t *redPeer[kMaxRanks]; // the same window on every rank
  int *errWord;
  unsigned long long waitTicks;
  int resFromSmoother;   // the full residual a level restricts comes from its pre-smoother's residual (cc_residual_local), not from the operator
};

struct CcShared {
  float2 xin[9][64];
  float2 part[4][64];
  float2 yout[64];
  double dacc[kRedMax];
  double red[kRedMax];
  double betaRe[kKrylovMax][kKrylovMax], betaIm[kKrylovMax][kKrylovMax];
  double alRe[kKrylovMax], alIm[kKrylovMax], gam[kKrylovMax], dlRe[kKrylovMax], dlIm[kKrylovMax];
  double wsum[4];
  int nbPar[9], nbIdx[9], nbZone[9];
  int abort;
};
// dynamic LDS: restrictor — the fine vectors of one aggregate [site in aggregate][component]; prolongator — the coarse vector
extern __shared__ float2 cc_agg[];

struct CcCtx {
  unsigned epoch;     // barriers passed (counted across launches)
  unsigned epoch0;    // its value when this launch began
  unsigned seq;       // halo exchanges done (flag of the last one)
  unsigned rseq;      // global sums done
  unsigned nred;      // grid reductions done (partial buffer = nred & 1)
  bool dead;          // an error was recorded: no more waiting
};

// links and V are read once per phase: non-temporal 16-byte loads, so they do not displace the work vectors from the L2
// Pointers come out of the argument block as generic pointers, and a load through a generic pointer is a FLAT load — which the compiler
// can only wait for with vmcnt(0), i.e. behind every link load in flight.  Everything outside LDS is therefore accessed through an
// explicitly GLOBAL pointer.
__device__ __forceinline__ float4 cc_ld(const float4 *p) {   // cached global load (links of a level small enough to stay in the L2s)
  typedef float f32x4_g __attribute__((ext_vector_type(4)));
  typedef const __attribute__((address_space(1))) f32x4_g *gptr;
  const f32x4_g t = *(gptr)(unsigned long long)p;
  return make_float4(t.x, t.y, t.z, t.w);
}
template <typename T> __device__ __forceinline__ __attribute__((address_space(1))) T *gp(T *p) { return (__attribute__((address_space(1))) T *)(unsigned long long)p; }
__device__ __forceinline__ float4 cc_ld_nt(const float4 *p) {
  typedef float f32x4_nt __attribute__((ext_vector_type(4)));
  typedef const __attribute__((address_space(1))) f32x4_nt *gptr;
  const f32x4_nt t = __builtin_nontemporal_load((gptr)(unsigned long long)p);
  return make_float4(t.x, t.y, t.z, t.w);
}
// Data written by one work-group and read by another inside the launch lives in ONE slab (a.slab: work vectors, partial sums) and moves through
// raw buffer loads / stores with the sc0 sc1 cache policy: coherent across the XCDs' L2s by themselves, and — unlike atomic loads — ordinary
// memory operations to the compiler, which issues the loads of a loop back to back instead of waiting for each one (a phase is a chain of
// ~1 us round trips; with relaxed atomic accesses it was 12 us long).
#ifndef QA_CC_AUX
#define QA_CC_AUX 16
#endif
constexpr int kCoherent = QA_CC_AUX;   // 16: sc1 (agent scope), 17: sc0 sc1 (system scope)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t cc_rsrc(const CcArg &a) { return __builtin_amdgcn_make_buffer_rsrc(a.slab, 0, (int)a.slabBytes, 0x00020000); }
__device__ __forceinline__ unsigned cc_off(const CcArg &a, const void *p) { return (unsigned)((const char *)p - (const char *)a.slab); }
__device__ __forceinline__ float2 ldc(const CcArg &a, const float2 *p) {
  return __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(cc_rsrc(a), cc_off(a, p), 0, kCoherent));
}
__device__ __forceinline__ void stc(const CcArg &a, float2 *p, float2 v) {
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, v), cc_rsrc(a), cc_off(a, p), 0, kCoherent);
}
__device__ __forceinline__ double ldc(const CcArg &a, const double *p) {
  return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(cc_rsrc(a), cc_off(a, p), 0, kCoherent));
}
__device__ __forceinline__ void stc(const CcArg &a, double *p, double v) {
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, v), cc_rsrc(a), cc_off(a, p), 0, kCoherent);
}
// the source of a level: the top level's is the caller's field (outside the slab, never written during the launch: plain load)
__device__ __forceinline__ float2 cc_ldb(const CcArg &a, const CcLevel &L, const float2 *p) { return L.bExternal ? __builtin_bit_cast(float2, *gp(reinterpret_cast<const unsigned long long *>(p))) : ldc(a, p); }

// An error (a wait that ran out, a GCR breakdown) is recorded in the host-visible error word AND raised in the abort word that sits next to
// the barrier's release word: the barrier's poll reads both with one 8-byte load, so noticing a failure costs no extra round trip.
__device__ __forceinline__ bool cc_fail(const CcArg &a, int code, int i1, int i2, int i3, int i4, int i5, int i6, int i7) {
  const bool first = atomicCAS(a.errWord, 0, code) == 0;
  if (first) { a.errWord[1] = i1; a.errWord[2] = i2; a.errWord[3] = i3; a.errWord[4] = i4; a.errWord[5] = i5; a.errWord[6] = i6; a.errWord[7] = i7; }
  __hip_atomic_store(gp(a.bar + 1), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return first;
}
__device__ __forceinline__ bool cc_aborted(const CcArg &a) { return __hip_atomic_load(gp(a.bar + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0; }
// Polling loops look at the clock and at the abort word only every 64th turn: s_memrealtime and a second memory round trip per turn made a
// turn ~6 us long — the granularity with which a 2 us barrier or a 5 us halo arrival was noticed (hop phases took 21 us, local ones 6-8).
#define QA_CC_POLL_CHECK(it, t0, timedOut, aborted)                                   \
  if ((++(it) & 63u) == 0) {                                                           \
    if (cc_aborted(a)) { aborted = true; }                                             \
    else { const unsigned long long tn_ = wall_clock64(); if (!(t0)) (t0) = tn_; else if (tn_ - (t0) > a.waitTicks) timedOut = true; } \
  }

// device-wide barrier: every work-group's (coherent) stores of the phase are visible to every work-group's (coherent) loads after it.
// a.bar: [0] release word, [1] abort word, [16 (1 + w)] arrival slot of work-group w; the epoch keeps counting across launches, so nothing
// is ever reset.
__device__ __forceinline__ void cc_barrier(const CcArg &a, CcShared &s, CcCtx &c) {
  __syncthreads();   // all waves' stores issued AND acknowledged (workgroup-scope release = s_waitcnt vmcnt(0) in front of s_barrier)
  c.epoch++;
  const unsigned epoch = c.epoch;
  if (threadIdx.x == 0) s.abort = 0;
  if (!c.dead) {
    if (threadIdx.x == 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_store(gp(a.bar + 16 * (1 + blockIdx.x)), epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (blockIdx.x == 0) {
      for (int w = threadIdx.x; w < (int)gridDim.x; w += kThreads) {
        unsigned long long t0 = 0; unsigned it = 0; bool timedOut = false, aborted = false;
        while ((int)(__hip_atomic_load(gp(a.bar + 16 * (1 + w)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - epoch) < 0) {
          QA_CC_POLL_CHECK(it, t0, timedOut, aborted)
          if (timedOut) cc_fail(a, 40, w, (int)epoch, (int)__hip_atomic_load(gp(a.bar + 16 * (1 + w)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), 0, (int)(epoch - c.epoch0), 0, 0);
          if (timedOut || aborted) break;
          __builtin_amdgcn_s_sleep(1);
        }
      }
      __syncthreads();
      if (threadIdx.x == 0) {
        __hip_atomic_store(gp(a.bar), epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s.abort = cc_aborted(a) ? 1 : 0;   // (after the release: the others do not wait for this load)
      }
    } else if (threadIdx.x == 0) {
      unsigned long long t0 = 0; unsigned it = 0; bool timedOut = false, aborted = false;
      const unsigned long long *rel = reinterpret_cast<const unsigned long long *>(a.bar);
      for (;;) {
        const unsigned long long v = __hip_atomic_load(gp(rel), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // {release, abort}
        if ((unsigned)(v >> 32)) { aborted = true; break; }
        if ((int)((unsigned)v - epoch) >= 0) break;
        QA_CC_POLL_CHECK(it, t0, timedOut, aborted)
        if (timedOut) cc_fail(a, 40, -1, (int)epoch, (int)(unsigned)v, 0, (int)(epoch - c.epoch0), 0, 0);
        if (timedOut || aborted) break;
        __builtin_amdgcn_s_sleep(1);
      }
      if (timedOut || aborted) s.abort = 1;
    }
  }
  if (a.timeline && blockIdx.x == 0 && threadIdx.x == 0 && epoch - c.epoch0 < 1024) a.timeline[epoch - c.epoch0] = wall_clock64();
  __syncthreads();
  if (s.abort) c.dead = true;   // the same decision in every thread of the work-group
}

// ---- sums ----
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}
__device__ __forceinline__ void cc_clear_acc(CcShared &s, int K) {
  if (threadIdx.x < K) s.dacc[threadIdx.x] = 0.0;
  __syncthreads();
}
// s.dacc[0..K) of every work-group -> s.red[0..K) (identical everywhere; all ranks when `global`).  Contains one device-wide barrier.
__device__ __forceinline__ void cc_reduce(const CcArg &a, CcShared &s, CcCtx &c, int K, bool global) {
  __syncthreads();
  double *part = a.partial + (size_t)(c.nred & 1) * gridDim.x * kRedMax;
  c.nred++;
  if (threadIdx.x < K) stc(a, part + (size_t)blockIdx.x * kRedMax + threadIdx.x, s.dacc[threadIdx.x]);
  cc_barrier(a, s, c);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int k = wave; k < K; k += 4) {
    double v = 0.0;
    for (int w0 = 0; w0 < (int)gridDim.x; w0 += 512) {   // 8 partial sums per lane requested back to back
      double t[8];
#pragma unroll
      for (int i = 0; i < 8; i++) { const int w = w0 + lane + 64 * i; t[i] = ldc(a, part + (size_t)(w < (int)gridDim.x ? w : 0) * kRedMax + k); }
#pragma unroll
      for (int i = 0; i < 8; i++) v += (w0 + lane + 64 * i < (int)gridDim.x) ? t[i] : 0.0;
    }
    v = wave_sum(v);
    if (lane == 0) s.red[k] = v;
  }
  __syncthreads();
  if (global && a.world > 1) {
    c.rseq++;
    const unsigned flag = c.rseq;
    const int buf = (int)(c.rseq & 1);
    if (blockIdx.x == 0) {
      for (int e = threadIdx.x; e < a.world * K; e += kThreads) {
        const int r = e / K, k = e - r * K;
        const unsigned long long bits = __builtin_bit_cast(unsigned long long, s.red[k]);
        unsigned long long *dst = reinterpret_cast<unsigned long long *>(a.redPeer[r] + ((size_t)buf * a.world + a.rank) * kRedMax + k);
        __hip_atomic_store(gp(dst), (bits & 0xffffffffull) | ((unsigned long long)flag << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(gp(dst + 1), (bits >> 32) | ((unsigned long long)flag << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
    __syncthreads();
    // every work-group reads all ranks' slots of its own window and adds them in rank order
    if (threadIdx.x < K) {
      double v = 0.0;
      for (int r = 0; r < a.world; r++) {
        const unsigned long long *src = reinterpret_cast<const unsigned long long *>(a.redOwn + ((size_t)buf * a.world + r) * kRedMax + threadIdx.x);
        unsigned long long lo, hi, t0 = 0;
        unsigned it = 0; bool timedOut = false, aborted = false;
        for (;;) {
          lo = __hip_atomic_load(gp(src), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          hi = __hip_atomic_load(gp(src + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          if ((unsigned)(lo >> 32) == flag && (unsigned)(hi >> 32) == flag) break;
          if (c.dead) break;
          QA_CC_POLL_CHECK(it, t0, timedOut, aborted)
          if (timedOut) cc_fail(a, 41, r, (int)flag, (int)(lo >> 32), (int)(hi >> 32), (int)c.rseq, buf, (int)threadIdx.x);
          if (timedOut || aborted) break;
          __builtin_amdgcn_s_sleep(1);
        }
        v += __builtin_bit_cast(double, (lo & 0xffffffffull) | ((hi & 0xffffffffull) << 32));
      }
      s.red[threadIdx.x] = v;
    }
    if (__syncthreads_or(!c.dead && cc_aborted(a))) c.dead = true;   // (multi-rank sums only: one more look at the abort word)
  }
}

// ---- geometry ----
__device__ __forceinline__ void cc_coords(const CcLevel &L, int par, int xcb, int c[4]) {
  const int Xh = L.Xc[0] >> 1;
  int l = xcb;
  const int xh = l % Xh; l /= Xh;
  const int y = l % L.Xc[1]; l /= L.Xc[1];
  const int z = l % L.Xc[2]; const int t = l / L.Xc[2];
  c[0] = 2 * xh + ((y + z + t + par) & 1); c[1] = y; c[2] = z; c[3] = t;
}

// push the faces of `v` (parities in pmask) of every partitioned dimension into the neighbours' windows; flag = c.seq (already advanced)
__device__ __forceinline__ void cc_push(const CcArg &a, const CcLevel &L, const CcCtx &c, const CcVec &v, int pmask) {
  const int buf = (int)(c.seq & (kHaloBufs - 1));
  const unsigned flag = c.seq;
  const int npar = (pmask == 3) ? 2 : 1, p0 = (pmask == 2) ? 1 : 0;
#pragma unroll
  for (int d = 0; d < 4; d++) {
    if (!((L.commMask >> d) & 1)) continue;
    const int nf = L.faceCB[d];
    const int items = 2 * npar * L.n * nf;
    int Lx[3], o[3], k3 = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) if (k != d) { Lx[k3] = L.Xc[k]; o[k3] = k; k3++; }
    for (int it = blockIdx.x * kThreads + threadIdx.x; it < items; it += gridDim.x * kThreads) {
      int e = it;
      const int f = e % nf; e /= nf;
      const int j = e % L.n; e /= L.n;
      const int q = e % npar; const int face = e / npar;
      const int par = p0 + q;
      int cc[4];
      int l = 2 * f;
      const int c0 = l % Lx[0]; l /= Lx[0];
      const int c1 = l % Lx[1]; const int c2 = l / Lx[1];
      cc[d] = face ? L.Xc[d] - 1 : 0;
      cc[o[0]] = c0; cc[o[1]] = c1; cc[o[2]] = c2;
      cc[o[0]] += (par + cc[0] + cc[1] + cc[2] + cc[3]) & 1;
      const int idx = (((cc[3] * L.Xc[2] + cc[2]) * L.Xc[1] + cc[1]) * L.Xc[0] + cc[0]) >> 1;
      const float2 val = ldc(a, v.p[par] + (size_t)j * v.stride + idx);
      unsigned long long *dst = reinterpret_cast<unsigned long long *>(L.peer[d][face][buf] + ((size_t)par * L.n + j) * nf + f);
      __hip_atomic_store(gp(dst), (unsigned long long)__builtin_bit_cast(unsigned, val.x) | ((unsigned long long)flag << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(gp(dst + 1), (unsigned long long)__builtin_bit_cast(unsigned, val.y) | ((unsigned long long)flag << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// neighbour of the site with coordinates c (parity par, checkerboard index xcb) for matrix m (m = 8: the site itself): its parity and index;
// zone >= 0: the hop crosses a partitioned face, idx = the face site in ghost zone (dimension zone / 2, side zone & 1)
__device__ __forceinline__ void cc_neighbour(const CcLevel &L, const int c[4], int par, int xcb, int m, int &npar, int &idx, int &zone) {
  npar = par; idx = xcb; zone = -1;
  if (m < 8) {
    const int mu = m >> 1;
    int cn[4], Lm = 0, cm = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) { cn[k] = c[k]; if (k == mu) { Lm = L.Xc[k]; cm = c[k]; } }   // (no run-time indexed local arrays: they would live in scratch)
    const int nm = (m & 1) ? (cm == 0 ? Lm - 1 : cm - 1) : (cm == Lm - 1 ? 0 : cm + 1);
#pragma unroll
    for (int k = 0; k < 4; k++) if (k == mu) cn[k] = nm;
    npar = (cn[0] + cn[1] + cn[2] + cn[3]) & 1;
    const bool cross = ((L.commMask >> mu) & 1) && ((m & 1) ? cm == 0 : cm == Lm - 1);
    if (cross) {
      int l = 0, mul = 1;
#pragma unroll
      for (int k = 0; k < 4; k++) if (k != mu) { l += cn[k] * mul; mul *= L.Xc[k]; }
      idx = l >> 1;
      zone = mu * 2 + ((m & 1) ? 0 : 1);   // backward hop: zone 0 (the -mu neighbour's L-1 face); forward hop: zone 1
    } else {
      idx = (((cn[3] * L.Xc[2] + cn[2]) * L.Xc[1] + cn[1]) * L.Xc[0] + cn[0]) >> 1;
    }
  }
}

// stage the input vectors of the matrices in mmask of site (par, xcb) into s.xin — neighbours from `in`, across a partitioned face from the
// ghost zone.  Thread (wave w, lane j) owns component j of the matrices w, w + 4, w + 8: it works out their neighbours itself and requests
// all its loads back to back (one round trip for the whole staging), then polls whatever ghost words have not arrived yet.
__device__ __forceinline__ void cc_stage(const CcArg &a, const CcLevel &L, CcShared &s, CcCtx &c, const CcVec &in, int par, int xcb, int mmask) {
  const int buf = (int)(c.seq & (kHaloBufs - 1));
  const unsigned flag = c.seq;
  const int wave = threadIdx.x >> 6, j = threadIdx.x & 63;
  int cs[4];
  cc_coords(L, par, xcb, cs);
  float2 val[3];
  const unsigned long long *gsrc[3];
  unsigned long long lo[3], hi[3];
  bool act[3], ghost[3];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const int m = wave + 4 * k;
    act[k] = m < 9 && j < L.n && ((mmask >> m) & 1);
    ghost[k] = false; gsrc[k] = nullptr; lo[k] = hi[k] = 0; val[k] = make_float2(0.f, 0.f);
    if (act[k]) {
      int npar, idx, zone;
      cc_neighbour(L, cs, par, xcb, m, npar, idx, zone);
      if (zone < 0) {
        val[k] = ldc(a, in.p[npar] + (size_t)j * in.stride + idx);
      } else {
        ghost[k] = true;
        const int mu = zone >> 1;
        gsrc[k] = reinterpret_cast<const unsigned long long *>(L.ghost[mu][zone & 1][buf] + ((size_t)npar * L.n + j) * L.faceCB[mu] + idx);
        lo[k] = __hip_atomic_load(gp(gsrc[k]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        hi[k] = __hip_atomic_load(gp(gsrc[k] + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 3; k++) {
    if (act[k] && ghost[k]) {
      unsigned long long t0 = 0;
      unsigned it = 0; bool timedOut = false, aborted = false;
      while (!((unsigned)(lo[k] >> 32) == flag && (unsigned)(hi[k] >> 32) == flag)) {
        if (c.dead) break;
        QA_CC_POLL_CHECK(it, t0, timedOut, aborted)
        if (timedOut) cc_fail(a, 42 + wave + 4 * k, xcb, (int)flag, (int)(lo[k] >> 32), (int)(hi[k] >> 32), (int)c.seq, buf, j);
        if (timedOut || aborted) break;
        __builtin_amdgcn_s_sleep(1);
        lo[k] = __hip_atomic_load(gp(gsrc[k]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        hi[k] = __hip_atomic_load(gp(gsrc[k] + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      val[k] = make_float2(__builtin_bit_cast(float, (unsigned)lo[k]), __builtin_bit_cast(float, (unsigned)hi[k]));
    }
  }
#pragma unroll
  for (int k = 0; k < 3; k++)
    if (act[k]) s.xin[wave + 4 * k][j] = val[k];
  __syncthreads();
}

// The dense product of a site task, s.yout[row] = sum_{m in [M0, M1)} G[site][m] s.xin[m], in two steps so that the link loads — which do not
// depend on the input vectors — are in flight while the inputs are being staged (a phase is a chain of latencies, not of bytes: with the
// matrices loaded one after the other behind the staging a hop phase took 21 us, 16 of them eight dependent link round trips):
//   cc_links_issue   requests the links of the first batch of matrices (all of them where they fit ~190 registers) into registers
//   cc_links_finish  multiplies (s.xin must be staged), requests and multiplies the second batch if there is one, adds the four waves' parts
// Every wave takes a quarter of the column pairs of every matrix; rows on lanes; 16-byte non-temporal loads.
template <int N, int M0, int M1> struct LinkRegs {
  static constexpr int NM = M1 - M0, NH = N / 2, CH = (NH + 3) / 4;
  static constexpr int B0 = (NM * CH > 48) ? (NM + 1) / 2 : NM;   // matrices of the first batch
  float4 w[B0][CH];
};
template <int N, int M0, int M1> __device__ __forceinline__ void cc_links_issue(LinkRegs<N, M0, M1> &r, const float4 *G, size_t site, int nt) {
  using R = LinkRegs<N, M0, M1>;
  static_assert(R::NH % 4 == 0, "column pairs must split evenly over the four waves");
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row = lane < N ? lane : 0;   // no branch around the loads (a branch per load made the compiler wait for every first load of a matrix)
#pragma unroll
  for (int i = 0; i < R::B0; i++) {
    const float4 *g = G + ((site * 9 + (M0 + i)) * R::NH + wave * R::CH) * N + row;
#pragma unroll
    for (int q = 0; q < R::CH; q++) r.w[i][q] = nt ? cc_ld_nt(g + (size_t)q * N) : cc_ld(g + (size_t)q * N);
  }
}
template <int N, int M0, int M1> __device__ __forceinline__ void cc_links_finish(CcShared &s, const LinkRegs<N, M0, M1> &r, const float4 *G, size_t site, int nt) {
  using R = LinkRegs<N, M0, M1>;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row = lane < N ? lane : 0;
  float re = 0.f, im = 0.f;
  float4 w2[R::NM - R::B0 > 0 ? R::NM - R::B0 : 1][R::CH];
  if (R::NM > R::B0) {
#pragma unroll
    for (int i = R::B0; i < R::NM; i++) {
      const float4 *g = G + ((site * 9 + (M0 + i)) * R::NH + wave * R::CH) * N + row;
#pragma unroll
      for (int q = 0; q < R::CH; q++) w2[i - R::B0][q] = nt ? cc_ld_nt(g + (size_t)q * N) : cc_ld(g + (size_t)q * N);
    }
  }
#pragma unroll
  for (int i = 0; i < R::NM; i++) {
#pragma unroll
    for (int q = 0; q < R::CH; q++) {
      const int jp = wave * R::CH + q;
      const float4 w = i < R::B0 ? r.w[i < R::B0 ? i : 0][q] : w2[i >= R::B0 ? i - R::B0 : 0][q];
      const float2 x0 = s.xin[M0 + i][2 * jp], x1 = s.xin[M0 + i][2 * jp + 1];
      re += w.x * x0.x - w.y * x0.y + w.z * x1.x - w.w * x1.y;
      im += w.x * x0.y + w.y * x0.x + w.z * x1.y + w.w * x1.x;
    }
  }
  if (lane < N) s.part[wave][lane] = make_float2(re, im);
  __syncthreads();
  if (threadIdx.x < N) {
    const int l = threadIdx.x;
    s.yout[l] = make_float2(s.part[0][l].x + s.part[1][l].x + s.part[2][l].x + s.part[3][l].x, s.part[0][l].y + s.part[1][l].y + s.part[2][l].y + s.part[3][l].y);
  }
  __syncthreads();
}

__device__ __forceinline__ CcVec cc_parity_vec(float2 *p, int par, int Vh) {
  CcVec v; v.p[par] = p; v.p[1 - par] = nullptr; v.stride = Vh; return v;
}
__device__ __forceinline__ CcVec cc_parity_of(const CcVec &full, int par) {
  CcVec v; v.p[par] = full.p[par]; v.p[1 - par] = nullptr; v.stride = full.stride; return v;
}
// producer-side push: component j of the site with coordinates cs (parity par) goes into the neighbours' zones of exchange number `seq`
// wherever the site lies on a partitioned face (the consumer's cc_exchange_begin is then told that the faces are on their way)
__device__ __forceinline__ void cc_push_site(const CcLevel &L, unsigned seq, int par, const int cs[4], int j, float2 val) {
  const int buf = (int)(seq & (kHaloBufs - 1));
#pragma unroll
  for (int d = 0; d < 4; d++) {
    if (!((L.commMask >> d) & 1)) continue;
    const bool lo = cs[d] == 0, hi = cs[d] == L.Xc[d] - 1;
    if (!lo && !hi) continue;
    int l = 0, mul = 1;
#pragma unroll
    for (int k = 0; k < 4; k++) if (k != d) { l += cs[k] * mul; mul *= L.Xc[k]; }
    const int f = l >> 1, nf = L.faceCB[d];
    const unsigned long long w0 = (unsigned long long)__builtin_bit_cast(unsigned, val.x) | ((unsigned long long)seq << 32), w1 = (unsigned long long)__builtin_bit_cast(unsigned, val.y) | ((unsigned long long)seq << 32);
    if (lo) {
      unsigned long long *dst = reinterpret_cast<unsigned long long *>(L.peer[d][0][buf] + ((size_t)par * L.n + j) * nf + f);
      __hip_atomic_store(gp(dst), w0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(gp(dst + 1), w1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (hi) {
      unsigned long long *dst = reinterpret_cast<unsigned long long *>(L.peer[d][1][buf] + ((size_t)par * L.n + j) * nf + f);
      __hip_atomic_store(gp(dst), w0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(gp(dst + 1), w1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}
__device__ __forceinline__ void cc_exchange_begin(const CcArg &a, const CcLevel &L, CcCtx &c, const CcVec &v, int pmask) {
  if (L.commMask) { c.seq++; cc_push(a, L, c, v, pmask); }
}
// all threads: s.dacc[slot] += sum over the work-group of v (waves added in wave order)
__device__ __forceinline__ void cc_block_add(CcShared &s, int slot, double v) {
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) s.wsum[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) s.dacc[slot] += (s.wsum[0] + s.wsum[1]) + (s.wsum[2] + s.wsum[3]);
  __syncthreads();
}

// ---- phases of one level (p = solvePar, q = 1 - p) ----
// Schur prepare of the even-odd preconditioned system (reference DiracCoarsePC::prepare, lib/dirac_coarse.cpp:296-330):
//   x_q = Xinv b_q (scratch) ;  bt_p = Xinv (b_p - D_pq x_q) ;  optionally |bt|^2 into dacc[0].  No barrier at the end.
template <int N> __device__ __forceinline__ void cc_prepare(const CcArg &a, const CcLevel &L, CcShared &s, CcCtx &c, bool norm) {
  const int p = L.solvePar, q = 1 - p;
  for (int t = blockIdx.x; t < L.Vh; t += gridDim.x) {
    LinkRegs<N, 8, 9> lk;
    cc_links_issue(lk, L.hat, (size_t)q * L.Vh + t, L.ntLinks);
    if (threadIdx.x < N) s.xin[8][threadIdx.x] = cc_ldb(a, L, L.b.p[q] + (size_t)threadIdx.x * L.b.stride + t);
    __syncthreads();
    cc_links_finish(s, lk, L.hat, (size_t)q * L.Vh + t, L.ntLinks);
    if (threadIdx.x < N) stc(a, L.x.p[q] + (size_t)threadIdx.x * L.x.stride + t, s.yout[threadIdx.x]);
  }
  cc_barrier(a, s, c);
  cc_exchange_begin(a, L, c, L.x, 1 << q);
  if (norm) cc_clear_acc(s, 1);
  for (int t = blockIdx.x; t < L.Vh; t += gridDim.x) {
    LinkRegs<N, 0, 8> lk;
    LinkRegs<N, 8, 9> lk2;
    cc_links_issue(lk, L.links, (size_t)p * L.Vh + t, L.ntLinks);
    cc_links_issue(lk2, L.hat, (size_t)p * L.Vh + t, L.ntLinks);
    cc_stage(a, L, s, c, L.x, p, t, 0xff);
    cc_links_finish(s, lk, L.links, (size_t)p * L.Vh + t, L.ntLinks);
    if (threadIdx.x < N) {
      const float2 bv = cc_ldb(a, L, L.b.p[p] + (size_t)threadIdx.x * L.b.stride + t), h = s.yout[threadIdx.x];
      s.xin[8][threadIdx.x] = make_float2(bv.x - h.x, bv.y - h.y);
    }
    __syncthreads();
    cc_links_finish(s, lk2, L.hat, (size_t)p * L.Vh + t, L.ntLinks);
    if (threadIdx.x < 64) {
      double v = 0.0;
      if (threadIdx.x < N) {
        const float2 o = s.yout[threadIdx.x];
        stc(a, L.bt + (size_t)threadIdx.x * L.Vh + t, o);
        v = (double)o.x * o.x + (double)o.y * o.y;
      }
      if (norm) { v = wave_sum(v); if (threadIdx.x == 0) s.dacc[0] += v; }
    }
  }
}

// x_q = Xinv (b_q - D_qp x_p)   (reference DiracCoarsePC::reconstruct, lib/dirac_coarse.cpp:352-372).  No barrier at the end.
template <int N> __device__ __forceinline__ void cc_reconstruct(const CcArg &a, const CcLevel &L, CcShared &s, CcCtx &c) {
  const int p = L.solvePar, q = 1 - p;
  cc_exchange_begin(a, L, c, L.x, 1 << p);
  for (int t = blockIdx.x; t < L.Vh; t += gridDim.x) {
    LinkRegs<N, 0, 8> lk;
    LinkRegs<N, 8, 9> lk2;
    cc_links_issue(lk, L.links, (size_t)q * L.Vh + t, L.ntLinks);
    cc_links_issue(lk2, L.hat, (size_t)q * L.Vh + t, L.ntLinks);
    cc_stage(a, L, s, c, L.x, q, t, 0xff);
    cc_links_finish(s, lk, L.links, (size_t)q * L.Vh + t, L.ntLinks);
    if (threadIdx.x < N) {
      const float2 bv = cc_ldb(a, L, L.b.p[q] + (size_t)threadIdx.x * L.b.stride + t), h = s.yout[threadIdx.x];
      s.xin[8][threadIdx.x] = make_float2(bv.x - h.x, bv.y - h.y);
    }
    __syncthreads();
    cc_links_finish(s, lk2, L.hat, (size_t)q * L.Vh + t, L.ntLinks);
    if (threadIdx.x < N) stc(a, L.x.p[q] + (size_t)threadIdx.x * L.x.stride + t, s.yout[threadIdx.x]);
  }
}

// w = Yhat_pq Yhat_qp in_p in two phases (Mhat in = in - w: reference DiracCoarsePC::M, lib/dirac_coarse.cpp:332-350):
//   phase 1: t_q = Yhat_qp in_p (ends with a barrier);  phase 2: w at every p site, handed to epi(site, row, w, in(site)[row], row < N) on
//   the first wave.  No barrier at the end.
template <int N, typename Epi> __device__ __forceinline__ void cc_matpc(const CcArg &a, const CcLevel &L, CcShared &s, CcCtx &c, const CcVec &vin, Epi epi, bool pushed = false) {
  const int p = L.solvePar, q = 1 - p;
  if (!pushed) cc_exchange_begin(a, L, c, vin, 1 << p);   // pushed: the faces of vin went out when it was produced (exchange number c.seq)
  const unsigned seqT = c.seq + 1;                         // exchange number of the intermediate vector t, pushed by the sites that produce it
  for (int t = blockIdx.x; t < L.Vh; t += gridDim.x) {
    LinkRegs<N, 0, 8> lk;
    cc_links_issue(lk, L.hat, (size_t)q * L.Vh + t, L.ntLinks);
    cc_stage(a, L, s, c, vin, q, t, 0xff);
    cc_links_finish(s, lk, L.hat, (size_t)q * L.Vh + t, L.ntLinks);
    if (threadIdx.x < N) {
      const float2 v = s.yout[threadIdx.x];
      stc(a, L.t + (size_t)threadIdx.x * L.Vh + t, v);
      if (L.commMask) { int cs[4]; cc_coords(L, q, t, cs); cc_push_site(L, seqT, q, cs, (int)threadIdx.x, v); }
    }
  }
  cc_barrier(a, s, c);
  const CcVec vt = cc_parity_vec(L.t, q, L.Vh);
  if (L.commMask) c.seq = seqT;
  for (int t = blockIdx.x; t < L.Vh; t += gridDim.x) {
    LinkRegs<N, 0, 8> lk;
    cc_links_issue(lk, L.hat, (size_t)p * L.Vh + t, L.ntLinks);
    float2 iv = make_float2(0.f, 0.f);   // the site's own input: requested with the links, used in the epilogue
    if (threadIdx.x < N) iv = ldc(a, vin.p[p] + (size_t)threadIdx.x * vin.stride + t);
    cc_stage(a, L, s, c, vt, p, t, 0xff);
    cc_links_finish(s, lk, L.hat, (size_t)p * L.Vh + t, L.ntLinks);
    if (threadIdx.x < 64) {
      const bool on = threadIdx.x < N;
      float2 w = make_float2(0.f, 0.f);
      if (on) w = s.yout[threadIdx.x];
      epi(t, (int)threadIdx.x, w, iv, on);
    }
  }
}

// MR on the even-odd preconditioned system (reference lib/inv_mr_quda.cpp:40-200; host form solver.cpp MR::operator()):
//   alpha = omega (Ar, r) / |Ar|^2 ;  x += alpha r ;  r -= alpha Ar.   guess = false: x starts at 0 and r at bt;  guess = true: r = bt - Mhat x
// first.  Ends with a barrier.
template <int N> __device__ __forceinline__ void cc_mr(const CcArg &a, const CcLevel &L, CcShared &s, CcCtx &c, int nu, bool guess) {
  const int p = L.solvePar, Vh = L.Vh, nel = N * Vh;
  const CcVec xp = cc_parity_of(L.x, p), vbt = cc_parity_vec(L.bt, p, Vh), vr = cc_parity_vec(L.r, p, Vh);
  bool fresh = !guess;
  bool pushed = false;   // the residual's faces went out when it was produced (exchange number c.seq)
  if (guess) {
    const bool pushR = L.commMask && nu > 0;
    const unsigned seqR = c.seq + 3;   // (the matpc below uses c.seq + 1 for x and + 2 for t)
    cc_matpc<N>(a, L, s, c, xp, [&](int t, int j, float2 w, float2 iv, bool on) {
      if (on) {
        const float2 b = ldc(a, L.bt + (size_t)j * Vh + t);
        const float2 r = make_float2(b.x - iv.x + w.x, b.y - iv.y + w.y);
        stc(a, L.r + (size_t)j * Vh + t, r);
        if (pushR) { int cs[4]; cc_coords(L, p, t, cs); cc_push_site(L, seqR, p, cs, j, r); }
      }
    });
    if (pushR) { c.seq = seqR; pushed = true; }
    cc_barrier(a, s, c);
  } else if (nu == 0) {
    for (int e = blockIdx.x * kThreads + threadIdx.x; e < nel; e += gridDim.x * kThreads) {
      const int j = e / Vh, t = e - j * Vh;
      stc(a, L.x.p[p] + (size_t)j * L.x.stride + t, make_float2(0.f, 0.f));
      stc(a, L.r + e, ldc(a, L.bt + e));
    }
    cc_barrier(a, s, c);
  }
  for (int it = 0; it < nu; it++) {
    cc_clear_acc(s, 3);
    cc_matpc<N>(a, L, s, c, fresh ? vbt : vr, [&](int t, int j, float2 w, float2 iv, bool on) {
      const float2 Ar = make_float2(iv.x - w.x, iv.y - w.y);
      if (on) stc(a, L.Ar + (size_t)j * Vh + t, Ar);
      double re = on ? (double)Ar.x * iv.x + (double)Ar.y * iv.y : 0.0, im = on ? (double)Ar.x * iv.y - (double)Ar.y * iv.x : 0.0, nn = on ? (double)Ar.x * Ar.x + (double)Ar.y * Ar.y : 0.0;
      re = wave_sum(re); im = wave_sum(im); nn = wave_sum(nn);
      if (j == 0) { s.dacc[0] += re; s.dacc[1] += im; s.dacc[2] += nn; }
    }, pushed);
    cc_reduce(a, s, c, 3, L.mrGlobal != 0);
    const double z = s.red[2], sc = z > 0.0 ? (double)L.omega / z : 0.0;
    const float ar = (float)(sc * s.red[0]), ai = (float)(sc * s.red[1]);
    const bool pushR = L.commMask && it + 1 < nu;   // the next iteration hops the new residual: its faces leave with the update
    const unsigned seqR = c.seq + 1;
    for (int e = blockIdx.x * kThreads + threadIdx.x; e < nel; e += gridDim.x * kThreads) {
      const int j = e / Vh, t = e - j * Vh;
      float2 *xe = L.x.p[p] + (size_t)j * L.x.stride + t;
      const float2 Ar = ldc(a, L.Ar + e);
      float2 rn;
      if (fresh) {
        const float2 b = ldc(a, L.bt + e);
        stc(a, xe, make_float2(ar * b.x - ai * b.y, ar * b.y + ai * b.x));
        rn = make_float2(b.x - (ar * Ar.x - ai * Ar.y), b.y - (ar * Ar.y + ai * Ar.x));
      } else {
        const float2 r = ldc(a, L.r + e), x0 = ldc(a, xe);
        stc(a, xe, make_float2(x0.x + ar * r.x - ai * r.y, x0.y + ar * r.y + ai * r.x));
        rn = make_float2(r.x - (ar * Ar.x - ai * Ar.y), r.y - (ar * Ar.y + ai * Ar.x));
      }
      stc(a, L.r + e, rn);
      if (pushR) { int cs[4]; cc_coords(L, p, t, cs); cc_push_site(L, seqR, p, cs, j, rn); }
    }
    pushed = pushR;
    if (pushR) c.seq = seqR;
    fresh = false;
    cc_barrier(a, s, c);
  }
}

// rf = b - M x on all sites (reference MG::operator(), lib/multigrid.cpp:540-548; M = DiracCoarse::M, lib/dslash_coarse.cu:216-234).  No barrier at the end.
template <int N> __device__ __forceinline__ void cc_residual(const CcArg &a, const CcLevel &L, CcShared &s, CcCtx &c) {
  cc_exchange_begin(a, L, c, L.x, 3);
  for (int A = blockIdx.x; A < 2 * L.Vh; A += gridDim.x) {
    const int par = A >= L.Vh, t = A - par * L.Vh;
    LinkRegs<N, 0, 9> lk;
    cc_links_issue(lk, L.links, (size_t)A, L.ntLinks);
    cc_stage(a, L, s, c, L.x, par, t, 0x1ff);
    cc_links_finish(s, lk, L.links, (size_t)A, L.ntLinks);
    if (threadIdx.x < N) {
      const float2 bv = cc_ldb(a, L, L.b.p[par] + (size_t)threadIdx.x * L.b.stride + t), h = s.yout[threadIdx.x];
      stc(a, L.rf.p[par] + (size_t)threadIdx.x * L.rf.stride + t, make_float2(bv.x - h.x, bv.y - h.y));
    }
  }
}

// The same residual without the operator: behind the even-odd pre-smoother (x_q reconstructed or not) the full residual is rf_p = X_pp r~ with r~ the
// residual MR ended with (L.r), rf_q = 0 — MG::imageOfLast / MG::cycleUnfused.  One local matrix per solved site, no halo, and the reconstruct phase
// in front of cc_residual is not needed either.  No barrier at the end.
template <int N> __device__ __forceinline__ void cc_residual_local(const CcArg &a, const CcLevel &L, CcShared &s, CcCtx &c) {
  const int p = L.solvePar, q = 1 - p;
  for (int t = blockIdx.x; t < L.Vh; t += gridDim.x) {
    LinkRegs<N, 8, 9> lk;
    cc_links_issue(lk, L.links, (size_t)p * L.Vh + t, L.ntLinks);
    if (threadIdx.x < N) s.xin[8][threadIdx.x] = ldc(a, L.r + (size_t)threadIdx.x * L.Vh + t);
    __syncthreads();
    cc_links_finish(s, lk, L.links, (size_t)p * L.Vh + t, L.ntLinks);
    if (threadIdx.x < N) {
      stc(a, L.rf.p[p] + (size_t)threadIdx.x * L.rf.stride + t, s.yout[threadIdx.x]);
      stc(a, L.rf.p[q] + (size_t)threadIdx.x * L.rf.stride + t, make_float2(0.f, 0.f));
    }
  }
}

// next.b = R rf: V^dagger summed over each aggregate, per chirality (reference lib/restrictor.cu:51-125; the lane-group scheme of
// transfer.hip restrict_small_kernel).  N = components of this level, NC = 2 Nvec of the next.  No barrier at the end.
template <int N> __device__ __forceinline__ void cc_restrict(const CcArg &a, const CcLevel &L, const CcLevel &C) {
  constexpr int NVEC = N / 2;   // every fused level has the same n (checked on the host)
  const int bv = L.blockVol, GS = L.GS;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int gpw = 64 / GS, b = lane % GS, grp = lane / GS;
  const int slot = wave * gpw + grp, nslots = 4 * gpw;
  for (int A = blockIdx.x; A < L.nAgg; A += gridDim.x) {
    __syncthreads();
    for (int e0 = threadIdx.x; e0 < bv * N; e0 += 4 * kThreads) {   // four elements per thread at a time: map loads, then field loads, back to back
      int f[4], kk[4], bb[4]; bool ok[4]; float2 v[4];
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const int e = e0 + i * kThreads;
        ok[i] = e < bv * N; bb[i] = ok[i] ? e / N : 0; kk[i] = ok[i] ? e - bb[i] * N : 0;
        f[i] = *gp(L.b2f + (size_t)A * bv + bb[i]);
      }
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const int par = f[i] >= L.Vh, x = f[i] - par * L.Vh;
        v[i] = ldc(a, L.rf.p[par] + (size_t)kk[i] * L.rf.stride + x);
      }
#pragma unroll
      for (int i = 0; i < 4; i++) if (ok[i]) cc_agg[bb[i] * N + kk[i]] = v[i];
    }
    __syncthreads();
    const int cpar = A >= C.Vh, xc = A - cpar * C.Vh;
    for (int it0 = 0; it0 < NVEC; it0 += nslots) {
      const int it = it0 + slot;
      const bool live = it < NVEC && b < bv;
      const int chi = live ? it / (NVEC / 2) : 0, vp = live ? it - chi * (NVEC / 2) : 0;
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      {
        float4 wv[N / 2];   // all V entries of the iteration requested back to back (one round trip instead of N / 2 dependent ones)
#pragma unroll
        for (int kk = 0; kk < N / 2; kk++) wv[kk] = cc_ld_nt(L.V + (((size_t)A * N + chi * (N / 2) + kk) * (NVEC / 2) + vp) * bv + (live ? b : 0));
#pragma unroll
        for (int kk = 0; kk < N / 2; kk++) {
          const int k = chi * (N / 2) + kk;
          const float4 w = live ? wv[kk] : make_float4(0.f, 0.f, 0.f, 0.f);
          const float2 r = cc_agg[(live ? b : 0) * N + k];
          acc.x += w.x * r.x + w.y * r.y; acc.y += w.x * r.y - w.y * r.x;
          acc.z += w.z * r.x + w.w * r.y; acc.w += w.z * r.y - w.w * r.x;
        }
      }
      for (int off = GS >> 1; off > 0; off >>= 1) {
        acc.x += __shfl_down(acc.x, off, GS); acc.y += __shfl_down(acc.y, off, GS); acc.z += __shfl_down(acc.z, off, GS); acc.w += __shfl_down(acc.w, off, GS);
      }
      if (live && b == 0) {
        const int c0 = chi * NVEC + 2 * vp;
        stc(a, C.b.p[cpar] + (size_t)c0 * C.b.stride + xc, make_float2(acc.x, acc.y));
        stc(a, C.b.p[cpar] + (size_t)(c0 + 1) * C.b.stride + xc, make_float2(acc.z, acc.w));
      }
    }
  }
}

// x += P next.x (reference lib/prolongator.cu:42-116, MG::operator() :575-580).  No barrier at the end.
template <int N> __device__ __forceinline__ void cc_prolong_add(const CcArg &a, const CcLevel &L, const CcLevel &C) {
  constexpr int NVEC = N / 2;
  const int bv = L.blockVol;
  for (int A = blockIdx.x; A < L.nAgg; A += gridDim.x) {
    const int cpar = A >= C.Vh, xc = A - cpar * C.Vh;
    __syncthreads();
    for (int j = threadIdx.x; j < C.n; j += kThreads) cc_agg[j] = ldc(a, C.x.p[cpar] + (size_t)j * C.x.stride + xc);
    __syncthreads();
    for (int e0 = threadIdx.x; e0 < bv * N; e0 += 3 * kThreads) {   // three fine elements per thread at a time, all their loads requested together
      float4 wv[3][NVEC / 2];
      float2 x0[3]; float2 *xe[3]; int kq[3]; bool ok[3];
#pragma unroll
      for (int i = 0; i < 3; i++) {
        const int e = e0 + i * kThreads;
        ok[i] = e < bv * N;
        const int k = ok[i] ? e / bv : 0, b = ok[i] ? e - k * bv : 0;
        kq[i] = k;
        const int f = *gp(L.b2f + (size_t)A * bv + b);
        const int par = f >= L.Vh, x = f - par * L.Vh;
        xe[i] = L.x.p[par] + (size_t)k * L.x.stride + x;
#pragma unroll
        for (int vp = 0; vp < NVEC / 2; vp++) wv[i][vp] = cc_ld_nt(L.V + (((size_t)A * N + k) * (NVEC / 2) + vp) * bv + b);
      }
#pragma unroll
      for (int i = 0; i < 3; i++) x0[i] = ldc(a, xe[i]);
#pragma unroll
      for (int i = 0; i < 3; i++) {
        const int chi = kq[i] / (N / 2);
        float re = 0.f, im = 0.f;
#pragma unroll
        for (int vp = 0; vp < NVEC / 2; vp++) {
          const float4 w = wv[i][vp];
          const float2 c0 = cc_agg[chi * NVEC + 2 * vp], c1 = cc_agg[chi * NVEC + 2 * vp + 1];
          re += w.x * c0.x - w.y * c0.y + w.z * c1.x - w.w * c1.y;
          im += w.x * c0.y + w.y * c0.x + w.z * c1.y + w.w * c1.x;
        }
        if (ok[i]) stc(a, xe[i], make_float2(x0[i].x + re, x0[i].y + im));
      }
    }
  }
}

// Coarsest grid: restarted GCR(nKrylov) on Mhat x_p = bt to |r| <= tol |bt| (reference lib/inv_gcr_quda.cpp:235-516 without a preconditioner,
// in the host form of solver.cpp GCR::operator(): all inner products of an iteration in one sweep, classical Gram-Schmidt with the norm
// inferred, the sequential chain where that difference loses two digits; true residual at every restart).  |bt|^2 is expected in s.red[0].
// Ends with a barrier.
template <int N> __device__ __forceinline__ void cc_gcr(const CcArg &a, const CcLevel &L, CcShared &s, CcCtx &c, unsigned stats[3]) {
  const int p = L.solvePar, Vh = L.Vh, nel = N * Vh;
  const double b2 = s.red[0];
  __syncthreads();
  const int e0 = blockIdx.x * kThreads + threadIdx.x, estep = gridDim.x * kThreads;
  if (!(b2 > 0.0)) {   // zero (or broken) source: x = b
    for (int e = e0; e < nel; e += estep) { const int j = e / Vh, t = e - j * Vh; stc(a, L.x.p[p] + (size_t)j * L.x.stride + t, ldc(a, L.bt + e)); }
    cc_barrier(a, s, c);
    return;
  }
  const double stop = a.tol * a.tol * b2;
  // (the faces of every new residual leave with the phase that produces it: pushR / `pushed`, see cc_matpc)
  auto pushElem = [&](unsigned seq, int e, float2 v) { const int j = e / Vh, t = e - j * Vh; int cs[4]; cc_coords(L, p, t, cs); cc_push_site(L, seq, p, cs, j, v); };
  const bool comm = L.commMask != 0;
  {
    const unsigned seqR = c.seq + 1;
    for (int e = e0; e < nel; e += estep) { const float2 v = ldc(a, L.bt + e); stc(a, L.r + e, v); stc(a, a.y + e, make_float2(0.f, 0.f)); if (comm) pushElem(seqR, e, v); }
    if (comm) c.seq = seqR;
  }
  bool pushed = comm;
  cc_barrier(a, s, c);
  const CcVec vr = cc_parity_vec(L.r, p, Vh), vy = cc_parity_vec(a.y, p, Vh);
  double r2 = b2, r2_old = b2;
  int k = 0, total = 0, restarts = 0, resInc = 0, resIncTotal = 0;
  bool l2conv = false;
  while (r2 > stop && total < a.maxiter && !c.dead) {
    float2 *Pk = a.P + (size_t)k * nel, *APk = a.AP + (size_t)k * nel;
    const int K = 2 * k + 3;
    cc_clear_acc(s, K);
    // p_k = r ; Ap_k = Mhat p_k ; all inner products of the iteration from the registers of the epilogue
    cc_matpc<N>(a, L, s, c, vr, [&](int t, int j, float2 w, float2 iv, bool on) {
      const float2 Ap = make_float2(iv.x - w.x, iv.y - w.y);
      const size_t o = (size_t)j * Vh + t;
      if (on) { stc(a, APk + o, Ap); stc(a, Pk + o, iv); }
      float2 qs[kKrylovMax];   // all directions' values requested back to back: one round trip, not k
#pragma unroll
      for (int i = 0; i < kKrylovMax; i++) qs[i] = (on && i < k) ? ldc(a, a.AP + (size_t)i * nel + o) : make_float2(0.f, 0.f);
#pragma unroll
      for (int i = 0; i < kKrylovMax; i++) {
        if (i >= k) break;
        const float2 q = qs[i];
        double re = (double)q.x * Ap.x + (double)q.y * Ap.y, im = (double)q.x * Ap.y - (double)q.y * Ap.x;
        re = wave_sum(re); im = wave_sum(im);
        if (j == 0) { s.dacc[2 * i] += re; s.dacc[2 * i + 1] += im; }
      }
      double re = on ? (double)Ap.x * iv.x + (double)Ap.y * iv.y : 0.0, im = on ? (double)Ap.x * iv.y - (double)Ap.y * iv.x : 0.0, nn = on ? (double)Ap.x * Ap.x + (double)Ap.y * Ap.y : 0.0;
      re = wave_sum(re); im = wave_sum(im); nn = wave_sum(nn);
      if (j == 0) { s.dacc[2 * k] += re; s.dacc[2 * k + 1] += im; s.dacc[2 * k + 2] += nn; }
    }, pushed);
    pushed = false;
    cc_reduce(a, s, c, K, true);
    const double apn = s.red[2 * k + 2];
    double g2 = apn;
    for (int i = 0; i < k; i++) g2 -= s.red[2 * i] * s.red[2 * i] + s.red[2 * i + 1] * s.red[2 * i + 1];
    if (!(apn > 0.0)) {   // GCR breakdown
      if (threadIdx.x == 0) cc_fail(a, 60, k, total, 0, 0, 0, 0, 0);
      c.dead = true;
      break;
    }
    if (g2 > 1e-2 * apn) {
      const double gamma = sqrt(g2), apr = s.red[2 * k], api = s.red[2 * k + 1];
      __syncthreads();
      if (threadIdx.x == 0) {
        for (int i = 0; i < k; i++) { s.betaRe[i][k] = s.red[2 * i]; s.betaIm[i][k] = s.red[2 * i + 1]; }
        s.gam[k] = gamma; s.alRe[k] = apr / gamma; s.alIm[k] = api / gamma;
      }
      __syncthreads();
      cc_clear_acc(s, 1);
      const float ig = (float)(1.0 / gamma), alr = (float)(apr / gamma), ali = (float)(api / gamma);
      double nn = 0.0;
      const unsigned seqR = c.seq + 1;
      for (int e = e0; e < nel; e += estep) {
        float2 v = ldc(a, APk + e);
        for (int i = 0; i < k; i++) {
          const float br = (float)s.betaRe[i][k], bi = (float)s.betaIm[i][k];
          const float2 q = ldc(a, a.AP + (size_t)i * nel + e);
          v.x -= br * q.x - bi * q.y; v.y -= br * q.y + bi * q.x;
        }
        v.x *= ig; v.y *= ig;
        stc(a, APk + e, v);
        float2 r = ldc(a, L.r + e);
        r.x -= alr * v.x - ali * v.y; r.y -= alr * v.y + ali * v.x;
        stc(a, L.r + e, r);
        if (comm) pushElem(seqR, e, r);
        nn += (double)r.x * r.x + (double)r.y * r.y;
      }
      if (comm) { c.seq = seqR; pushed = true; }
      cc_block_add(s, 0, nn);
      cc_reduce(a, s, c, 1, true);
      r2 = s.red[0];
    } else {
      // the new direction lies almost in the span of the old ones: modified Gram-Schmidt, every norm measured (solver.cpp orthoDir)
      for (int i = 0; i < k; i++) {
        cc_clear_acc(s, 2);
        double re = 0.0, im = 0.0;
        for (int e = e0; e < nel; e += estep) { const float2 q = ldc(a, a.AP + (size_t)i * nel + e), v = ldc(a, APk + e); re += (double)q.x * v.x + (double)q.y * v.y; im += (double)q.x * v.y - (double)q.y * v.x; }
        cc_block_add(s, 0, re); cc_block_add(s, 1, im);
        cc_reduce(a, s, c, 2, true);
        const float br = (float)s.red[0], bi = (float)s.red[1];
        __syncthreads();
        if (threadIdx.x == 0) { s.betaRe[i][k] = s.red[0]; s.betaIm[i][k] = s.red[1]; }
        for (int e = e0; e < nel; e += estep) { const float2 q = ldc(a, a.AP + (size_t)i * nel + e); float2 v = ldc(a, APk + e); v.x -= br * q.x - bi * q.y; v.y -= br * q.y + bi * q.x; stc(a, APk + e, v); }
      }
      cc_clear_acc(s, 3);
      double re = 0.0, im = 0.0, nn = 0.0;
      for (int e = e0; e < nel; e += estep) { const float2 v = ldc(a, APk + e), r = ldc(a, L.r + e); re += (double)v.x * r.x + (double)v.y * r.y; im += (double)v.x * r.y - (double)v.y * r.x; nn += (double)v.x * v.x + (double)v.y * v.y; }
      cc_block_add(s, 0, re); cc_block_add(s, 1, im); cc_block_add(s, 2, nn);
      cc_reduce(a, s, c, 3, true);
      if (!(s.red[2] > 0.0)) {
        if (threadIdx.x == 0) cc_fail(a, 60, k, total, 0, 0, 0, 0, 0);
        c.dead = true;
        break;
      }
      const double gamma = sqrt(s.red[2]), apr = s.red[0], api = s.red[1];
      __syncthreads();
      if (threadIdx.x == 0) { s.gam[k] = gamma; s.alRe[k] = apr / gamma; s.alIm[k] = api / gamma; }
      cc_clear_acc(s, 1);
      const float ig = (float)(1.0 / gamma), alr = (float)(apr / gamma), ali = (float)(api / gamma);
      double n2 = 0.0;
      const unsigned seqR = c.seq + 1;
      for (int e = e0; e < nel; e += estep) {
        float2 v = ldc(a, APk + e); v.x *= ig; v.y *= ig; stc(a, APk + e, v);
        float2 r = ldc(a, L.r + e); r.x -= alr * v.x - ali * v.y; r.y -= alr * v.y + ali * v.x; stc(a, L.r + e, r);
        if (comm) pushElem(seqR, e, r);
        n2 += (double)r.x * r.x + (double)r.y * r.y;
      }
      if (comm) { c.seq = seqR; pushed = true; }
      cc_block_add(s, 0, n2);
      cc_reduce(a, s, c, 1, true);
      r2 = s.red[0];
    }
    k++; total++;
    if (k == a.nKrylov || total == a.maxiter || (r2 < stop && !l2conv) || sqrt(r2 / r2_old) < a.delta) {
      // update the solution (back substitution, reference :125-157), then the true residual r = bt - Mhat y
      __syncthreads();
      if (threadIdx.x == 0) {
        for (int i = k - 1; i >= 0; i--) {
          double dr = s.alRe[i], di = s.alIm[i];
          for (int j = i + 1; j < k; j++) { dr -= s.betaRe[i][j] * s.dlRe[j] - s.betaIm[i][j] * s.dlIm[j]; di -= s.betaRe[i][j] * s.dlIm[j] + s.betaIm[i][j] * s.dlRe[j]; }
          s.dlRe[i] = dr / s.gam[i]; s.dlIm[i] = di / s.gam[i];
        }
      }
      __syncthreads();
      for (int e = e0; e < nel; e += estep) {
        float2 yv = ldc(a, a.y + e);
        for (int i = 0; i < k; i++) {
          const float dr = (float)s.dlRe[i], di = (float)s.dlIm[i];
          const float2 q = ldc(a, a.P + (size_t)i * nel + e);
          yv.x += dr * q.x - di * q.y; yv.y += dr * q.y + di * q.x;
        }
        stc(a, a.y + e, yv);
      }
      cc_barrier(a, s, c);
      cc_clear_acc(s, 1);
      const unsigned seqR2 = c.seq + 3;   // (the matpc below: + 1 for y, + 2 for t)
      cc_matpc<N>(a, L, s, c, vy, [&](int t, int j, float2 w, float2 iv, bool on) {
        const size_t o = (size_t)j * Vh + t;
        float2 rr = make_float2(0.f, 0.f);
        if (on) {
          const float2 b = ldc(a, L.bt + o); rr = make_float2(b.x - iv.x + w.x, b.y - iv.y + w.y); stc(a, L.r + o, rr);
          if (comm) { int cs[4]; cc_coords(L, p, t, cs); cc_push_site(L, seqR2, p, cs, j, rr); }
        }
        double nn = (double)rr.x * rr.x + (double)rr.y * rr.y;
        nn = wave_sum(nn);
        if (j == 0) s.dacc[0] += nn;
      });
      if (comm) { c.seq = seqR2; pushed = true; }
      cc_reduce(a, s, c, 1, true);
      r2 = s.red[0];
      if (r2 > r2_old) {
        resInc++; resIncTotal++;
        if (resInc > a.maxResInc || resIncTotal > a.maxResIncTotal) break;
      } else {
        resInc = 0;
      }
      k = 0;
      if (r2 > stop) { restarts++; if (r2 < stop) l2conv = true; }
      r2_old = r2;
    }
  }
  for (int e = e0; e < nel; e += estep) {
    const int j = e / Vh, t = e - j * Vh;
    stc(a, L.x.p[p] + (size_t)j * L.x.stride + t, total > 0 ? ldc(a, a.y + e) : make_float2(0.f, 0.f));
  }
  stats[0] = (unsigned)total; stats[1] = (unsigned)restarts; stats[2] = r2 > stop ? 1u : 0u;
  cc_barrier(a, s, c);
}

}  // namespace

// (outside the anonymous namespace, so that profilers print a plain kernel name)
// The argument block lives in device memory (the levels are indexed at run time: a by-value kernel argument would be copied to scratch)
template <int N> __global__ void __launch_bounds__(kThreads) coarse_cycle_kernel(const CcArg *__restrict__ ap) {
  // The argument block is copied into LDS first: the phases read its fields (extents, pointers, strides) all the time, and read from global
  // memory every such read is a ~1 us round trip the compiler cannot hoist across the stores in between (the first version spent most of a
  // 21 us hop phase on them).
  __shared__ CcArg sa;
  {
    const unsigned *src = reinterpret_cast<const unsigned *>(ap);
    unsigned *dst = reinterpret_cast<unsigned *>(&sa);
    for (int i = threadIdx.x; i < (int)(sizeof(CcArg) / sizeof(unsigned)); i += kThreads) dst[i] = src[i];
    __syncthreads();
  }
  const CcArg &a = sa;
  __shared__ CcShared s;
  CcCtx c;
  c.nred = 0; c.dead = false;
  c.seq = *gp(a.state); c.rseq = *gp(a.state + 1); c.epoch = c.epoch0 = *gp(a.state + 7);
  unsigned gcrStats[3] = {0, 0, 0};
  const unsigned seq0 = c.seq;
  if (a.timeline && blockIdx.x == 0 && threadIdx.x == 0) a.timeline[0] = wall_clock64();
  const int last = a.nl - 1;
  // down: pre-smooth, residual, restrict
  for (int l = 0; l < last; l++) {
    const CcLevel &L = a.L[l];
    cc_prepare<N>(a, L, s, c, false);
    cc_barrier(a, s, c);
    cc_mr<N>(a, L, s, c, L.nuPre, false);
    if (a.resFromSmoother && L.nuPre > 0) {
      cc_residual_local<N>(a, L, s, c);
    } else {
      cc_reconstruct<N>(a, L, s, c);
      cc_barrier(a, s, c);
      cc_residual<N>(a, L, s, c);
    }
    cc_barrier(a, s, c);
    cc_restrict<N>(a, L, a.L[l + 1]);
    cc_barrier(a, s, c);
  }
  // coarsest grid
  {
    const CcLevel &L = a.L[last];
    cc_prepare<N>(a, L, s, c, true);
    cc_reduce(a, s, c, 1, true);
    cc_gcr<N>(a, L, s, c, gcrStats);
    cc_reconstruct<N>(a, L, s, c);
    cc_barrier(a, s, c);
  }
  // up: prolongate and correct, post-smooth
  for (int l = last - 1; l >= 0; l--) {
    const CcLevel &L = a.L[l];
    cc_prolong_add<N>(a, L, a.L[l + 1]);
    cc_barrier(a, s, c);
    cc_mr<N>(a, L, s, c, L.nuPost, true);
    cc_reconstruct<N>(a, L, s, c);
    cc_barrier(a, s, c);
  }
  // the solution leaves the slab
  {
    const CcLevel &L = a.L[0];
    const int nel2 = 2 * N * L.Vh;
    for (int e = blockIdx.x * kThreads + threadIdx.x; e < nel2; e += gridDim.x * kThreads) {
      const int par = e >= N * L.Vh, f = e - par * N * L.Vh, j = f / L.Vh, t = f - j * L.Vh;
      *gp(reinterpret_cast<unsigned long long *>(a.xOut.p[par] + (size_t)j * a.xOut.stride + t)) = __builtin_bit_cast(unsigned long long, ldc(a, L.x.p[par] + (size_t)j * L.x.stride + t));
    }
  }
  // (every work-group has passed the last barrier, i.e. nobody still reads the state words of this launch)
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    a.state[0] = c.seq; a.state[1] = c.rseq; a.state[7] = c.epoch;
    a.state[2] = c.epoch - c.epoch0; a.state[3] = gcrStats[0]; a.state[4] = gcrStats[1]; a.state[5] = c.seq - seq0; a.state[6] = gcrStats[2];
  }
}


// ================================================================================================
// host side
// ================================================================================================
class CoarseCycle {
 public:
  CcArg arg;
  CcArg *d_arg = nullptr;        // device copy, refreshed when the caller's fields change
  bool argDirty = true;
  int n = 0, grid = 0;
  size_t ldsBytes = 0;
  float2 *work = nullptr;        // the slab: work vectors, then the partial sums
  size_t workBytes = 0;
  unsigned long long *timeline = nullptr;
  unsigned *sync = nullptr;      // state[16] + barrier words (release word + one 64-byte slot per work-group)
  double *partial = nullptr;
  char *window = nullptr;        // halo zones of all levels
  PeerMap *map = nullptr;
  char *redWindow = nullptr;
  std::vector<void *> redByRank, redOpened;
  long long launches = 0;
};

static int g_fusedEnabled = -1;
int coarseCycleEnabled() {
  if (g_fusedEnabled < 0) { const char *e = getenv("QUDA_AMD_MG_FUSED"); g_fusedEnabled = e ? atoi(e) : 1; }
  return g_fusedEnabled;
}
void coarseCycleSetEnabled(int on) { g_fusedEnabled = on ? 1 : 0; }

static int envInt(const char *name, int dflt) { const char *e = getenv(name); return e ? atoi(e) : dflt; }

template <int N> static void launchCycle(const CoarseCycle &cc) {
  hipLaunchKernelGGL((coarse_cycle_kernel<N>), dim3(cc.grid), dim3(kThreads), cc.ldsBytes, computeStream(), (const CcArg *)cc.d_arg);
}
template <int N> static int occupancyOf(size_t lds) {
  int nb = 0;
  HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, coarse_cycle_kernel<N>, kThreads, lds));
  return nb;
}

CoarseCycle *coarseCycleCreate(MG &top) {
  if (!coarseCycleEnabled()) return nullptr;
  const MGParam &tp = top.params();
  if (tp.level < 1) return nullptr;
  const int nl = tp.Nlevel - tp.level;
  if (nl < 1 || nl > kMaxLevels) return nullptr;
  // (half-precision storage of the hierarchy: the fused cycle keeps reading the fp32 masters of the coarse links — below level 0 a phase is
  // bound by latency, not by bytes — while R / P of level 0 and its smoother use the fp16 / 16-bit mirrors)
  const CommGrid &g = commGrid();
  if (g.size > kMaxRanks) return nullptr;
  if (commReductionsNeeded() && g.size == 1) return nullptr;          // RCCL self-test mode: sums must go through the collective
  if (p2pDeviceShared() && !envInt("QUDA_AMD_MG_FUSED_SHARED", 0)) return nullptr;   // ranks sharing a device: their persistent kernels may not be co-resident
  CoarseCycle *cc = new CoarseCycle;
  CcArg &a = cc->arg;
  memset(&a, 0, sizeof(a));
  a.nl = nl;
  const int maxSites = envInt("QUDA_AMD_MG_FUSED_MAX_SITES", 4096);
  MG *m = &top;
  bool ok = true, anyComm = false;
  size_t elems = 0;      // float2 elements of the work slab
  size_t winBytes = 0;
  size_t lds = 0;
  int maxTasks = 0;
  for (int l = 0; l < nl && ok; l++, m = m->getCoarse()) {
    if (!m) { ok = false; break; }
    const MGParam &p = m->params();
    const bool coarsest = p.level == p.Nlevel - 1;
    const DiracCoarse *dc = dynamic_cast<const DiracCoarse *>(p.matResidual.Expose());
    const DiracCoarsePC *ds = dynamic_cast<const DiracCoarsePC *>(p.matSmooth.Expose());
    if (!dc || !ds || !m->smootherIsPC()) { ok = false; break; }
    const CoarseGauge &Y = dc->Links(), &H = ds->HatLinks();
    CcLevel &L = a.L[l];
    for (int d = 0; d < 4; d++) { L.Xc[d] = Y.Xc[d]; if (Y.Xc[d] % 2) ok = false; }
    L.Vh = Y.nSites / 2; L.n = Y.n;
    if (l == 0) { cc->n = Y.n; if (Y.nSites > maxSites) ok = false; }
    if (Y.n != cc->n || (Y.n != 16 && Y.n != 32 && Y.n != 48 && Y.n != 64)) ok = false;
    L.links = reinterpret_cast<const float4 *>(Y.data); L.hat = reinterpret_cast<const float4 *>(H.data);
    const QudaMatPCType pc = ds->getMatPCType();
    if (pc == QUDA_MATPC_EVEN_EVEN) L.solvePar = 0;
    else if (pc == QUDA_MATPC_ODD_ODD) L.solvePar = 1;
    else ok = false;
    L.omega = (float)p.omega;
    L.mrGlobal = p.global_reduction && commReductionsNeeded() ? 1 : 0;
    L.ntLinks = (size_t)2 * Y.nSites * 9 * Y.n * Y.n * 8 > ((size_t)24 << 20) ? 1 : 0;   // links + preconditioned links against the 8 x 4 MB of L2
    L.commMask = 0;
    for (int d = 0; d < 4; d++) { L.faceCB[d] = L.Vh / L.Xc[d]; if (g.partitioned(d)) L.commMask |= 1 << d; }
    if (L.commMask) anyComm = true;
    maxTasks = std::max(maxTasks, 2 * L.Vh);
    if (!coarsest) {
      const Transfer *T = m->getTransfer();
      if (p.smoother != QUDA_MR_INVERTER) ok = false;
      if (!(p.cycle_type == QUDA_MG_CYCLE_VCYCLE || p.level == p.Nlevel - 2)) ok = false;
      if (!T || T->spin_bs != 1 || T->fineSpin != 2 || 2 * T->fineColor != Y.n || T->blockVol > 64 || T->Nvec % 2) { ok = false; break; }
      L.nuPre = p.nu_pre; L.nuPost = p.nu_post;
      L.V = reinterpret_cast<const float4 *>(T->V); L.b2f = T->block_to_fine; L.blockVol = T->blockVol; L.nAgg = T->nAgg;
      int GS = 1; while (GS < T->blockVol) GS <<= 1;
      L.GS = GS;
      lds = std::max(lds, (size_t)std::max(T->blockVol * Y.n, 2 * T->Nvec) * sizeof(float2));
      maxTasks = std::max(maxTasks, T->nAgg);
    } else {
      const SolverParam *sp = m->preSmootherParam();
      if (!sp || sp->inv_type != QUDA_GCR_INVERTER || sp->Nkrylov > kKrylovMax || sp->Nkrylov < 1) { ok = false; break; }
      a.nKrylov = sp->Nkrylov; a.maxiter = sp->maxiter; a.tol = sp->tol; a.delta = sp->delta;
      a.maxResInc = sp->max_res_increase; a.maxResIncTotal = sp->max_res_increase_total;
    }
    // work vectors: rf (full), bt, r, Ar, t (parity); below the top level also b and x (full); coarsest: 2 nKrylov + 1 parity vectors
    const size_t pv = (size_t)L.n * L.Vh;
    elems += 6 * pv + 2 * pv + (l > 0 ? 2 * pv : 0) + (coarsest ? (size_t)(2 * a.nKrylov + 1) * pv : 0);
    if (L.commMask)
      for (int d = 0; d < 4; d++) if ((L.commMask >> d) & 1) winBytes += (size_t)2 * kHaloBufs * 2 * L.n * L.faceCB[d] * sizeof(u32x4_t);
  }
  if (ok && anyComm && !p2pHaloEnabled()) ok = false;   // staged (RCCL) transport: no peer windows to push into
  if (!ok) { delete cc; return nullptr; }
  cc->ldsBytes = lds;
  // all work-groups must be resident at once
  int occ = 0;
  switch (cc->n) {
    case 16: occ = occupancyOf<16>(lds); break;
    case 32: occ = occupancyOf<32>(lds); break;
    case 48: occ = occupancyOf<48>(lds); break;
    default: occ = occupancyOf<64>(lds); break;
  }
  int dev = 0;
  HIP_CHECK(hipGetDevice(&dev));
  hipDeviceProp_t prop;
  HIP_CHECK(hipGetDeviceProperties(&prop, dev));
  int cap = std::min(occ, 2) * prop.multiProcessorCount;
  if (p2pDeviceShared()) cap = std::min(cap, 64);
  cap = std::min(cap, envInt("QUDA_AMD_MG_FUSED_GRID", 512));
  cc->grid = std::max(1, std::min(cap, maxTasks));
  if (occ < 1) { delete cc; return nullptr; }

  const size_t partialBytes = (size_t)2 * cc->grid * kRedMax * sizeof(double);
  cc->workBytes = elems * sizeof(float2) + partialBytes;
  if (cc->workBytes >= ((size_t)1 << 31)) { delete cc; return nullptr; }   // 32-bit buffer offsets
  HIP_CHECK(qaMalloc(&cc->work, cc->workBytes));
  HIP_CHECK(hipMemsetAsync(cc->work, 0, cc->workBytes, computeStream()));
  a.slab = (char *)cc->work; a.slabBytes = (unsigned)cc->workBytes;
  cc->partial = (double *)((char *)cc->work + elems * sizeof(float2));
  if (envInt("QUDA_AMD_MG_FUSED_TIMELINE", 0)) {
    HIP_CHECK(hipHostMalloc((void **)&cc->timeline, 1024 * sizeof(unsigned long long), hipHostMallocDefault));
    memset(cc->timeline, 0, 1024 * sizeof(unsigned long long));
    a.timeline = cc->timeline;
  }
  HIP_CHECK(qaMalloc(&cc->d_arg, sizeof(CcArg)));
  const size_t syncWords = 16 + 16 * (size_t)(1 + cc->grid);
  HIP_CHECK(qaMalloc(&cc->sync, syncWords * sizeof(unsigned)));
  HIP_CHECK(hipMemsetAsync(cc->sync, 0, syncWords * sizeof(unsigned), computeStream()));
  a.state = cc->sync; a.bar = cc->sync + 16; a.partial = cc->partial;
  a.world = g.size; a.rank = g.rank;
  a.errWord = p2pErrorWord(); a.waitTicks = p2pTimeoutTicks();
  { const char *e = getenv("QUDA_AMD_MG_SMOOTHER_RESIDUAL"); a.resFromSmoother = e ? atoi(e) : 1; }   // the same switch as MG::cycleUnfused
  float2 *w = cc->work;
  for (int l = 0; l < nl; l++) {
    CcLevel &L = a.L[l];
    const size_t pv = (size_t)L.n * L.Vh;
    auto full = [&](CcVec &v) { v.p[0] = w; v.p[1] = w + pv; v.stride = L.Vh; w += 2 * pv; };
    full(L.rf);
    L.bt = w; w += pv; L.r = w; w += pv; L.Ar = w; w += pv; L.t = w; w += pv;
    full(L.x);
    if (l > 0) full(L.b);
    L.bExternal = l == 0;
    if (l == nl - 1) { a.P = w; w += (size_t)a.nKrylov * pv; a.AP = w; w += (size_t)a.nKrylov * pv; a.y = w; w += pv; }
  }
  if (anyComm) {
    cc->window = (char *)p2pAlloc(winBytes);
    HIP_CHECK(hipMemset(cc->window, 0, winBytes));
    cc->map = new PeerMap;
    if (!commMapPeers(cc->window, *cc->map)) errorQuda("peer mapping of the fused coarse-cycle window failed after the transport probe succeeded");
    size_t off = 0;
    for (int l = 0; l < nl; l++) {
      CcLevel &L = a.L[l];
      for (int d = 0; d < 4; d++) {
        if (!((L.commMask >> d) & 1)) continue;
        const size_t zone = (size_t)2 * L.n * L.faceCB[d] * sizeof(u32x4_t);
        for (int k = 0; k < 2; k++) {
          // zone k = 0: filled by the -d neighbour with its x_d = L-1 face; zone k = 1: by the +d neighbour with its x_d = 0 face
          const int face = k == 0 ? 1 : 0;          // which of MY faces goes into the neighbour's zone k
          const int slot = 2 * d + (face ? 1 : 0);  // face L-1 travels to the +d neighbour, face 0 to the -d neighbour
          for (int buf = 0; buf < kHaloBufs; buf++) {
            L.ghost[d][k][buf] = (u32x4_t *)(cc->window + off);
            L.peer[d][face][buf] = (u32x4_t *)((char *)cc->map->peer[slot] + off);
            off += zone;
          }
        }
      }
    }
    commBarrier();
  }
  if (g.size > 1) {
    const size_t rb = (size_t)2 * g.size * kRedMax * sizeof(u32x4_t);
    cc->redWindow = (char *)p2pAlloc(rb);
    HIP_CHECK(hipMemset(cc->redWindow, 0, rb));
    if (!commMapAllRanks(cc->redWindow, cc->redByRank, cc->redOpened)) errorQuda("peer mapping of the fused coarse-cycle sum window failed after the transport probe succeeded");
    a.redOwn = (u32x4_t *)cc->redWindow;
    for (int r = 0; r < g.size; r++) a.redPeer[r] = (u32x4_t *)cc->redByRank[r];
    commBarrier();
  }
  if (getVerbosity() >= QUDA_SUMMARIZE)
    printfQuda("MG level %d: levels %d..%d run as one persistent kernel (%d work-groups, %zu B dynamic LDS, %.1f MB work space%s)\n", tp.level + 1, tp.level + 1, tp.Nlevel, cc->grid,
               cc->ldsBytes, cc->workBytes * 1e-6, anyComm ? ", halo through peer windows" : "");
  return cc;
}

void coarseCycleDestroy(CoarseCycle *cc) {
  if (!cc) return;
  HIP_CHECK(hipStreamSynchronize(computeStream()));
  if (cc->timeline && cc->launches > 0) {
    // QUDA_AMD_MG_FUSED_TIMELINE=1: the last launch phase by phase (100 MHz wall clock at every barrier, seen by work-group 0)
    printfQuda("fused coarse cycle, last launch: start");
    for (int i = 1; i < 1024 && cc->timeline[i]; i++) printfQuda(" %.2f", 1e-2 * (double)(cc->timeline[i] - cc->timeline[i - 1]));
    printfQuda(" us\n");
  }
  if (cc->window || cc->redWindow) commBarrier();
  if (cc->map) { commUnmapPeers(*cc->map); delete cc->map; }
  for (void *p : cc->redOpened) (void)hipIpcCloseMemHandle(p);
  if (cc->window || cc->redWindow) commBarrier();
  if (cc->window) p2pFree(cc->window);
  if (cc->redWindow) p2pFree(cc->redWindow);
  if (cc->d_arg) (void)hipFree(cc->d_arg);
  if (cc->work) (void)hipFree(cc->work);
  if (cc->sync) (void)hipFree(cc->sync);
  if (cc->timeline) (void)hipHostFree(cc->timeline);
  delete cc;
}

bool coarseCycleApply(CoarseCycle *cc, ColorSpinorField &x, ColorSpinorField &b) {
  if (!cc) return false;
  CcLevel &L = cc->arg.L[0];
  if (x.SiteSubset() != QUDA_FULL_SITE_SUBSET || b.SiteSubset() != QUDA_FULL_SITE_SUBSET || x.Precision() != QUDA_SINGLE_PRECISION || b.Precision() != QUDA_SINGLE_PRECISION ||
      x.Nspin() != 2 || 2 * x.Ncolor() != L.n || x.VolumeCB() != L.Vh || b.VolumeCB() != L.Vh || x.V() == b.V())
    return false;
  float2 *bp[2] = {(float2 *)b.Even().V(), (float2 *)b.Odd().V()}, *xp[2] = {(float2 *)x.Even().V(), (float2 *)x.Odd().V()};
  CcVec &xo = cc->arg.xOut;
  if (cc->argDirty || L.b.p[0] != bp[0] || L.b.p[1] != bp[1] || xo.p[0] != xp[0] || xo.p[1] != xp[1] || L.b.stride != b.Stride() || xo.stride != x.Stride()) {
    L.b.p[0] = bp[0]; L.b.p[1] = bp[1]; L.b.stride = b.Stride();
    xo.p[0] = xp[0]; xo.p[1] = xp[1]; xo.stride = x.Stride();
    HIP_CHECK(hipMemcpyAsync(cc->d_arg, &cc->arg, sizeof(CcArg), hipMemcpyHostToDevice, computeStream()));
    cc->argDirty = false;
  }
  if (g_acctOn) {
    char tag[64];
    snprintf(tag, sizeof(tag), "fused cycle from %dx%dx%dx%d n %d, %d levels", L.Xc[0], L.Xc[1], L.Xc[2], L.Xc[3], L.n, cc->arg.nl);
    // links of the operator applications of the smoothers (per level: prepare 2, nu x 2, reconstruct 2 x 1, residual 1, guess 2) — the
    // coarsest GCR's share depends on its iteration count and is left out: a lower bound
    double bytes = 0;
    for (int l = 0; l < cc->arg.nl - 1; l++) {
      const CcLevel &Q = cc->arg.L[l];
      const double mat = (double)Q.n * Q.n * 8;
      bytes += (double)Q.Vh * mat * (8.0 * (2 * (Q.nuPre + Q.nuPost) + 2 + 3) + 2.0 * 9 + 4.0);
    }
    acct("coarse_cycle_kernel", bytes, tag);
  }
  switch (cc->n) {
    case 16: launchCycle<16>(*cc); break;
    case 32: launchCycle<32>(*cc); break;
    case 48: launchCycle<48>(*cc); break;
    default: launchCycle<64>(*cc); break;
  }
  HIP_CHECK(hipGetLastError());
  cc->launches++;
  return true;
}

void coarseCycleStats(const CoarseCycle *cc, long long out[5]) {
  for (int i = 0; i < 5; i++) out[i] = 0;
  if (!cc) return;
  unsigned st[8];
  HIP_CHECK(hipStreamSynchronize(computeStream()));
  HIP_CHECK(hipMemcpy(st, cc->arg.state, sizeof(st), hipMemcpyDeviceToHost));
  out[0] = st[2]; out[1] = st[3]; out[2] = st[4]; out[3] = st[5]; out[4] = cc->grid;
}

}  // namespace quda
