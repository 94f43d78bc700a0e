// blas.hip — fused BLAS-1 updates and reductions on spinor fields (gfx950).
//
// One templated grid-stride kernel; the functor says which of x,y,z,w it reads / writes and how many
// double-precision partial sums it produces.  Every operand is streamed once with 16-byte per-lane
// accesses (HBM-bound: (n_in + n_out) x field bytes); sums are accumulated in fp64 regardless of the
// storage precision (reference QudaSumFloat, lib/reduce_quda.cu:61-63), reduced by wave shuffles
// (64 lanes) -> LDS -> one fp64 partial per block; the last block to finish (completion counter) adds the partials in block
// order — bit-reproducible, unlike fp64 atomics — and writes the result straight into pinned host memory (single rank) or
// into the device word RCCL all-reduces.  No memset before and no copy after the kernel: a reduction is ONE launch + one
// stream synchronisation (the memset/copy pair cost ~7 us per reduction and 35 ms of the 16^4 MG setup+solve profile).
// fp64/fp32 fields of any spin/colour are treated as flat arrays (complex pairs stay adjacent in the
// FLOAT2/FLOAT4 planar orders); 16-bit fields go site by site because of their per-site scale.
#include <cstring>
#include <sys/time.h>
#include <unistd.h>
#include "blas.h"

#include "device_io.h"
#include "halo.h"
#include "p2p.h"

namespace quda {
namespace blas {

unsigned long long flops = 0;
unsigned long long bytes = 0;

static double *d_red = nullptr;   // device results (what RCCL all-reduces)
static double *h_red = nullptr;   // pinned, device-mapped host results
static double *h_red_dev = nullptr;  // device address of h_red
static double *d_part = nullptr;  // per-block partial sums
static unsigned *d_count = nullptr;  // completion counter (zero between launches)
constexpr int kMaxBlocks = 4096;
constexpr int kMaxRed = 64;
constexpr int kSlotDoubles = 48;  // doubles per (buffer, rank) slot of the all-reduce window (the blocked GCR orthogonalisation reduces 2 k + 3 <= 43 sums at once)

// Block-level all-reduce over ranks through the peer windows (called by ONE block per rank, all threads of it).  Thread r
// delivers this rank's nred sums to rank r with system-scope write-through stores followed by a fire-and-forget counter
// bump; thread 0 then waits — for at most waitTicks — until every rank's contribution has arrived in the own window, and
// thread k < nred returns sum_r slot[r][k], added in rank order on every rank, so all ranks hold bit-identical results.
// *done = 0 if the wait ran out: ranks may be arbitrarily far apart when they reach a collective (one of them still busy on
// its host), which is not an error — the caller then finishes the reduction from the host side (finishAllreduceOnHost);
// this rank's own contribution has been delivered either way.
template <int NRED>
__device__ __forceinline__ double peer_allreduce(const double *mine, double *const *peerSlots, unsigned *const *peerCount, int nranks, int rank, int buf,
                                                 unsigned expect, unsigned long long waitTicks, int *done) {
  if ((int)threadIdx.x < nranks) {
    double *slot = peerSlots[threadIdx.x] + ((size_t)buf * nranks + rank) * kSlotDoubles;
    for (int k = 0; k < NRED; k++)
      __hip_atomic_store(reinterpret_cast<unsigned long long *>(slot + k), __builtin_bit_cast(unsigned long long, mine[k]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    // the slot stores are write-through (sc0 sc1) and must have been acknowledged before the counter can announce them: a
    // workgroup-scope fence emits nothing here, so drain this wave's stores explicitly (inline asm: the compiler may drop a
    // builtin wait it believes redundant), then bump the peer's counter.  The add itself stays relaxed: a release at system scope
    // would add a write-back of the whole dirty L2 (buffer_wbl2), which orders nothing these write-through stores need
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    (void)__hip_atomic_fetch_add(peerCount[threadIdx.x] + buf, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  if (threadIdx.x == 0) {
    const unsigned long long t0 = wall_clock64();
    int ok = 1;
    while ((int)(__hip_atomic_load(peerCount[rank] + buf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - expect) < 0) {
      if (wall_clock64() - t0 > waitTicks) { ok = 0; break; }
      __builtin_amdgcn_s_sleep(1);
    }
    // one acquire after the poll (not inside it): nothing read below may come from a line cached before the counter matched
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    *done = ok;
  }
  __syncthreads();
  double v = 0;
  if (threadIdx.x < NRED && *done) {
    const double *slots = peerSlots[rank] + (size_t)buf * nranks * kSlotDoubles;
    for (int r = 0; r < nranks; r++)
      v += __builtin_bit_cast(double, __hip_atomic_load(reinterpret_cast<const unsigned long long *>(slots + (size_t)r * kSlotDoubles + threadIdx.x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
  }
  return v;
}

// start-up probe: one all-reduce of known values per round; result[0] = 1 if every sum came out right and in time
__global__ void allreduce_probe_kernel(double *const *peerSlots, unsigned *const *peerCount, int nranks, int rank, int buf, unsigned expect,
                                       unsigned long long waitTicks, int round, int *result) {
  __shared__ double mine[kSlotDoubles];
  __shared__ int done;
  if (threadIdx.x < kSlotDoubles) mine[threadIdx.x] = (double)((rank + 1) * (round + 1) + 1000 * (int)threadIdx.x);
  __syncthreads();
  const double v = peer_allreduce<kSlotDoubles>(mine, peerSlots, peerCount, nranks, rank, buf, expect, waitTicks, &done);
  bool ok = true;
  if (threadIdx.x < kSlotDoubles) ok = v == (double)((round + 1) * (nranks * (nranks + 1) / 2) + 1000 * (int)threadIdx.x * nranks);
  const int allOk = __syncthreads_and(ok);
  if (threadIdx.x == 0) result[0] = allOk && done;
}

// ---- all-reduce window: [2 buffers][ranks][kSlotDoubles] doubles + 2 counters, fine-grained, mapped into every rank ----
struct ReduceWindow {
  char *window = nullptr;
  double **d_slots = nullptr;      // device tables [rank]
  unsigned **d_counts = nullptr;
  std::vector<void *> opened;
  unsigned long seq = 0;
  unsigned uses[2] = {0, 0};
  int state = -1;                  // -1 not decided, 0 unavailable (RCCL all-reduce), 1 active
};
static ReduceWindow g_rw;

static void releaseReduceWindow() {
  if (g_rw.window) {
    HIP_CHECK(hipDeviceSynchronize());
    commBarrier();
    for (void *p : g_rw.opened) (void)hipIpcCloseMemHandle(p);
    commBarrier();
    p2pFree(g_rw.window);
    (void)hipFree(g_rw.d_slots);
    (void)hipFree(g_rw.d_counts);
  }
  g_rw = ReduceWindow();
}

// collective (first global reduction of a multi-rank run)
static bool reduceWindowActive() {
  if (g_rw.state >= 0) return g_rw.state == 1;
  const CommGrid &g = commGrid();
  g_rw.state = 0;
  // QUDA_AMD_ALLREDUCE=rccl keeps global sums on the collective library; default: peer stores wherever the halo uses them
  const char *e = getenv("QUDA_AMD_ALLREDUCE");
  if (e && !strcmp(e, "rccl")) return false;
  if (e && strcmp(e, "p2p")) errorQuda("QUDA_AMD_ALLREDUCE=%s: expected rccl or p2p", e);
  if (g.size < 2 || g.size > 64 || !p2pHaloEnabled()) return false;
  const size_t slotBytes = (size_t)2 * g.size * kSlotDoubles * sizeof(double);
  g_rw.window = (char *)p2pAlloc(slotBytes + 256);
  std::vector<void *> byRank;
  if (!commMapAllRanks(g_rw.window, byRank, g_rw.opened)) { p2pFree(g_rw.window); g_rw.window = nullptr; return false; }
  std::vector<double *> hs(g.size);
  std::vector<unsigned *> hc(g.size);
  for (int r = 0; r < g.size; r++) { hs[r] = (double *)byRank[r]; hc[r] = (unsigned *)((char *)byRank[r] + slotBytes); }
  HIP_CHECK(qaMalloc((void **)&g_rw.d_slots, g.size * sizeof(double *)));
  HIP_CHECK(qaMalloc((void **)&g_rw.d_counts, g.size * sizeof(unsigned *)));
  HIP_CHECK(hipMemcpy(g_rw.d_slots, hs.data(), g.size * sizeof(double *), hipMemcpyHostToDevice));
  HIP_CHECK(hipMemcpy(g_rw.d_counts, hc.data(), g.size * sizeof(unsigned *), hipMemcpyHostToDevice));
  HIP_CHECK(hipDeviceSynchronize());
  commBarrier();
  // probe: three all-reduces of known values with the production kernel code; any rank that sees a wrong sum or a
  // timeout vetoes, and then every rank drops back to the RCCL all-reduce
  int *d_res = nullptr;
  HIP_CHECK(qaMalloc((void **)&d_res, sizeof(int)));
  double fail = 0;
  for (int round = 0; round < 3; round++) {
    const int buf = (int)(++g_rw.seq & 1);
    const unsigned expect = (g_rw.uses[buf] += (unsigned)g.size);
    hipLaunchKernelGGL(allreduce_probe_kernel, dim3(1), dim3(64), 0, computeStream(), (double *const *)g_rw.d_slots, (unsigned *const *)g_rw.d_counts, g.size, g.rank,
                       buf, expect, (unsigned long long)3e8, round, d_res);
    int res = 0;
    HIP_CHECK(hipMemcpyAsync(&res, d_res, sizeof(int), hipMemcpyDeviceToHost, computeStream()));
    HIP_CHECK(hipStreamSynchronize(computeStream()));
    if (!res) fail = 1;
  }
  (void)hipFree(d_res);
  comm_allreduce(&fail, 1);
  if (fail != 0) {
    if (g.rank == 0) warningQuda("peer-store all-reduce probe failed on some rank: global sums use the collective library");
    releaseReduceWindow();
    g_rw.state = 0;
    return false;
  }
  g_rw.state = 1;
  return true;
}
// The in-kernel wait of a peer all-reduce ran out (another rank has not reached this reduction yet — it may be busy on its
// host for as long as it likes): wait for the missing contributions from the host side, with the patience of a blocking
// collective (QUDA_AMD_COLLECTIVE_TIMEOUT_S, default 600 s), then add the slots in rank order exactly as the kernel would.
static void finishAllreduceOnHost(int buf, unsigned expect, int nred, double *out) {
  const CommGrid &g = commGrid();
  const size_t slotBytes = (size_t)2 * g.size * kSlotDoubles * sizeof(double);
  const unsigned *d_count = (const unsigned *)(g_rw.window + slotBytes) + buf;
  static double limit = -1;
  if (limit < 0) { const char *e = getenv("QUDA_AMD_COLLECTIVE_TIMEOUT_S"); limit = e ? atof(e) : 600.0; }
  timeval t0; gettimeofday(&t0, nullptr);
  for (long it = 0;; it++) {
    unsigned cnt = 0;
    HIP_CHECK(hipMemcpy(&cnt, d_count, sizeof(unsigned), hipMemcpyDeviceToHost));
    if ((int)(cnt - expect) >= 0) break;
    timeval t1; gettimeofday(&t1, nullptr);
    if ((t1.tv_sec - t0.tv_sec) + 1e-6 * (t1.tv_usec - t0.tv_usec) > limit)
      errorQuda("all-reduce: %u of %u contributions after %.0f s (a rank never reached this reduction)", cnt - (expect - (unsigned)g.size), (unsigned)g.size, limit);
    usleep(it < 1000 ? 20 : 500);
  }
  std::vector<double> slots((size_t)g.size * kSlotDoubles);
  HIP_CHECK(hipMemcpy(slots.data(), g_rw.window + (size_t)buf * g.size * kSlotDoubles * sizeof(double), slots.size() * sizeof(double), hipMemcpyDeviceToHost));
  for (int k = 0; k < nred; k++) {
    double v = 0;
    for (int r = 0; r < g.size; r++) v += slots[(size_t)r * kSlotDoubles + k];
    out[k] = v;
  }
}
static bool g_global_reduction = true;

void init() {
  if (!d_red) HIP_CHECK(qaMalloc((void **)&d_red, kMaxRed * sizeof(double)));
  if (!h_red) {
    HIP_CHECK(hipHostMalloc((void **)&h_red, kMaxRed * sizeof(double), hipHostMallocMapped));
    HIP_CHECK(hipHostGetDevicePointer((void **)&h_red_dev, h_red, 0));
  }
  if (!d_part) HIP_CHECK(qaMalloc((void **)&d_part, (size_t)kMaxBlocks * kSlotDoubles * sizeof(double)));
  if (!d_count) {
    HIP_CHECK(qaMalloc((void **)&d_count, sizeof(unsigned)));
    HIP_CHECK(hipMemset(d_count, 0, sizeof(unsigned)));
    HIP_CHECK(hipDeviceSynchronize());
  }
}
void end() {
  releaseReduceWindow();
  if (d_red) (void)hipFree(d_red);
  if (h_red) (void)hipHostFree(h_red);
  if (d_part) (void)hipFree(d_part);
  if (d_count) (void)hipFree(d_count);
  d_red = h_red = h_red_dev = d_part = nullptr;
  d_count = nullptr;
}
void setGlobalReduction(bool on) { g_global_reduction = on; }
bool globalReduction() { return g_global_reduction; }

struct Seg {           // one contiguous parity block of a field
  void *v[2];
  float *norm[2];
};

template <typename F> struct BlasArg {
  Seg x, y, z, w;
  int nseg;
  long n;        // chunks per segment
  int stride;    // site path: plane stride
  F f;
  double *red;       // device result
  double *hred;      // pinned host result (nullptr: an all-reduce follows, the host reads the device word afterwards)
  double *part;      // [block][nred] partial sums
  unsigned *count;   // completion counter
  // all-reduce through peer-mapped windows (p2p.h), done by the last block: every rank stores its sums into slot [buf][rank]
  // of every rank's window, bumps that rank's counter, waits for its own counter and adds the slots in rank order —
  // bit-identical on all ranks, no collective-library launch, no device-to-host copy
  double *const *peerSlots;     // [rank] -> that rank's window (nullptr table: no peer all-reduce)
  unsigned *const *peerCount;   // [rank] -> that rank's counters
  int nranks, rank, buf;
  unsigned expect;
  unsigned long long waitTicks;
};

template <typename real, int M> struct alignas(16) Chunk { real v[M]; };

// ---- functors: operate on one chunk (M reals, complex pairs adjacent) ----
#define QA_FLAGS(RX, RY, RZ, RW, WX, WY, WZ, WW, NRED)                                        \
  static constexpr bool rx = RX, ry = RY, rz = RZ, rw = RW, wx = WX, wy = WY, wz = WZ, ww = WW; \
  static constexpr int nred = NRED;

struct Norm2F { QA_FLAGS(1, 0, 0, 0, 0, 0, 0, 0, 1)
  template <typename real, int M> __device__ void operator()(real *x, real *, real *, real *, double *r) const {
#pragma unroll
    for (int i = 0; i < M; i++) r[0] += (double)x[i] * (double)x[i];
  } };
struct ReDotF { QA_FLAGS(1, 1, 0, 0, 0, 0, 0, 0, 1)
  template <typename real, int M> __device__ void operator()(real *x, real *y, real *, real *, double *r) const {
#pragma unroll
    for (int i = 0; i < M; i++) r[0] += (double)x[i] * (double)y[i];
  } };
// r0 + i r1 = conj(x) y ; r2 = |x|^2 (NORM=1) or |y|^2 (NORM=2)
template <int NORM> struct CDotF { QA_FLAGS(1, 1, 0, 0, 0, 0, 0, 0, (NORM ? 3 : 2))
  template <typename real, int M> __device__ void operator()(real *x, real *y, real *, real *, double *r) const {
#pragma unroll
    for (int i = 0; i < M; i += 2) {
      const double xr = x[i], xi = x[i + 1], yr = y[i], yi = y[i + 1];
      r[0] += xr * yr + xi * yi;
      r[1] += xr * yi - xi * yr;
      if (NORM == 1) r[2] += xr * xr + xi * xi;
      if (NORM == 2) r[2] += yr * yr + yi * yi;
    }
  } };
struct AxF { double a; QA_FLAGS(1, 0, 0, 0, 1, 0, 0, 0, 0)
  template <typename real, int M> __device__ void operator()(real *x, real *, real *, real *, double *) const {
#pragma unroll
    for (int i = 0; i < M; i++) x[i] *= (real)a;
  } };
// y = a x + b y (+ |y|^2 if NRM)
template <int NRM> struct AxpbyF { double a, b; QA_FLAGS(1, 1, 0, 0, 0, 1, 0, 0, NRM)
  template <typename real, int M> __device__ void operator()(real *x, real *y, real *, real *, double *r) const {
#pragma unroll
    for (int i = 0; i < M; i++) { y[i] = (real)a * x[i] + (real)b * y[i]; if (NRM) r[0] += (double)y[i] * (double)y[i]; }
  } };
// y = a x + b y complex (+ |y|^2)
template <int NRM> struct CaxpbyF { double ar, ai, br, bi; QA_FLAGS(1, 1, 0, 0, 0, 1, 0, 0, NRM)
  template <typename real, int M> __device__ void operator()(real *x, real *y, real *, real *, double *r) const {
#pragma unroll
    for (int i = 0; i < M; i += 2) {
      const real xr = x[i], xi = x[i + 1], yr = y[i], yi = y[i + 1];
      y[i] = (real)ar * xr - (real)ai * xi + (real)br * yr - (real)bi * yi;
      y[i + 1] = (real)ar * xi + (real)ai * xr + (real)br * yi + (real)bi * yr;
      if (NRM) r[0] += (double)y[i] * (double)y[i] + (double)y[i + 1] * (double)y[i + 1];
    }
  } };
// z = x - y   (z is not read: it may hold anything)
struct XmyzF { QA_FLAGS(1, 1, 0, 0, 0, 0, 1, 0, 0)
  template <typename real, int M> __device__ void operator()(real *x, real *y, real *z, real *, double *) const {
#pragma unroll
    for (int i = 0; i < M; i++) z[i] = x[i] - y[i];
  } };
// z = x + a y + b z
struct CxpaypbzF { double ar, ai, br, bi; QA_FLAGS(1, 1, 1, 0, 0, 0, 1, 0, 0)
  template <typename real, int M> __device__ void operator()(real *x, real *y, real *z, real *, double *) const {
#pragma unroll
    for (int i = 0; i < M; i += 2) {
      const real yr = y[i], yi = y[i + 1], zr = z[i], zi = z[i + 1];
      z[i] = x[i] + (real)ar * yr - (real)ai * yi + (real)br * zr - (real)bi * zi;
      z[i + 1] = x[i + 1] + (real)ar * yi + (real)ai * yr + (real)br * zi + (real)bi * zr;
    }
  } };
// y += a x ; x -= a z  (+ |x|^2)
template <int NRM> struct CaxpyXmazF { double ar, ai; QA_FLAGS(1, 1, 1, 0, 1, 1, 0, 0, NRM)
  template <typename real, int M> __device__ void operator()(real *x, real *y, real *z, real *, double *r) const {
#pragma unroll
    for (int i = 0; i < M; i += 2) {
      const real xr = x[i], xi = x[i + 1], zr = z[i], zi = z[i + 1];
      y[i] += (real)ar * xr - (real)ai * xi;
      y[i + 1] += (real)ar * xi + (real)ai * xr;
      x[i] = xr - ((real)ar * zr - (real)ai * zi);
      x[i + 1] = xi - ((real)ar * zi + (real)ai * zr);
      if (NRM) r[0] += (double)x[i] * (double)x[i] + (double)x[i + 1] * (double)x[i + 1];
    }
  } };
// y = a x ; x -= a z      (first iteration of MR from a zero start: y is not read)
struct CaxXmazF { double ar, ai; QA_FLAGS(1, 0, 1, 0, 1, 1, 0, 0, 0)
  template <typename real, int M> __device__ void operator()(real *x, real *y, real *z, real *, double *) const {
#pragma unroll
    for (int i = 0; i < M; i += 2) {
      const real xr = x[i], xi = x[i + 1], zr = z[i], zi = z[i + 1];
      y[i] = (real)ar * xr - (real)ai * xi;
      y[i + 1] = (real)ar * xi + (real)ai * xr;
      x[i] = xr - ((real)ar * zr - (real)ai * zi);
      x[i + 1] = xi - ((real)ar * zi + (real)ai * zr);
    }
  } };
// y = a x ; w = x - a z    (the same with the residual still in the source field x, which stays untouched)
struct CaxInitF { double ar, ai; QA_FLAGS(1, 0, 1, 0, 0, 1, 0, 1, 0)
  template <typename real, int M> __device__ void operator()(real *x, real *y, real *z, real *w, double *) const {
#pragma unroll
    for (int i = 0; i < M; i += 2) {
      const real xr = x[i], xi = x[i + 1], zr = z[i], zi = z[i + 1];
      y[i] = (real)ar * xr - (real)ai * xi;
      y[i + 1] = (real)ar * xi + (real)ai * xr;
      w[i] = xr - ((real)ar * zr - (real)ai * zi);
      w[i + 1] = xi - ((real)ar * zi + (real)ai * zr);
    }
  } };
// x = a x ; y += b x (+ |y|^2)
template <int NRM> struct CabxpyAxF { double a, br, bi; QA_FLAGS(1, 1, 0, 0, 1, 1, 0, 0, NRM)
  template <typename real, int M> __device__ void operator()(real *x, real *y, real *, real *, double *r) const {
#pragma unroll
    for (int i = 0; i < M; i += 2) {
      const real xr = (real)a * x[i], xi = (real)a * x[i + 1];
      x[i] = xr; x[i + 1] = xi;
      y[i] += (real)br * xr - (real)bi * xi;
      y[i + 1] += (real)br * xi + (real)bi * xr;
      if (NRM) r[0] += (double)y[i] * (double)y[i] + (double)y[i + 1] * (double)y[i + 1];
    }
  } };
// y += a x ; (z, y)
struct CaxpyDotzyF { double ar, ai; QA_FLAGS(1, 1, 1, 0, 0, 1, 0, 0, 2)
  template <typename real, int M> __device__ void operator()(real *x, real *y, real *z, real *, double *r) const {
#pragma unroll
    for (int i = 0; i < M; i += 2) {
      y[i] += (real)ar * x[i] - (real)ai * x[i + 1];
      y[i + 1] += (real)ar * x[i + 1] + (real)ai * x[i];
      const double zr = z[i], zi = z[i + 1], yr = y[i], yi = y[i + 1];
      r[0] += zr * yr + zi * yi;
      r[1] += zr * yi - zi * yr;
    }
  } };
// z += a x + b y ; y -= b w
struct CaxpbypzYmbwF { double ar, ai, br, bi; QA_FLAGS(1, 1, 1, 1, 0, 1, 1, 0, 0)
  template <typename real, int M> __device__ void operator()(real *x, real *y, real *z, real *w, double *) const {
#pragma unroll
    for (int i = 0; i < M; i += 2) {
      const real yr = y[i], yi = y[i + 1];
      z[i] += (real)ar * x[i] - (real)ai * x[i + 1] + (real)br * yr - (real)bi * yi;
      z[i + 1] += (real)ar * x[i + 1] + (real)ai * x[i] + (real)br * yi + (real)bi * yr;
      y[i] = yr - ((real)br * w[i] - (real)bi * w[i + 1]);
      y[i + 1] = yi - ((real)br * w[i + 1] + (real)bi * w[i]);
    }
  } };

// ---- the tail of every reducing kernel: wave shuffles -> LDS -> per-block partial -> the LAST block (completion counter) adds the
// partials in block order (bit-reproducible), optionally does the all-reduce over ranks through the peer windows, and writes the
// sums to device and pinned host memory.  NRED sums per thread come in, every thread of the block must call it. ----
struct RedCtl {
  double *red, *hred, *part;
  unsigned *count;
  double *const *peerSlots;
  unsigned *const *peerCount;
  int nranks, rank, buf;
  unsigned expect;
  unsigned long long waitTicks;
};
template <int NRED> __device__ __forceinline__ void finish_reduction(const double *red, const RedCtl &arg) {
  __shared__ double lds[4][NRED];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < NRED; k++) {
    double v = red[k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if (lane == 0) lds[wave][k] = v;
  }
  __syncthreads();
  // block partial -> memory (agent-scope write-through stores: no L2 write-back needed before the counter is bumped)
  if (threadIdx.x < NRED) {
    double v = 0;
    for (int wv = 0; wv < (int)(blockDim.x >> 6); wv++) v += lds[wv][threadIdx.x];
    __hip_atomic_store(&arg.part[(size_t)blockIdx.x * NRED + threadIdx.x], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __shared__ int isLast;
  // the partials are sc1 (write-through) stores; the storing wave waits for their acknowledgement BEFORE the barrier, so the
  // counter bump below cannot overtake them on their way to memory (the last block may sit on another XCD with its own L2);
  // it then reads them with sc1 loads behind its own barrier — the drained-sc1 hand-off of MI355X_MICROARCH.md
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) isLast = __hip_atomic_fetch_add(arg.count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
  __syncthreads();
  if (isLast) {
    // the last block adds the partials in a fixed order: thread t takes blocks t, t + blockDim, ...; then the block tree
#pragma unroll 1
    for (int k = 0; k < NRED; k++) {
      double v = 0;
      for (int b = threadIdx.x; b < (int)gridDim.x; b += blockDim.x) v += __hip_atomic_load(&arg.part[(size_t)b * NRED + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
      __syncthreads();
      if (lane == 0) lds[wave][k] = v;
    }
    __syncthreads();
    if (arg.peerSlots) {
      __shared__ double mine[NRED];
      if (threadIdx.x < NRED) {
        double v = 0;
        for (int wv = 0; wv < (int)(blockDim.x >> 6); wv++) v += lds[wv][threadIdx.x];
        mine[threadIdx.x] = v;
      }
      __syncthreads();
      __shared__ int done;
      const double v = peer_allreduce<NRED>(mine, arg.peerSlots, arg.peerCount, arg.nranks, arg.rank, arg.buf, arg.expect, arg.waitTicks, &done);
      if (threadIdx.x < NRED) {
        arg.red[threadIdx.x] = v;
        __hip_atomic_store(&arg.hred[threadIdx.x], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      // last word of the host buffer: 1 = the sums above are the global ones, 0 = the host has to finish the reduction
      if (threadIdx.x == 0) __hip_atomic_store(&arg.hred[kMaxRed - 1], done ? 1.0 : 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    } else if (threadIdx.x < NRED) {
      double v = 0;
      for (int wv = 0; wv < (int)(blockDim.x >> 6); wv++) v += lds[wv][threadIdx.x];
      arg.red[threadIdx.x] = v;
      if (arg.hred) __hip_atomic_store(&arg.hred[threadIdx.x], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (threadIdx.x == 0) __hip_atomic_store(arg.count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// ---- device-side scalars: the coefficient of an update comes from the sums the PREVIOUS kernel of the stream left in device memory, so a fixed-length
// iteration (the MR smoother: alpha = omega (Ar, r) / |Ar|^2, lib/inv_mr_quda.cpp:60-120) runs without a host round trip per step — 10-15 us of idle
// device per step, which is a quarter of a multigrid cycle once the kernels are 8 x shorter on an 8-GPU sub-lattice.  Rank-local sums only
// (the smoothers' global_reduction = false): a global sum goes through the host or the peer windows. ----
template <typename Base> struct DevAlpha : Base {
  const double *dres; double omega;
  __device__ __forceinline__ void prepare() {
    const double z = dres[2];                 // (re, im, |Ar|^2) of CDotF<1>
    const double sc = z > 0.0 ? omega / z : 0.0;   // zero source or breakdown: nothing to add (the host loop breaks there)
    this->ar = sc * dres[0]; this->ai = sc * dres[1];
  }
};
template <typename F> __device__ __forceinline__ auto blas_prepare(F &f, int) -> decltype(f.prepare(), void()) { f.prepare(); }
template <typename F> __device__ __forceinline__ void blas_prepare(F &, long) {}

// ---- kernel ----
template <typename T, int M, bool SITE, typename F>
__global__ void __launch_bounds__(256) blas_kernel(BlasArg<F> arg) {
  using real = typename Store<T>::real;
  double red[F::nred > 0 ? F::nred : 1];
#pragma unroll
  for (int k = 0; k < (F::nred > 0 ? F::nred : 1); k++) red[k] = 0.0;
  const long total = arg.n * arg.nseg;
  F f = arg.f;
  blas_prepare(f, 0);
  if constexpr (!SITE) {
    // Four chunks per thread and trip, all their loads requested before the first is used: a streaming kernel needs ~64 KB in flight per CU
    // to cover the HBM latency at 8 TB/s (Little), and two 16-byte loads per lane in 8 waves per CU are 16 KB — the one-chunk loop ran
    // axpy at 4.65 TB/s where the copy rate of the device is 6.3.  The loads cannot be hoisted by the compiler (the stores of a chunk may
    // alias the next chunk's loads for all it knows), hence by hand.
    constexpr int UN = 4;
    using V = Chunk<real, M>;
    const long step = (long)gridDim.x * blockDim.x;
    for (long i0 = blockIdx.x * (long)blockDim.x + threadIdx.x; i0 < total; i0 += step * UN) {
      V xv[UN], yv[UN], zv[UN], wv[UN];
      int seg[UN]; long jj[UN]; bool ok[UN];
#pragma unroll
      for (int u = 0; u < UN; u++) {
        const long i = i0 + u * step;
        ok[u] = i < total;
        const long ic = ok[u] ? i : i0;
        seg[u] = ic >= arg.n ? 1 : 0;
        jj[u] = ic - seg[u] * arg.n;
        if (F::rx) xv[u] = reinterpret_cast<const V *>(arg.x.v[seg[u]])[jj[u]];
        if (F::ry) yv[u] = reinterpret_cast<const V *>(arg.y.v[seg[u]])[jj[u]];
        if (F::rz) zv[u] = reinterpret_cast<const V *>(arg.z.v[seg[u]])[jj[u]];
        if (F::rw) wv[u] = reinterpret_cast<const V *>(arg.w.v[seg[u]])[jj[u]];
      }
#pragma unroll
      for (int u = 0; u < UN; u++) {
        if (!ok[u]) continue;
        f.template operator()<real, M>(xv[u].v, yv[u].v, zv[u].v, wv[u].v, red);
        if (F::wx) reinterpret_cast<V *>(arg.x.v[seg[u]])[jj[u]] = xv[u];
        if (F::wy) reinterpret_cast<V *>(arg.y.v[seg[u]])[jj[u]] = yv[u];
        if (F::wz) reinterpret_cast<V *>(arg.z.v[seg[u]])[jj[u]] = zv[u];
        if (F::ww) reinterpret_cast<V *>(arg.w.v[seg[u]])[jj[u]] = wv[u];
      }
    }
  } else {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int seg = i >= arg.n ? 1 : 0;
    const long j = i - seg * arg.n;
    alignas(16) real x[M];
    alignas(16) real y[M];
    alignas(16) real z[M];
    alignas(16) real w[M];
    if (F::rx) Planar<T, M>::load(x, arg.x.v[seg], arg.stride, (int)j, arg.x.norm[seg], (int)j);
    if (F::ry) Planar<T, M>::load(y, arg.y.v[seg], arg.stride, (int)j, arg.y.norm[seg], (int)j);
    if (F::rz) Planar<T, M>::load(z, arg.z.v[seg], arg.stride, (int)j, arg.z.norm[seg], (int)j);
    if (F::rw) Planar<T, M>::load(w, arg.w.v[seg], arg.stride, (int)j, arg.w.norm[seg], (int)j);
    f.template operator()<real, M>(x, y, z, w, red);
    if (F::wx) Planar<T, M>::store(x, arg.x.v[seg], arg.stride, (int)j, arg.x.norm[seg], (int)j);
    if (F::wy) Planar<T, M>::store(y, arg.y.v[seg], arg.stride, (int)j, arg.y.norm[seg], (int)j);
    if (F::wz) Planar<T, M>::store(z, arg.z.v[seg], arg.stride, (int)j, arg.z.norm[seg], (int)j);
    if (F::ww) Planar<T, M>::store(w, arg.w.v[seg], arg.stride, (int)j, arg.w.norm[seg], (int)j);
  }
  }
  if (F::nred > 0) {
    RedCtl c;
    c.red = arg.red; c.hred = arg.hred; c.part = arg.part; c.count = arg.count; c.peerSlots = arg.peerSlots; c.peerCount = arg.peerCount;
    c.nranks = arg.nranks; c.rank = arg.rank; c.buf = arg.buf; c.expect = arg.expect; c.waitTicks = arg.waitTicks;
    finish_reduction<(F::nred > 0 ? F::nred : 1)>(red, c);
  }
}

// ================================================================================================
// Blocked orthogonalisation of GCR (reference lib/inv_gcr_quda.cpp:53-84, :103-121: its pipelined forms compute N dots in one pass
// and apply N caxpys in one pass).  Iteration k of GCR has to orthogonalise Ap_k against the k normalised directions before it,
// normalise it and update the residual.  One direction after the other (modified Gram-Schmidt, solver.cpp orthoDir) that is
// 4 (k - 1) + 9 field passes; here it is TWO kernels:
//   multi_dot_kernel      beta_i = (Ap_i, Ap_k) for all i < k, (Ap_k, r) and |Ap_k|^2 in one sweep           (k + 2 reads)
//   multi_caxpy_kernel    Ap_k <- (Ap_k - sum_i beta_i Ap_i) / gamma ; r <- r - alpha Ap_k ; |r|^2, |Ap_k|^2   (k + 2 reads, 2 writes)
// with gamma^2 = |Ap_k|^2 - sum |beta_i|^2 and alpha = (Ap_k, r) / gamma between them on the host (classical Gram-Schmidt; the caller
// falls back to the sequential form when gamma^2 is a small difference of large numbers).  The same second kernel, without the
// residual part, adds the k search directions to the solution at a restart (updateSolution).  k <= kMaxDirs, fp64 / fp32 fields.
// ================================================================================================
constexpr int kMaxDirs = 20;
struct MultiArg {
  const void *f[kMaxDirs][2];   // the k fields [i][segment]
  void *y[2];                   // Ap_k (dots: read; caxpy: read-modify-write) or the solution
  void *r[2];                   // residual (may be null for the plain multi-caxpy)
  double cr[kMaxDirs], ci[kMaxDirs];   // coefficients of the fields
  double scale, ar, ai;         // y <- scale (y + sum c_i f_i) ; r <- r - (ar + i ai) y
  int k, nseg;
  long n;
  RedCtl c;
};
// KB: compile-time bucket >= k (4, 8, 12, 16, 20): the loads of all KB fields of a chunk are issued back to back, without a branch between
// them (a per-field `if (d < k)` made every load its own basic block: load, wait, use — twenty dependent latencies per chunk); the
// surplus slots point at field 0 again and carry a zero coefficient / an ignored sum
template <typename real, int M, int KB> __global__ void __launch_bounds__(256) multi_dot_kernel(const MultiArg arg) {
  double red[2 * KB + 3];
#pragma unroll
  for (int q = 0; q < 2 * KB + 3; q++) red[q] = 0.0;
  using V = Chunk<real, M>;
  const long total = arg.n * arg.nseg;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int seg = i >= arg.n ? 1 : 0;
    const long j = i - seg * arg.n;
    V y, r, f[KB];
    y = reinterpret_cast<const V *>(arg.y[seg])[j];
    r = reinterpret_cast<const V *>(arg.r[seg])[j];
#pragma unroll
    for (int d = 0; d < KB; d++) f[d] = reinterpret_cast<const V *>(arg.f[d][seg])[j];
#pragma unroll
    for (int e = 0; e < M; e += 2) {
      const double yr = y.v[e], yi = y.v[e + 1], rr = r.v[e], ri = r.v[e + 1];
      red[2 * KB] += yr * rr + yi * ri;        // conj(y) r
      red[2 * KB + 1] += yr * ri - yi * rr;
      red[2 * KB + 2] += yr * yr + yi * yi;
#pragma unroll
      for (int d = 0; d < KB; d++) {
        const double fr = f[d].v[e], fi = f[d].v[e + 1];
        red[2 * d] += fr * yr + fi * yi;             // conj(f) y
        red[2 * d + 1] += fr * yi - fi * yr;
      }
    }
  }
  // the bucket's own 2 KB + 3 sums, not the 43 of the largest one: the last block adds the partials of every sum over all blocks, which at 43 sums
  // and 1024 blocks was 65 of the 82 us this kernel took on a 12.6 MB field (the 32 x 16 x 16 x 16 sub-lattice of an 8-GPU split)
  finish_reduction<2 * KB + 3>(red, arg.c);
}
template <typename real, int M, bool RES, int KB> __global__ void __launch_bounds__(256) multi_caxpy_kernel(const MultiArg arg) {
  double red[2] = {0.0, 0.0};
  using V = Chunk<real, M>;
  const long total = arg.n * arg.nseg;
  real cr[KB], ci[KB];
#pragma unroll
  for (int d = 0; d < KB; d++) { cr[d] = (real)arg.cr[d]; ci[d] = (real)arg.ci[d]; }
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int seg = i >= arg.n ? 1 : 0;
    const long j = i - seg * arg.n;
    V y, r, f[KB];
    y = reinterpret_cast<const V *>(arg.y[seg])[j];
    if (RES) r = reinterpret_cast<const V *>(arg.r[seg])[j];
#pragma unroll
    for (int d = 0; d < KB; d++) f[d] = reinterpret_cast<const V *>(arg.f[d][seg])[j];
    // the sum in the field's own precision, as the one-at-a-time caxpys have it
#pragma unroll
    for (int d = 0; d < KB; d++)
#pragma unroll
      for (int e = 0; e < M; e += 2) {
        y.v[e] += cr[d] * f[d].v[e] - ci[d] * f[d].v[e + 1];
        y.v[e + 1] += cr[d] * f[d].v[e + 1] + ci[d] * f[d].v[e];
      }
#pragma unroll
    for (int e = 0; e < M; e++) y.v[e] *= (real)arg.scale;
    reinterpret_cast<V *>(arg.y[seg])[j] = y;
    if (RES) {
      const real ar = (real)arg.ar, ai = (real)arg.ai;
#pragma unroll
      for (int e = 0; e < M; e += 2) {
        r.v[e] -= ar * y.v[e] - ai * y.v[e + 1];
        r.v[e + 1] -= ar * y.v[e + 1] + ai * y.v[e];
        red[0] += (double)r.v[e] * (double)r.v[e] + (double)r.v[e + 1] * (double)r.v[e + 1];
        red[1] += (double)y.v[e] * (double)y.v[e] + (double)y.v[e + 1] * (double)y.v[e + 1];
      }
      reinterpret_cast<V *>(arg.r[seg])[j] = r;
    }
  }
  if (RES) finish_reduction<2>(red, arg.c);
}

static Seg segOf(const ColorSpinorField &f) {
  Seg s;
  ColorSpinorField &g = const_cast<ColorSpinorField &>(f);
  if (f.SiteSubset() == QUDA_FULL_SITE_SUBSET) {
    s.v[0] = g.Even().V(); s.v[1] = g.Odd().V();
    s.norm[0] = (float *)g.Even().Norm(); s.norm[1] = (float *)g.Odd().Norm();
  } else {
    s.v[0] = g.V(); s.v[1] = nullptr; s.norm[0] = (float *)g.Norm(); s.norm[1] = nullptr;
  }
  return s;
}

static void checkSame(const ColorSpinorField &a, const ColorSpinorField &b) {
  if (a.Location() != QUDA_CUDA_FIELD_LOCATION || b.Location() != QUDA_CUDA_FIELD_LOCATION) errorQuda("blas needs device fields");
  if (a.Precision() != b.Precision()) errorQuda("blas precision mismatch %d vs %d (copy to a common precision first)", a.Precision(), b.Precision());
  if (a.VolumeCB() != b.VolumeCB() || a.SiteSubset() != b.SiteSubset() || a.Nspin() != b.Nspin() || a.Ncolor() != b.Ncolor() || a.Stride() != b.Stride())
    errorQuda("blas geometry mismatch");
}

template <typename F>
static void launch(const F &f, const ColorSpinorField &x, const ColorSpinorField *y, const ColorSpinorField *z, const ColorSpinorField *w, double *out) {
  if (!d_red) init();
  if (y) checkSame(x, *y);
  if (z) checkSame(x, *z);
  if (w) checkSame(x, *w);
  BlasArg<F> arg;
  arg.x = segOf(x);
  arg.y = y ? segOf(*y) : arg.x;
  arg.z = z ? segOf(*z) : arg.x;
  arg.w = w ? segOf(*w) : arg.x;
  arg.nseg = x.SiteSubset() == QUDA_FULL_SITE_SUBSET ? 2 : 1;
  arg.stride = x.Stride();
  arg.f = f;
  arg.red = d_red;
  arg.part = d_part;
  arg.count = d_count;
  bool allreduce = g_global_reduction && commReductionsNeeded(), peer = false;
  arg.hred = allreduce ? nullptr : h_red_dev;
  arg.peerSlots = nullptr; arg.peerCount = nullptr; arg.nranks = 1; arg.rank = 0; arg.buf = 0; arg.expect = 0; arg.waitTicks = 0;
  if (allreduce && F::nred > 0 && F::nred <= kSlotDoubles && reduceWindowActive()) {
    // the last block of the kernel does the all-reduce itself through the peer windows and writes the global sums to the host
    const CommGrid &cg = commGrid();
    arg.peerSlots = g_rw.d_slots; arg.peerCount = g_rw.d_counts;
    arg.nranks = cg.size; arg.rank = cg.rank;
    arg.buf = (int)(++g_rw.seq & 1);
    arg.expect = (g_rw.uses[arg.buf] += (unsigned)cg.size);
    arg.waitTicks = 2000000ull;   // 20 ms of in-kernel waiting (ranks in step arrive within microseconds), then the host takes over
    arg.hred = h_red_dev;
    allreduce = false;
    peer = true;
    p2pStats()[4]++;
  } else if (allreduce) p2pStats()[5]++;
  hipStream_t s = computeStream();
  const long nreal = (long)x.Stride() * x.Nspin() * x.Ncolor() * 2;
  const int bs = 256;
  // grid cap: 512 blocks (2 per CU) measured best on MI355X — 32x16x16x16 fp64 norm2 65 us at 4096 blocks (the completion
  // counter / fp64 atomics of 3072 blocks serialise on one address) -> 34 us at 512; streaming axpy 5.5 -> 6.0 TB/s
  static int cap = 0;
  if (!cap) { const char *e = getenv("QUDA_AMD_BLAS_BLOCKS"); cap = e ? atoi(e) : 512; if (cap < 1 || cap > kMaxBlocks) cap = 512; }
  auto grid = [&](long n) { long b = (n * arg.nseg + bs - 1) / bs; return (int)(b > cap ? cap : (b < 1 ? 1 : b)); };
  switch (x.Precision()) {
    case QUDA_DOUBLE_PRECISION:
      arg.n = nreal / 2;
      hipLaunchKernelGGL((blas_kernel<double, 2, false, F>), dim3(grid(arg.n)), dim3(bs), 0, s, arg);
      break;
    case QUDA_SINGLE_PRECISION:
      arg.n = nreal / 4;
      if (nreal % 4) errorQuda("field length %ld not a multiple of 4", nreal);
      hipLaunchKernelGGL((blas_kernel<float, 4, false, F>), dim3(grid(arg.n)), dim3(bs), 0, s, arg);
      break;
    case QUDA_HALF_PRECISION:
      if (x.Nspin() != 4 || x.Ncolor() != 3) errorQuda("16-bit blas only for fine-grid spinors");
      arg.n = x.VolumeCB();
      hipLaunchKernelGGL((blas_kernel<short, 24, true, F>), dim3(grid(arg.n)), dim3(bs), 0, s, arg);
      break;
    default: errorQuda("bad precision %d", x.Precision());
  }
  HIP_CHECK(hipGetLastError());
  if (F::nred > 0 && !out) {   // deferred: the sums stay in d_red for the next kernel of the stream (DevAlpha); rank-local sums only
    if (allreduce || peer) errorQuda("a deferred reduction cannot be a global one");
  } else if (F::nred > 0) {
    if (allreduce) {
      commAllreduceDevice(d_red, F::nred, s);  // RCCL all-reduce of the rank sums, in stream order
      HIP_CHECK(hipMemcpyAsync(h_red, d_red, F::nred * sizeof(double), hipMemcpyDeviceToHost, s));
    }
    HIP_CHECK(hipStreamSynchronize(s));
    if (peer && h_red[kMaxRed - 1] == 0.0) finishAllreduceOnHost(arg.buf, arg.expect, F::nred, h_red);
    for (int k = 0; k < F::nred; k++) out[k] = h_red[k];
  }
  const int nrd = F::rx + F::ry + F::rz + F::rw, nwr = F::wx + F::wy + F::wz + F::ww;
  if (g_acctOn) { char tag[48]; snprintf(tag, sizeof(tag), "%s prec %d", x.Nspin() == 4 ? "level 0" : "coarse", (int)x.Precision()); acct("blas_kernel", (double)(nrd + nwr) * x.RealLength() * x.Precision(), tag); }
  bytes += (unsigned long long)(nrd + nwr) * x.RealLength() * x.Precision();
  flops += (unsigned long long)2 * x.RealLength() * (nrd + nwr);
}

// ---- host side of the multi-field kernels ----
bool multiSupported(const ColorSpinorField &x, int k) {
  static int off = -1;
  if (off < 0) { const char *e = getenv("QUDA_AMD_GCR_BLOCK_ORTHO"); off = (e && !atoi(e)) ? 1 : 0; }
  return !off && k >= 0 && k <= kMaxDirs && (x.Precision() == QUDA_DOUBLE_PRECISION || x.Precision() == QUDA_SINGLE_PRECISION) && x.Location() == QUDA_CUDA_FIELD_LOCATION;
}
struct RedPlan { bool allreduce, peer; };
static RedPlan planReduction(RedCtl &c, int nred) {
  if (!d_red) init();
  RedPlan p;
  p.allreduce = g_global_reduction && commReductionsNeeded(); p.peer = false;
  c.red = d_red; c.part = d_part; c.count = d_count;
  c.hred = p.allreduce ? nullptr : h_red_dev;
  c.peerSlots = nullptr; c.peerCount = nullptr; c.nranks = 1; c.rank = 0; c.buf = 0; c.expect = 0; c.waitTicks = 0;
  if (p.allreduce && nred <= kSlotDoubles && reduceWindowActive()) {
    const CommGrid &cg = commGrid();
    c.peerSlots = g_rw.d_slots; c.peerCount = g_rw.d_counts;
    c.nranks = cg.size; c.rank = cg.rank;
    c.buf = (int)(++g_rw.seq & 1);
    c.expect = (g_rw.uses[c.buf] += (unsigned)cg.size);
    c.waitTicks = 2000000ull;
    c.hred = h_red_dev;
    p.allreduce = false; p.peer = true;
    p2pStats()[4]++;
  } else if (p.allreduce) p2pStats()[5]++;
  return p;
}
static void finishPlan(const RedPlan &p, const RedCtl &c, int nred, double *out, hipStream_t s) {
  if (p.allreduce) {
    commAllreduceDevice(d_red, nred, s);
    HIP_CHECK(hipMemcpyAsync(h_red, d_red, nred * sizeof(double), hipMemcpyDeviceToHost, s));
  }
  HIP_CHECK(hipStreamSynchronize(s));
  if (p.peer && h_red[kMaxRed - 1] == 0.0) finishAllreduceOnHost(c.buf, c.expect, nred, h_red);
  for (int k = 0; k < nred; k++) out[k] = h_red[k];
}
// work-groups of the multi-field kernels.  Measured at 48^3 x 96 (fp32, k = 1..9 fields, tools/profile_mg_solve.sh with QUDA_AMD_BLAS_BLOCKS):
// the k-field update runs at 0.43 / 0.61 / 0.69 of the HBM roofline with 512 / 1024 / 2048 work-groups (one chunk of every field per
// thread and trip: more resident waves, not deeper loops, keep the k + 2 streams busy), the k-field dots at 0.56 / 0.59 / 0.54 (their
// 43-value block reduction grows with the grid); the one- to four-field kernels above prefer 512 (0.72 against 0.67).
static int multiGrid(long chunks, int dflt) {
  static int cap = -1;
  if (cap < 0) { const char *e = getenv("QUDA_AMD_BLAS_BLOCKS"); cap = e ? atoi(e) : 0; if (cap < 1 || cap > kMaxBlocks) cap = 0; }
  const int c = cap ? cap : dflt;
  const long b = (chunks + 255) / 256;
  return (int)(b > c ? c : (b < 1 ? 1 : b));
}
static void fillMulti(MultiArg &a, const std::vector<ColorSpinorField *> &f, int k, const ColorSpinorField &y, const ColorSpinorField *r) {
  if (!multiSupported(y, k)) errorQuda("multi-field blas: %d fields of precision %d not supported", k, y.Precision());
  if ((int)f.size() < k) errorQuda("multi-field blas: %zu fields given, %d wanted", f.size(), k);
  const Seg sy = segOf(y);
  a.y[0] = sy.v[0]; a.y[1] = sy.v[1];
  a.r[0] = a.r[1] = nullptr;
  if (r) { checkSame(y, *r); const Seg sr = segOf(*r); a.r[0] = sr.v[0]; a.r[1] = sr.v[1]; }
  for (int i = 0; i < kMaxDirs; i++) { a.f[i][0] = a.f[i][1] = nullptr; a.cr[i] = a.ci[i] = 0.0; }
  for (int i = 0; i < k; i++) { checkSame(y, *f[i]); const Seg sf = segOf(*f[i]); a.f[i][0] = sf.v[0]; a.f[i][1] = sf.v[1]; }
  // surplus slots of the kernel's bucket (multiple of 4 >= k): a valid field again — y itself where there is none — with a zero
  // coefficient / an ignored sum
  for (int i = k; i < kMaxDirs; i++) { a.f[i][0] = k ? a.f[0][0] : a.y[0]; a.f[i][1] = k ? a.f[0][1] : a.y[1]; }
  a.k = k; a.nseg = y.SiteSubset() == QUDA_FULL_SITE_SUBSET ? 2 : 1;
  const long nreal = (long)y.Stride() * y.Nspin() * y.Ncolor() * 2;
  a.n = y.Precision() == QUDA_DOUBLE_PRECISION ? nreal / 2 : nreal / 4;
  if (y.Precision() == QUDA_SINGLE_PRECISION && nreal % 4) errorQuda("field length %ld not a multiple of 4", nreal);
  a.scale = 1.0; a.ar = a.ai = 0.0;
}
void multiDot(Complex *beta, Complex &yr, double &ynorm, const std::vector<ColorSpinorField *> &f, int k, const ColorSpinorField &y, const ColorSpinorField &r) {
  constexpr int NRED = 2 * kMaxDirs + 3;
  MultiArg a;
  fillMulti(a, f, k, y, &r);
  const int KBsel = k <= 4 ? 4 : (k <= 8 ? 8 : (k <= 12 ? 12 : (k <= 16 ? 16 : 20))), nred = 2 * KBsel + 3;   // sums of the bucket: [2 d], [2 d + 1] for its KBsel fields, then (y, r) and |y|^2
  const RedPlan p = planReduction(a.c, nred);
  hipStream_t s = computeStream();
#define QA_MD(KB) { if (y.Precision() == QUDA_DOUBLE_PRECISION) hipLaunchKernelGGL((multi_dot_kernel<double, 2, KB>), dim3(multiGrid(a.n * a.nseg, 1024)), dim3(256), 0, s, a); \
                   else hipLaunchKernelGGL((multi_dot_kernel<float, 4, KB>), dim3(multiGrid(a.n * a.nseg, 1024)), dim3(256), 0, s, a); }
  if (k <= 4) QA_MD(4) else if (k <= 8) QA_MD(8) else if (k <= 12) QA_MD(12) else if (k <= 16) QA_MD(16) else QA_MD(20)
#undef QA_MD
  HIP_CHECK(hipGetLastError());
  double out[NRED];
  finishPlan(p, a.c, nred, out, s);
  for (int i = 0; i < k; i++) beta[i] = Complex(out[2 * i], out[2 * i + 1]);
  yr = Complex(out[2 * KBsel], out[2 * KBsel + 1]);
  ynorm = out[2 * KBsel + 2];
  acct("multi_dot_kernel", (double)(k + 2) * y.RealLength() * y.Precision(), y.Nspin() == 4 ? "level 0" : "coarse");
  bytes += (unsigned long long)(k + 2) * y.RealLength() * y.Precision();
  flops += (unsigned long long)(8 * k + 12) * (y.RealLength() / 2);
}
void multiCaxpyResidual(double &r2, double &y2, const Complex *c, const std::vector<ColorSpinorField *> &f, int k, double scale, ColorSpinorField &y, const Complex &a_, ColorSpinorField &r) {
  MultiArg a;
  fillMulti(a, f, k, y, &r);
  for (int i = 0; i < k; i++) { a.cr[i] = c[i].real(); a.ci[i] = c[i].imag(); }
  a.scale = scale; a.ar = a_.real(); a.ai = a_.imag();
  const RedPlan p = planReduction(a.c, 2);
  hipStream_t s = computeStream();
#define QA_MC(KB) { if (y.Precision() == QUDA_DOUBLE_PRECISION) hipLaunchKernelGGL((multi_caxpy_kernel<double, 2, true, KB>), dim3(multiGrid(a.n * a.nseg, 2048)), dim3(256), 0, s, a); \
                   else hipLaunchKernelGGL((multi_caxpy_kernel<float, 4, true, KB>), dim3(multiGrid(a.n * a.nseg, 2048)), dim3(256), 0, s, a); }
  if (k <= 4) QA_MC(4) else if (k <= 8) QA_MC(8) else if (k <= 12) QA_MC(12) else if (k <= 16) QA_MC(16) else QA_MC(20)
#undef QA_MC
  HIP_CHECK(hipGetLastError());
  double out[2];
  finishPlan(p, a.c, 2, out, s);
  r2 = out[0]; y2 = out[1];
  acct("multi_caxpy_kernel", (double)(k + 4) * y.RealLength() * y.Precision(), y.Nspin() == 4 ? "level 0" : "coarse");
  bytes += (unsigned long long)(k + 4) * y.RealLength() * y.Precision();
  flops += (unsigned long long)(8 * k + 16) * (y.RealLength() / 2);
}
void multiCaxpy(const Complex *c, const std::vector<ColorSpinorField *> &f, int k, ColorSpinorField &y) {
  MultiArg a;
  fillMulti(a, f, k, y, nullptr);
  for (int i = 0; i < k; i++) { a.cr[i] = c[i].real(); a.ci[i] = c[i].imag(); }
  memset(&a.c, 0, sizeof(a.c));
  hipStream_t s = computeStream();
#define QA_MC(KB) { if (y.Precision() == QUDA_DOUBLE_PRECISION) hipLaunchKernelGGL((multi_caxpy_kernel<double, 2, false, KB>), dim3(multiGrid(a.n * a.nseg, 2048)), dim3(256), 0, s, a); \
                   else hipLaunchKernelGGL((multi_caxpy_kernel<float, 4, false, KB>), dim3(multiGrid(a.n * a.nseg, 2048)), dim3(256), 0, s, a); }
  if (k <= 4) QA_MC(4) else if (k <= 8) QA_MC(8) else if (k <= 12) QA_MC(12) else if (k <= 16) QA_MC(16) else QA_MC(20)
#undef QA_MC
  HIP_CHECK(hipGetLastError());
  acct("multi_caxpy_kernel", (double)(k + 2) * y.RealLength() * y.Precision(), y.Nspin() == 4 ? "level 0" : "coarse");
  bytes += (unsigned long long)(k + 2) * y.RealLength() * y.Precision();
  flops += (unsigned long long)(8 * k) * (y.RealLength() / 2);
}

void zero(ColorSpinorField &a) { a.zero(); }
void copy(ColorSpinorField &dst, const ColorSpinorField &src) { copyColorSpinor(dst, src); }

double norm2(const ColorSpinorField &a) { double r[1]; launch(Norm2F(), a, nullptr, nullptr, nullptr, r); return r[0]; }
double reDotProduct(const ColorSpinorField &x, const ColorSpinorField &y) { double r[1]; launch(ReDotF(), x, &y, nullptr, nullptr, r); return r[0]; }
Complex cDotProduct(const ColorSpinorField &x, const ColorSpinorField &y) { double r[2]; launch(CDotF<0>(), x, &y, nullptr, nullptr, r); return Complex(r[0], r[1]); }
double3_t cDotProductNormA(const ColorSpinorField &x, const ColorSpinorField &y) { double r[3]; launch(CDotF<1>(), x, &y, nullptr, nullptr, r); return {r[0], r[1], r[2]}; }
double3_t cDotProductNormB(const ColorSpinorField &x, const ColorSpinorField &y) { double r[3]; launch(CDotF<2>(), x, &y, nullptr, nullptr, r); return {r[0], r[1], r[2]}; }

void ax(const double &a, ColorSpinorField &x) { AxF f; f.a = a; launch(f, x, nullptr, nullptr, nullptr, nullptr); }
void axpby(const double &a, const ColorSpinorField &x, const double &b, ColorSpinorField &y) { AxpbyF<0> f; f.a = a; f.b = b; launch(f, x, &y, nullptr, nullptr, nullptr); }
void axpy(const double &a, const ColorSpinorField &x, ColorSpinorField &y) { axpby(a, x, 1.0, y); }
void xpy(const ColorSpinorField &x, ColorSpinorField &y) { axpby(1.0, x, 1.0, y); }
void xpay(const ColorSpinorField &x, const double &a, ColorSpinorField &y) { axpby(1.0, x, a, y); }
void mxpy(const ColorSpinorField &x, ColorSpinorField &y) { axpby(-1.0, x, 1.0, y); }
double xmyNorm(const ColorSpinorField &x, ColorSpinorField &y) { AxpbyF<1> f; f.a = 1.0; f.b = -1.0; double r[1]; launch(f, x, &y, nullptr, nullptr, r); return r[0]; }
double axpyNorm(const double &a, const ColorSpinorField &x, ColorSpinorField &y) { AxpbyF<1> f; f.a = a; f.b = 1.0; double r[1]; launch(f, x, &y, nullptr, nullptr, r); return r[0]; }

void caxpby(const Complex &a, const ColorSpinorField &x, const Complex &b, ColorSpinorField &y) {
  CaxpbyF<0> f; f.ar = a.real(); f.ai = a.imag(); f.br = b.real(); f.bi = b.imag();
  launch(f, x, &y, nullptr, nullptr, nullptr);
}
void caxpy(const Complex &a, const ColorSpinorField &x, ColorSpinorField &y) { caxpby(a, x, Complex(1.0, 0.0), y); }
double caxpyNorm(const Complex &a, const ColorSpinorField &x, ColorSpinorField &y) {
  CaxpbyF<1> f; f.ar = a.real(); f.ai = a.imag(); f.br = 1.0; f.bi = 0.0; double r[1];
  launch(f, x, &y, nullptr, nullptr, r); return r[0];
}
void xmyz(const ColorSpinorField &x, const ColorSpinorField &y, ColorSpinorField &z) { launch(XmyzF(), x, &y, &z, nullptr, nullptr); }
void cxpaypbz(const ColorSpinorField &x, const Complex &a, const ColorSpinorField &y, const Complex &b, ColorSpinorField &z) {
  CxpaypbzF f; f.ar = a.real(); f.ai = a.imag(); f.br = b.real(); f.bi = b.imag();
  launch(f, x, &y, &z, nullptr, nullptr);
}
void caxpyXmaz(const Complex &a, ColorSpinorField &x, ColorSpinorField &y, const ColorSpinorField &z) {
  CaxpyXmazF<0> f; f.ar = a.real(); f.ai = a.imag(); launch(f, x, &y, &z, nullptr, nullptr);
}
void caxpyXmazMR(const Complex &a, ColorSpinorField &x, ColorSpinorField &y, const ColorSpinorField &z) { caxpyXmaz(a, x, y, z); }
bool deviceScalars() {
  static int on = -1;
  if (on < 0) { const char *e = getenv("QUDA_AMD_DEVICE_SCALARS"); on = e ? atoi(e) : 1; }
  return on && !(g_global_reduction && commReductionsNeeded());
}
void cDotProductNormADev(const ColorSpinorField &x, const ColorSpinorField &y) {
  if (!deviceScalars()) errorQuda("device-side scalars need rank-local reductions");
  launch(CDotF<1>(), x, &y, nullptr, nullptr, nullptr);
}
void caxpyXmazDev(double omega, ColorSpinorField &x, ColorSpinorField &y, const ColorSpinorField &z) {
  DevAlpha<CaxpyXmazF<0>> f; f.ar = f.ai = 0; f.dres = d_red; f.omega = omega; launch(f, x, &y, &z, nullptr, nullptr);
}
void caxXmazDev(double omega, ColorSpinorField &x, ColorSpinorField &y, const ColorSpinorField &z) {
  DevAlpha<CaxXmazF> f; f.ar = f.ai = 0; f.dres = d_red; f.omega = omega; launch(f, x, &y, &z, nullptr, nullptr);
}
void caxInitDev(double omega, const ColorSpinorField &x, ColorSpinorField &y, const ColorSpinorField &z, ColorSpinorField &w) {
  DevAlpha<CaxInitF> f; f.ar = f.ai = 0; f.dres = d_red; f.omega = omega; launch(f, x, &y, &z, &w, nullptr);
}
void caxXmaz(const Complex &a, ColorSpinorField &x, ColorSpinorField &y, const ColorSpinorField &z) {
  CaxXmazF f; f.ar = a.real(); f.ai = a.imag(); launch(f, x, &y, &z, nullptr, nullptr);
}
void caxInit(const Complex &a, const ColorSpinorField &x, ColorSpinorField &y, const ColorSpinorField &z, ColorSpinorField &w) {
  CaxInitF f; f.ar = a.real(); f.ai = a.imag(); launch(f, x, &y, &z, &w, nullptr);
}
double caxpyXmazNormX(const Complex &a, ColorSpinorField &x, ColorSpinorField &y, const ColorSpinorField &z) {
  CaxpyXmazF<1> f; f.ar = a.real(); f.ai = a.imag(); double r[1]; launch(f, x, &y, &z, nullptr, r); return r[0];
}
void cabxpyAx(const double &a, const Complex &b, ColorSpinorField &x, ColorSpinorField &y) {
  CabxpyAxF<0> f; f.a = a; f.br = b.real(); f.bi = b.imag(); launch(f, x, &y, nullptr, nullptr, nullptr);
}
double cabxpyAxNorm(const double &a, const Complex &b, ColorSpinorField &x, ColorSpinorField &y) {
  CabxpyAxF<1> f; f.a = a; f.br = b.real(); f.bi = b.imag(); double r[1]; launch(f, x, &y, nullptr, nullptr, r); return r[0];
}
Complex caxpyDotzy(const Complex &a, const ColorSpinorField &x, ColorSpinorField &y, const ColorSpinorField &z) {
  CaxpyDotzyF f; f.ar = a.real(); f.ai = a.imag(); double r[2]; launch(f, x, &y, &z, nullptr, r); return Complex(r[0], r[1]);
}
void caxpbypzYmbw(const Complex &a, const ColorSpinorField &x, const Complex &b, ColorSpinorField &y, ColorSpinorField &z, const ColorSpinorField &w) {
  CaxpbypzYmbwF f; f.ar = a.real(); f.ai = a.imag(); f.br = b.real(); f.bi = b.imag();
  launch(f, x, &y, &z, &w, nullptr);
}

// heavy-quark residual: sum_sites |r(x)|^2 / |x(x)|^2 — needs the site structure (reference lib/blas_cpu.cpp:311-352)
template <typename T> __global__ void hq_kernel(const void *x, const float *xn, const void *r, const float *rn, int stride, int Vh, double *red) {
  using real = typename Store<T>::real;
  double acc[3] = {0, 0, 0};
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < Vh; i += gridDim.x * blockDim.x) {
    real a[24], b[24];
    Planar<T, 24>::load(a, x, stride, i, xn, i);
    Planar<T, 24>::load(b, r, stride, i, rn, i);
    double x2 = 0, r2 = 0;
#pragma unroll
    for (int k = 0; k < 24; k++) { x2 += (double)a[k] * a[k]; r2 += (double)b[k] * b[k]; }
    acc[0] += x2; acc[1] += r2; acc[2] += x2 > 0.0 ? r2 / x2 : 1.0;
  }
  for (int k = 0; k < 3; k++) {
    double v = acc[k];
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(&red[k], v);
  }
}

double3_t HeavyQuarkResidualNorm(const ColorSpinorField &x, const ColorSpinorField &r) {
  if (!d_red) init();
  checkSame(x, r);
  if (x.Nspin() != 4 || x.Ncolor() != 3) errorQuda("heavy-quark residual only for fine-grid spinors");
  hipStream_t s = computeStream();
  HIP_CHECK(hipMemsetAsync(d_red, 0, 3 * sizeof(double), s));
  const Seg sx = segOf(x), sr = segOf(r);
  const int nseg = x.SiteSubset() == QUDA_FULL_SITE_SUBSET ? 2 : 1, bs = 256;
  for (int sg = 0; sg < nseg; sg++) {
    int nb = (x.VolumeCB() + bs - 1) / bs; if (nb > 2048) nb = 2048;
    switch (x.Precision()) {
      case QUDA_DOUBLE_PRECISION: hipLaunchKernelGGL((hq_kernel<double>), dim3(nb), dim3(bs), 0, s, sx.v[sg], sx.norm[sg], sr.v[sg], sr.norm[sg], x.Stride(), x.VolumeCB(), d_red); break;
      case QUDA_SINGLE_PRECISION: hipLaunchKernelGGL((hq_kernel<float>), dim3(nb), dim3(bs), 0, s, sx.v[sg], sx.norm[sg], sr.v[sg], sr.norm[sg], x.Stride(), x.VolumeCB(), d_red); break;
      default: hipLaunchKernelGGL((hq_kernel<short>), dim3(nb), dim3(bs), 0, s, sx.v[sg], sx.norm[sg], sr.v[sg], sr.norm[sg], x.Stride(), x.VolumeCB(), d_red); break;
    }
  }
  HIP_CHECK(hipMemcpyAsync(h_red, d_red, 3 * sizeof(double), hipMemcpyDeviceToHost, s));
  HIP_CHECK(hipStreamSynchronize(s));
  double out[3] = {h_red[0], h_red[1], h_red[2]};
  double vol = (double)x.Volume();
  if (g_global_reduction) { comm_allreduce(out, 3); double v[1] = {vol}; comm_allreduce(v, 1); vol = v[0]; }
  return {out[0], out[1], out[2] / vol};
}

// vectorised forms: y_j += sum_i a[i*ny + j] x_i ; result[i*nb + j] = (a_i, b_j)
void caxpy(const Complex *a, std::vector<ColorSpinorField *> &x, std::vector<ColorSpinorField *> &y) {
  for (size_t j = 0; j < y.size(); j++)
    for (size_t i = 0; i < x.size(); i++) caxpy(a[i * y.size() + j], *x[i], *y[j]);
}
void cDotProduct(Complex *result, std::vector<ColorSpinorField *> &a, std::vector<ColorSpinorField *> &b) {
  for (size_t i = 0; i < a.size(); i++)
    for (size_t j = 0; j < b.size(); j++) result[i * b.size() + j] = cDotProduct(*a[i], *b[j]);
}

}  // namespace blas
}  // namespace quda
