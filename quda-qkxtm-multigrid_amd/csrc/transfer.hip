// transfer.hip — restrictor / prolongator / block Gram-Schmidt kernels and the Transfer host class (see transfer.h).
#include "transfer.h"
#include "device_io.h"   // pk:: packed fp32 complex arithmetic

#include <cstring>

namespace quda {

constexpr int kMaxVec = 32;

// fine-vector element access: complex number sc of site x in one parity block of a planar field with NV reals per vector
template <int NV> __device__ __forceinline__ size_t fidx(int stride, int x, int sc) {
  const int r = 2 * sc;
  return ((size_t)(r / NV) * stride + x) * NV + (r % NV);
}

// the K complex components of fine site x into r[]: for the float4-plane order (NV = 4: components 2m, 2m + 1 share a plane entry)
// with one 16-byte load per pair — as 4-byte loads the gather of the four fine vectors of restrict4_kernel was 96 instructions per
// thread, each touching 32 cache lines per wave
template <int NV, int K> __device__ __forceinline__ void load_fine_site(float2 *r, const float *base, int stride, int x) {
  if (NV == 4 && K % 2 == 0) {
#pragma unroll
    for (int m = 0; m < K / 2; m++) {
      const float4 v = reinterpret_cast<const float4 *>(base)[(size_t)m * stride + x];
      r[2 * m] = make_float2(v.x, v.y); r[2 * m + 1] = make_float2(v.z, v.w);
    }
  } else {
#pragma unroll
    for (int k = 0; k < K; k++) { const size_t i = fidx<NV>(stride, x, k); r[k] = make_float2(base[i], base[i + 1]); }
  }
}

struct FineVec {
  float *v[2];  // even / odd parity block
  int stride, Vh;
};
struct CoarseVec {
  float *v[2];
  int stride, Vh;
};

// parity >= 0: `f` is a single-parity field holding that parity; the other parity is absent (nullptr): it restricts as zero
// and is not written by the prolongator (reference Transfer::setSiteSubset, lib/transfer.cpp:276-290)
static FineVec fineVec(const ColorSpinorField &f, int parity = -1) {
  if (f.Location() != QUDA_CUDA_FIELD_LOCATION || f.Precision() != QUDA_SINGLE_PRECISION)
    errorQuda("transfer operators work on fp32 device fields (precision %d)", f.Precision());
  ColorSpinorField &g = const_cast<ColorSpinorField &>(f);
  FineVec r;
  if (f.SiteSubset() == QUDA_FULL_SITE_SUBSET) {
    r.v[0] = (float *)g.Even().V(); r.v[1] = (float *)g.Odd().V();
  } else {
    if (parity != 0 && parity != 1) errorQuda("single-parity field without Transfer::setSiteSubset(QUDA_PARITY_SITE_SUBSET, parity)");
    r.v[parity] = (float *)g.V(); r.v[1 - parity] = nullptr;
  }
  r.stride = f.Stride(); r.Vh = f.VolumeCB();
  return r;
}

struct MaskArg {
  int dir;       // -1: no mask
  int boundary;  // 1: keep sites whose dir-neighbour is outside the aggregate, 0: inside
  int bs[4];
  int single[4]; // coarse extent 1 in that dimension: the neighbour always wraps into the same aggregate
  int pm;        // parity-major order of the sites inside an aggregate (Transfer::parityMajor), else lexicographic
};
// coordinates of site b of an aggregate.  Lexicographic (x fastest), or — parity-major, all block extents even — the even sites of the
// aggregate first, then the odd ones, each half in lexicographic order halved (the checkerboard numbering of the lattice applied to
// the block): a single-parity fine field (Transfer::setSiteSubset) then meets ONE contiguous half of every (component, vector pair)
// row of V, and the waves that own the absent parity issue no loads at all — half the V bytes for R and P of an even-odd cycle.
__device__ __forceinline__ void block_coords(int *y, const MaskArg &m, int b) {
  int l = b, p = 0;
  if (m.pm) {
    const int half = (m.bs[0] * m.bs[1] * m.bs[2] * m.bs[3]) >> 1;
    p = b >= half;
    l = 2 * (b - p * half);
  }
  y[0] = l % m.bs[0]; l /= m.bs[0];
  y[1] = l % m.bs[1]; l /= m.bs[1];
  y[2] = l % m.bs[2]; y[3] = l / m.bs[2];
  if (m.pm) y[0] += (p + y[1] + y[2] + y[3]) & 1;   // y[0] is even here: the site of the pair that has parity p
}

__device__ __forceinline__ bool mask_keep(const MaskArg &m, int b) {
  if (m.dir < 0) return true;
  const int mu = m.dir >> 1, fwd = !(m.dir & 1);
  int y[4];
  block_coords(y, m, b);
  const bool out = !m.single[mu] && (fwd ? y[mu] == m.bs[mu] - 1 : y[mu] == 0);
  return out == (m.boundary != 0);
}

// ---- block reduction of one float4 across the work-group (wave64 shuffles, then LDS across waves) ----
__device__ __forceinline__ float4 block_sum4(float4 v, float4 *lds) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    v.x += __shfl_down(v.x, off, 64); v.y += __shfl_down(v.y, off, 64);
    v.z += __shfl_down(v.z, off, 64); v.w += __shfl_down(v.w, off, 64);
  }
  const int wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) lds[wave] = v;
  __syncthreads();
  float4 r = lds[0];
  for (int w = 1; w < nw; w++) { r.x += lds[w].x; r.y += lds[w].y; r.z += lds[w].z; r.w += lds[w].w; }
  return r;
}

// ---- restrictor: one work-group per aggregate, one thread per fine site of it.  DUAL (Galerkin construction): the sites whose
// mask.dir neighbour leaves the aggregate go into `out`, the others into `out2`, in ONE pass over V (V is the whole cost) ----
__device__ __forceinline__ bool mask_outside(const MaskArg &m, int b) {
  const int mu = m.dir >> 1, fwd = !(m.dir & 1);
  int y[4];
  block_coords(y, m, b);
  return !m.single[mu] && (fwd ? y[mu] == m.bs[mu] - 1 : y[mu] == 0);
}

typedef _Float16 vhalf4_t __attribute__((ext_vector_type(4)));
template <bool HALF> __device__ __forceinline__ float4 load_v(const void *V, size_t i) {
  if (HALF) {
    const vhalf4_t h = reinterpret_cast<const vhalf4_t *>(V)[i];
    return make_float4((float)h.x, (float)h.y, (float)h.z, (float)h.w);
  }
  // V is streamed once per kernel (24.5 GB at 48^3 x 96): non-temporal loads keep it from displacing the fine and coarse vectors the
  // same kernel reads and writes (prolongator 0.58 -> 0.76 of the HBM roofline at 32^4)
  typedef float f32x4_nt __attribute__((ext_vector_type(4)));
  const f32x4_nt t = __builtin_nontemporal_load(reinterpret_cast<const f32x4_nt *>(V) + i);
  return make_float4(t.x, t.y, t.z, t.w);
}

// work-group -> aggregate.  xgroup > 0: the four aggregates that are neighbours along x (xgroup = Xc[0] / 4 such groups per row) take
// four consecutive slots of ONE XCD (work-groups b, b + 8, b + 16, b + 24).  An aggregate 4 sites wide covers 2 consecutive
// checkerboard sites per row and parity, i.e. a 32-byte piece of every 128-byte line of the fine field; written a quarter at a
// time by work-groups far apart in the grid (neighbours along x differ in coarse parity: half a grid apart) every line went to
// memory in pieces once the field no longer fits the Infinity Cache — prolongator at 48^3 x 96: 0.56 of the HBM roofline against
// 0.76 at 32^4.  With the four quarters written from one L2 within microseconds the line leaves it whole.  The restrictors READ the
// fine vectors in the same 32-byte pieces (restrict4_kernel four of them at once): same order, so that three of the four pieces of a
// line are L2 hits.
struct AggMap { int xgroup, X0, X1, X2, Vh; };
__device__ __forceinline__ int aggregate_of_block(const AggMap &m) {
  const int b = blockIdx.x;
  if (!m.xgroup) return b;
  const int xcd = b & 7, within = b >> 3;
  int g = xcd + 8 * (within >> 2);
  const int r = within & 3;
  const int xg = g % m.xgroup; g /= m.xgroup;
  const int y = g % m.X1; g /= m.X1;
  const int z = g % m.X2; const int t = g / m.X2;
  const int x = 4 * xg + r;
  return ((x + y + z + t) & 1) * m.Vh + ((((t * m.X2 + z) * m.X1 + y) * m.X0 + x) >> 1);
}

template <int NSF, int NCF, int NVEC, int NV, bool DUAL, bool HALF = false>
__global__ void restrict_kernel(CoarseVec out, CoarseVec out2, FineVec in, const void *V, const int *block_to_fine, int blockVol, int spin_bs, MaskArg mask, AggMap amap) {
  constexpr int K = NSF * NCF;
  __shared__ float4 lds[16];
  const int A = aggregate_of_block(amap), b = threadIdx.x;
  const bool active = b < blockVol && (DUAL || mask_keep(mask, b));
  const bool outside = DUAL && b < blockVol && mask_outside(mask, b);
  float2 r[K];
  bool have = false;
  if (active) {
    const int f = block_to_fine[(size_t)A * blockVol + b];
    const int parity = f >= in.Vh, x = f - parity * in.Vh;
    const float *base = in.v[parity];
    if (base) {   // nullptr: this parity is absent from a single-parity field
      have = true;
load_fine_site<NV, K>(r, base, in.stride, x);
    }
  }
  const int cpar = A >= out.Vh, xc = A - cpar * out.Vh;
  float *ob = out.v[cpar], *ob2 = DUAL ? out2.v[cpar] : nullptr;
  for (int chi = 0; chi < 2; chi++) {
    for (int vp = 0; vp < NVEC / 2; vp++) {
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      if (have) {
#pragma unroll
        for (int k = 0; k < K; k++) {
          if ((k / NCF) / spin_bs != chi) continue;
          const float4 w = load_v<HALF>(V, (((size_t)A * K + k) * (NVEC / 2) + vp) * blockVol + b);
          // conj(V) * r
          acc.x += w.x * r[k].x + w.y * r[k].y; acc.y += w.x * r[k].y - w.y * r[k].x;
          acc.z += w.z * r[k].x + w.w * r[k].y; acc.w += w.z * r[k].y - w.w * r[k].x;
        }
      }
      const int c0 = chi * NVEC + 2 * vp;
      if (DUAL) {
        const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 so = block_sum4(outside ? acc : zero4, lds);
        const float4 si = block_sum4(outside ? zero4 : acc, lds);
        if (threadIdx.x == 0) {
          ob[((size_t)c0 * out.stride + xc) * 2] = so.x; ob[((size_t)c0 * out.stride + xc) * 2 + 1] = so.y;
          ob[((size_t)(c0 + 1) * out.stride + xc) * 2] = so.z; ob[((size_t)(c0 + 1) * out.stride + xc) * 2 + 1] = so.w;
          ob2[((size_t)c0 * out2.stride + xc) * 2] = si.x; ob2[((size_t)c0 * out2.stride + xc) * 2 + 1] = si.y;
          ob2[((size_t)(c0 + 1) * out2.stride + xc) * 2] = si.z; ob2[((size_t)(c0 + 1) * out2.stride + xc) * 2 + 1] = si.w;
        }
      } else {
        const float4 s = block_sum4(acc, lds);
        if (threadIdx.x == 0) {
          ob[((size_t)c0 * out.stride + xc) * 2] = s.x; ob[((size_t)c0 * out.stride + xc) * 2 + 1] = s.y;
          ob[((size_t)(c0 + 1) * out.stride + xc) * 2] = s.z; ob[((size_t)(c0 + 1) * out.stride + xc) * 2 + 1] = s.w;
        }
      }
    }
  }
}

// ---- the restrictor of the solve phase (fine level, no split): the V stream without a barrier per step.  restrict_kernel above sums
// every (chirality, vector pair) step over the work-group behind two barriers with only that step's six V entries in flight: bound
// by latency, and with a single-parity source (even-odd cycle, half of the waves idle) it moved the half V in the time the full one
// takes (0.48 of the HBM roofline at 48^3 x 96 against 0.78 for full fields; fp16 V 0.30).  Here a wave keeps three steps of requests
// in flight (raw register images, converted at their use), sums four steps at a time with a reduce-scatter butterfly (17 cross-lane
// moves per four steps instead of 96) into its own LDS row, and the work-group meets at ONE barrier.  A wave without a site of the
// present parity (parity-major V) issues nothing.
typedef float f32x4_raw_t __attribute__((ext_vector_type(4)));
template <bool HALF> struct VRaw { typedef f32x4_raw_t type; };
template <> struct VRaw<true> { typedef vhalf4_t type; };
template <bool HALF> __device__ __forceinline__ typename VRaw<HALF>::type load_v_raw(const void *V, size_t i) {
  if constexpr (HALF) return reinterpret_cast<const vhalf4_t *>(V)[i];
  else return __builtin_nontemporal_load(reinterpret_cast<const f32x4_raw_t *>(V) + i);
}
__device__ __forceinline__ float4 v_unpack(const f32x4_raw_t &t) { return make_float4(t.x, t.y, t.z, t.w); }
__device__ __forceinline__ float4 v_unpack(const vhalf4_t &h) { return make_float4((float)h.x, (float)h.y, (float)h.z, (float)h.w); }

template <int NSF, int NCF, int NVEC, int NV, bool HALF>
__global__ void __launch_bounds__(256) restrict_stream_kernel(CoarseVec out, FineVec in, const void *V, const int *block_to_fine, int blockVol, MaskArg mask, AggMap amap) {
  constexpr int K = NSF * NCF, KH = K / 2, NVP = NVEC / 2, NST = 2 * NVP;
  static_assert(NVP % 4 == 0 && NST <= 64, "steps are summed four at a time");
  typedef typename VRaw<HALF>::type raw_t;
  __shared__ float4 part[4][NST];   // [wave][chirality * NVP + vector pair]
  const int A = aggregate_of_block(amap), b = threadIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  float2 r[K];
#pragma unroll
  for (int k = 0; k < K; k++) r[k] = make_float2(0.f, 0.f);
  bool have = false;
  if (b < blockVol && mask_keep(mask, b)) {
    const int f = block_to_fine[(size_t)A * blockVol + b];
    const int parity = f >= in.Vh, x = f - parity * in.Vh;
    const float *base = in.v[parity];
    if (base) { have = true; load_fine_site<NV, K>(r, base, in.stride, x); }   // nullptr: this parity is absent from a single-parity field
  }
  if (__builtin_amdgcn_ballot_w64(have) == 0) {   // wave-uniform
    if (lane < NST) part[wave][lane] = make_float4(0.f, 0.f, 0.f, 0.f);
  } else {
    // unconditional loads (a lane without a site reads entry 0 against r = 0): a load under a per-lane condition is its own basic
    // block and the wait counts fall back to vmcnt(0)
    const int bl = b < blockVol ? b : 0;
#pragma unroll
    for (int chi = 0; chi < 2; chi++) {
      raw_t w0[KH], w1[KH], w2[KH];
      auto vload = [&](raw_t *dst, int vpl) {
        if (vpl >= NVP) return;
#pragma unroll
        for (int kk = 0; kk < KH; kk++) dst[kk] = load_v_raw<HALF>(V, (((size_t)A * K + chi * KH + kk) * NVP + vpl) * blockVol + bl);
      };
      vload(w0, 0); vload(w1, 1); vload(w2, 2);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int g = 0; g < NVP / 4; g++) {
        float w[16];
#pragma unroll
        for (int s4 = 0; s4 < 4; s4++) {
          const int vp = 4 * g + s4, ph = vp % 3;
          float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
          for (int kk = 0; kk < KH; kk++) {
            const float4 v = v_unpack(ph == 0 ? w0[kk] : (ph == 1 ? w1[kk] : w2[kk]));
            const float2 rr = r[chi * KH + kk];
            acc.x += v.x * rr.x + v.y * rr.y; acc.y += v.x * rr.y - v.y * rr.x;   // conj(V) r
            acc.z += v.z * rr.x + v.w * rr.y; acc.w += v.z * rr.y - v.w * rr.x;
          }
          w[4 * s4] = acc.x; w[4 * s4 + 1] = acc.y; w[4 * s4 + 2] = acc.z; w[4 * s4 + 3] = acc.w;
          __builtin_amdgcn_sched_barrier(0);
          if (ph == 0) vload(w0, vp + 3); else if (ph == 1) vload(w1, vp + 3); else vload(w2, vp + 3);   // the buffer just used: three steps ahead
          __builtin_amdgcn_sched_barrier(0);
        }
#define QA_BFLY(HALFN, M)                                                                  \
        {                                                                                  \
          const bool up = (lane & M) != 0;                                                 \
          _Pragma("unroll") for (int j = 0; j < HALFN; j++) {                              \
            const float keep = up ? w[HALFN + j] : w[j], give = up ? w[j] : w[HALFN + j]; \
            w[j] = keep + __shfl_xor(give, M, 64);                                         \
          }                                                                                \
        }
        QA_BFLY(8, 1) QA_BFLY(4, 2) QA_BFLY(2, 4) QA_BFLY(1, 8)
#undef QA_BFLY
        w[0] += __shfl_xor(w[0], 16, 64);
        w[0] += __shfl_xor(w[0], 32, 64);
        if (lane < 16) {   // lane l holds the total of value index (bit-reversed l): 4 * step + component
          const int vi = ((lane & 1) << 3) | ((lane & 2) << 1) | ((lane & 4) >> 1) | ((lane & 8) >> 3);
          reinterpret_cast<float *>(&part[wave][chi * NVP + 4 * g])[vi] = w[0];
        }
      }
    }
  }
  __syncthreads();
  if ((int)threadIdx.x < NST) {
    const int it = threadIdx.x;
    float4 s = part[0][it];
    for (int w2 = 1; w2 < nw; w2++) { const float4 t = part[w2][it]; s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w; }
    const int cpar = A >= out.Vh, xc = A - cpar * out.Vh, c0 = 2 * it;   // = chi * NVEC + 2 * vp
    float *ob = out.v[cpar];
    ob[((size_t)c0 * out.stride + xc) * 2] = s.x; ob[((size_t)c0 * out.stride + xc) * 2 + 1] = s.y;
    ob[((size_t)(c0 + 1) * out.stride + xc) * 2] = s.z; ob[((size_t)(c0 + 1) * out.stride + xc) * 2 + 1] = s.w;
  }
}

// ---- FOUR SOURCES per pass over V (invertMultiSrcQuda, block_solver.cpp): the restrictor / prolongator of the fine level are pure streams of V
// (2304 B per fine site against 96 B of vector), so a lockstep multi-source cycle that restricts its sources one after the other reads V once per
// source.  Same structure as restrict_stream_kernel / prolong_kernel — three steps of V requests in flight per thread, fenced — with the V entries
// of a step meeting the site's components of four sources; the reduce-scatter butterfly that summed four STEPS of one source there sums one step
// of four SOURCES here (16 values -> lanes 0..15). ----
struct Src4 { FineVec f[4]; CoarseVec c[4]; };
// BLK: the four fine vectors are columns col0 .. col0 + 3 of a pair-major block field of ONE parity (block.h; the multi-source smoother keeps its
// residuals and solutions there): word (x 6 + k) nrhs + column holds components 2k, 2k + 1 — no unpacking into fields in front of the restrictor, no
// packing behind the prolongator
struct Blk4 { float4 *panel; int nrhs, col0, parity, Vh, accumulate; };
template <int NCF, int NVEC, int NV, bool BLK = false>
__global__ void __launch_bounds__(256) restrict_stream4_kernel(Src4 a, const void *V, const int *block_to_fine, int blockVol, MaskArg mask, AggMap amap, Blk4 blk) {
  constexpr int K = 4 * NCF, KH = K / 2, NVP = NVEC / 2, NST = 2 * NVP;
  typedef typename VRaw<false>::type raw_t;
  __shared__ float4 part[4][NST][4];   // [wave][chirality * NVP + vector pair][source]
  const int A = aggregate_of_block(amap), b = threadIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  float2 r[4][K];
#pragma unroll
  for (int s = 0; s < 4; s++)
#pragma unroll
    for (int k = 0; k < K; k++) r[s][k] = make_float2(0.f, 0.f);
  bool have = false;
  if (b < blockVol && mask_keep(mask, b)) {
    const int f = block_to_fine[(size_t)A * blockVol + b];
    if constexpr (BLK) {
      const int parity = f >= blk.Vh, x = f - parity * blk.Vh;
      if (parity == blk.parity) {
        have = true;
        const float4 *p = blk.panel + (size_t)x * 6 * blk.nrhs + blk.col0;
#pragma unroll
        for (int kk = 0; kk < K / 2; kk++)
#pragma unroll
          for (int s = 0; s < 4; s++) {
            const float4 w = p[kk * blk.nrhs + s];
            r[s][2 * kk] = make_float2(w.x, w.y); r[s][2 * kk + 1] = make_float2(w.z, w.w);
          }
      }
    } else {
      const int parity = f >= a.f[0].Vh, x = f - parity * a.f[0].Vh;
      if (a.f[0].v[parity]) {
        have = true;
#pragma unroll
        for (int s = 0; s < 4; s++) load_fine_site<NV, K>(r[s], a.f[s].v[parity], a.f[s].stride, x);
      }
    }
  }
  if (__builtin_amdgcn_ballot_w64(have) == 0) {   // wave-uniform
    for (int e = lane; e < NST * 4; e += 64) part[wave][e >> 2][e & 3] = make_float4(0.f, 0.f, 0.f, 0.f);
  } else {
    const int bl = b < blockVol ? b : 0;
#pragma unroll
    for (int chi = 0; chi < 2; chi++) {
      raw_t w0[KH], w1[KH], w2[KH];
      auto vload = [&](raw_t *dst, int vpl) {
        if (vpl >= NVP) return;
#pragma unroll
        for (int kk = 0; kk < KH; kk++) dst[kk] = load_v_raw<false>(V, (((size_t)A * K + chi * KH + kk) * NVP + vpl) * blockVol + bl);
      };
      vload(w0, 0); vload(w1, 1); vload(w2, 2);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int vp = 0; vp < NVP; vp++) {
        const int ph = vp % 3;
        float w[16];
#pragma unroll
        for (int s = 0; s < 4; s++) {
          pkf2 accA = {0.f, 0.f}, accB = {0.f, 0.f};   // conj(V) r for the two vectors of the pair: two packed multiply-adds each (the 4 x 6 x 8 scalar ones per step held the kernel at 0.53-0.59)
#pragma unroll
          for (int kk = 0; kk < KH; kk++) {
            const float4 v = v_unpack(ph == 0 ? w0[kk] : (ph == 1 ? w1[kk] : w2[kk]));
            const pkf2 rr = {r[s][chi * KH + kk].x, r[s][chi * KH + kk].y};
            accA = pk::cmacc(accA, (pkf2){v.x, v.y}, rr);
            accB = pk::cmacc(accB, (pkf2){v.z, v.w}, rr);
          }
          w[4 * s] = accA.x; w[4 * s + 1] = accA.y; w[4 * s + 2] = accB.x; w[4 * s + 3] = accB.y;
        }
        __builtin_amdgcn_sched_barrier(0);
        if (ph == 0) vload(w0, vp + 3); else if (ph == 1) vload(w1, vp + 3); else vload(w2, vp + 3);   // the buffer just used: three steps ahead
        __builtin_amdgcn_sched_barrier(0);
#define QA_BFLY(HALFN, M)                                                                  \
        {                                                                                  \
          const bool up = (lane & M) != 0;                                                 \
          _Pragma("unroll") for (int j = 0; j < HALFN; j++) {                              \
            const float keep = up ? w[HALFN + j] : w[j], give = up ? w[j] : w[HALFN + j]; \
            w[j] = keep + __shfl_xor(give, M, 64);                                         \
          }                                                                                \
        }
        QA_BFLY(8, 1) QA_BFLY(4, 2) QA_BFLY(2, 4) QA_BFLY(1, 8)
#undef QA_BFLY
        w[0] += __shfl_xor(w[0], 16, 64);
        w[0] += __shfl_xor(w[0], 32, 64);
        if (lane < 16) {   // lane l holds the total of value index (bit-reversed l) = 4 * source + component
          const int vi = ((lane & 1) << 3) | ((lane & 2) << 1) | ((lane & 4) >> 1) | ((lane & 8) >> 3);
          reinterpret_cast<float *>(&part[wave][chi * NVP + vp][0])[vi] = w[0];
        }
      }
    }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < NST * 4; e += blockDim.x) {
    const int it = e >> 2, s = e & 3;
    float4 t = part[0][it][s];
    for (int w2 = 1; w2 < nw; w2++) { const float4 u = part[w2][it][s]; t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; }
    const CoarseVec &o = a.c[s];
    const int cpar = A >= o.Vh, xc = A - cpar * o.Vh, c0 = 2 * it;   // = chi * NVEC + 2 * vp
    float *ob = o.v[cpar];
    ob[((size_t)c0 * o.stride + xc) * 2] = t.x; ob[((size_t)c0 * o.stride + xc) * 2 + 1] = t.y;
    ob[((size_t)(c0 + 1) * o.stride + xc) * 2] = t.z; ob[((size_t)(c0 + 1) * o.stride + xc) * 2 + 1] = t.w;
  }
}
template <int NCF, int NVEC, int NV, bool BLK = false>
__global__ void __launch_bounds__(256) prolong4_kernel(Src4 a, const void *V, const int *block_to_fine, int blockVol, AggMap amap, Blk4 blk) {
  constexpr int K = 4 * NCF, KH = K / 2, NVP = NVEC / 2;
  typedef typename VRaw<false>::type raw_t;
  __shared__ float2 xc_s[4][2 * NVEC];
  const int A = aggregate_of_block(amap), b = threadIdx.x;
  for (int e = threadIdx.x; e < 4 * 2 * NVEC; e += blockDim.x) {
    const int s = e / (2 * NVEC), j = e - s * 2 * NVEC;
    const CoarseVec &in = a.c[s];
    const int cpar = A >= in.Vh, xc = A - cpar * in.Vh;
    const float *p = in.v[cpar] + ((size_t)j * in.stride + xc) * 2;
    xc_s[s][j] = make_float2(p[0], p[1]);
  }
  __syncthreads();
  if (b >= blockVol) return;
  const int f = block_to_fine[(size_t)A * blockVol + b];
  const int fVh = BLK ? blk.Vh : a.f[0].Vh;
  const int parity = f >= fVh, x = f - parity * fVh;
  if (BLK ? parity != blk.parity : !a.f[0].v[parity]) return;   // this parity is absent from single-parity output fields
  pkf2 acc[4][K];
#pragma unroll
  for (int s = 0; s < 4; s++)
#pragma unroll
    for (int k = 0; k < K; k++) acc[s][k] = (pkf2){0.f, 0.f};
#pragma unroll
  for (int chi = 0; chi < 2; chi++) {
    raw_t w0[KH], w1[KH], w2[KH];
    auto vload = [&](raw_t *dst, int vpl) {
      if (vpl >= NVP) return;
#pragma unroll
      for (int kk = 0; kk < KH; kk++) dst[kk] = load_v_raw<false>(V, (((size_t)A * K + chi * KH + kk) * NVP + vpl) * blockVol + b);
    };
    vload(w0, 0); vload(w1, 1); vload(w2, 2);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int vp = 0; vp < NVP; vp++) {
      const int ph = vp % 3;
#pragma unroll
      for (int s = 0; s < 4; s++) {
        const float2 d0 = xc_s[s][chi * NVEC + 2 * vp], d1 = xc_s[s][chi * NVEC + 2 * vp + 1];
        const pkf2 c0 = {d0.x, d0.y}, c1 = {d1.x, d1.y};
#pragma unroll
        for (int kk = 0; kk < KH; kk++) {
          const float4 w = v_unpack(ph == 0 ? w0[kk] : (ph == 1 ? w1[kk] : w2[kk]));
          acc[s][chi * KH + kk] = pk::cmac(pk::cmac(acc[s][chi * KH + kk], (pkf2){w.x, w.y}, c0), (pkf2){w.z, w.w}, c1);   // V c: four packed multiply-adds
        }
      }
      // pin the sums here (as prolong_kernel): without it the multiply-adds sink below the last fence and every load stays live
#pragma unroll
      for (int s = 0; s < 4; s++)
#pragma unroll
        for (int kk = 0; kk < KH; kk++) asm volatile("" : "+v"(acc[s][chi * KH + kk]));
      __builtin_amdgcn_sched_barrier(0);
      if (ph == 0) vload(w0, vp + 3); else if (ph == 1) vload(w1, vp + 3); else vload(w2, vp + 3);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if constexpr (BLK) {
    float4 *p = blk.panel + (size_t)x * 6 * blk.nrhs + blk.col0;
#pragma unroll
    for (int kk = 0; kk < K / 2; kk++)
#pragma unroll
      for (int s = 0; s < 4; s++) {
        float4 o = make_float4(acc[s][2 * kk].x, acc[s][2 * kk].y, acc[s][2 * kk + 1].x, acc[s][2 * kk + 1].y);
        if (blk.accumulate) { const float4 w = p[kk * blk.nrhs + s]; o.x += w.x; o.y += w.y; o.z += w.z; o.w += w.w; }
        p[kk * blk.nrhs + s] = o;
      }
    return;
  }
#pragma unroll
  for (int s = 0; s < 4; s++) {
    float *base = a.f[s].v[parity];
    if (NV == 4) {
#pragma unroll
      for (int k = 0; k < K; k += 2) *reinterpret_cast<float4 *>(base + fidx<NV>(a.f[s].stride, x, k)) = make_float4(acc[s][k].x, acc[s][k].y, acc[s][k + 1].x, acc[s][k + 1].y);
    } else {
#pragma unroll
      for (int k = 0; k < K; k++) *reinterpret_cast<float2 *>(base + fidx<NV>(a.f[s].stride, x, k)) = make_float2(acc[s][k].x, acc[s][k].y);
    }
  }
}

// ---- four right-hand sides per pass over V (Galerkin construction of the first coarse level: the 8 single-direction hops of
// one probe are restricted in two launches instead of eight; V is the whole cost of a restriction) ----
struct Multi4 {
  CoarseVec out[4], out2[4];
  FineVec in[4];
  int dir[4];
};
__device__ __forceinline__ bool mask_outside_dir(const MaskArg &m, int dir, int b) {
  const int mu = dir >> 1, fwd = !(dir & 1);
  int y[4];
  block_coords(y, m, b);
  return !m.single[mu] && (fwd ? y[mu] == m.bs[mu] - 1 : y[mu] == 0);
}
// every wave keeps its partial sums of ALL (chirality, vector pair) steps in its own LDS rows, so the steps run back to back
// without a barrier (24 steps x 2 barriers made this kernel 2.5x slower than the V stream); one barrier, then the block sum
template <int NSF, int NCF, int NVEC, int NV>
__global__ void __launch_bounds__(256) restrict4_kernel(Multi4 a, const float4 *V, const int *block_to_fine, int blockVol, int spin_bs, MaskArg mask, AggMap amap) {
  constexpr int K = NSF * NCF, NIT = NVEC;   // steps = 2 chiralities x NVEC / 2 vector pairs
  __shared__ float4 part[4][NIT][8];          // [wave][step][rhs*2 + (0 leaving, 1 staying)]
  const int A = aggregate_of_block(amap), b = threadIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  const bool site = b < blockVol;
  float2 r[4][K];
  bool outside[4];
  if (site) {
    const int f = block_to_fine[(size_t)A * blockVol + b];
    const int parity = f >= a.in[0].Vh, x = f - parity * a.in[0].Vh;
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const float *base = a.in[q].v[parity];
      outside[q] = mask_outside_dir(mask, a.dir[q], b);
load_fine_site<NV, K>(r[q], base, a.in[q].stride, x);
    }
  }
  // The two chiralities are separate (unrolled) loops so that the K / 2 rows of a step are compile-time register indices, and the
  // V entries of step vp + 1 are requested before the cross-lane reduction of step vp: one step's loads -> wait -> sums -> shuffles
  // in series left the kernel latency-bound at two waves per SIMD (10 ms per pass over V against 4.4 ms for the plain restrictor).
  constexpr int KH = K / 2;
#pragma unroll
  for (int chi = 0; chi < 2; chi++) {
  // Three buffers of V entries, statically named and refilled in place right after their use, fenced against the scheduler (which
  // would sink every request down to its use): three steps of requests in flight per thread
  float4 w0[KH], w1[KH], w2[KH];
  // unconditional loads (a thread beyond the aggregate reads entry 0 and its sums are masked out below): a load under a per-lane
  // condition becomes its own basic block, and across blocks the wait-count bookkeeping falls back to vmcnt(0) again
  const int bl = site ? b : 0;
  auto vload = [&](float4 *dst, int vpl) {
    if (vpl >= NVEC / 2) return;
#pragma unroll
    for (int kk = 0; kk < KH; kk++) dst[kk] = load_v<false>(V, (((size_t)A * K + chi * KH + kk) * (NVEC / 2) + vpl) * blockVol + bl);
  };
  vload(w0, 0); vload(w1, 1); vload(w2, 2);
  __builtin_amdgcn_sched_barrier(0);
  // fully unrolled: with a loop the wait-count bookkeeping of the compiler merges to s_waitcnt vmcnt(0) at the loop header and at
  // every use — the three buffers were drained at each step, which is why neither they nor anything else moved this kernel
#pragma unroll
  for (int vp3 = 0; vp3 < NVEC / 2; vp3 += 3) {
#pragma unroll
  for (int ph = 0; ph < 3; ph++) {
    const int vp = vp3 + ph;
    if (vp >= NVEC / 2) break;
    const int it = chi * (NVEC / 2) + vp;
    float4 wc[KH];
#pragma unroll
    for (int kk = 0; kk < KH; kk++) wc[kk] = ph == 0 ? w0[kk] : (ph == 1 ? w1[kk] : w2[kk]);
    float4 acc[4];
#pragma unroll
    for (int q = 0; q < 4; q++) acc[q] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int kk = 0; kk < KH; kk++) {
      const float4 w = wc[kk];
      const int k = chi * KH + kk;
#pragma unroll
      for (int q = 0; q < 4; q++) {
        acc[q].x += w.x * r[q][k].x + w.y * r[q][k].y; acc[q].y += w.x * r[q][k].y - w.y * r[q][k].x;
        acc[q].z += w.z * r[q][k].x + w.w * r[q][k].y; acc[q].w += w.z * r[q][k].y - w.w * r[q][k].x;
      }
    }
    // the buffer just used takes the entries of three steps ahead
    __builtin_amdgcn_sched_barrier(0);
    if (ph == 0) vload(w0, vp + 3); else if (ph == 1) vload(w1, vp + 3); else vload(w2, vp + 3);
    __builtin_amdgcn_sched_barrier(0);
    // wave sum of the 32 partial reals (4 right-hand sides x {leaving, staying} x float4) by a reduce-scatter butterfly: at every
    // stage a lane keeps one half of its values and adds the partner's copy of that half, so the five xor stages move
    // 16 + 8 + 4 + 2 + 1 values instead of 32 each, one more adds the two half-waves: 32 cross-lane moves instead of 192
    // (the shuffle chains, not the V stream, held this kernel at 2.3 TB/s)
    float w[32];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const bool so = site && outside[q], si = site && !outside[q];
      w[8 * q + 0] = so ? acc[q].x : 0.f; w[8 * q + 1] = so ? acc[q].y : 0.f; w[8 * q + 2] = so ? acc[q].z : 0.f; w[8 * q + 3] = so ? acc[q].w : 0.f;
      w[8 * q + 4] = si ? acc[q].x : 0.f; w[8 * q + 5] = si ? acc[q].y : 0.f; w[8 * q + 6] = si ? acc[q].z : 0.f; w[8 * q + 7] = si ? acc[q].w : 0.f;
    }
    // (written out stage by stage: with the stage as a loop variable the register array was indexed dynamically, 1900 selects per step)
#define QA_BFLY(HALF, M)                                                                 \
    {                                                                                    \
      const bool up = (lane & M) != 0;                                                   \
      _Pragma("unroll") for (int j = 0; j < HALF; j++) {                                 \
        const float keep = up ? w[HALF + j] : w[j], give = up ? w[j] : w[HALF + j];     \
        w[j] = keep + __shfl_xor(give, M, 64);                                           \
      }                                                                                  \
    }
    QA_BFLY(16, 1) QA_BFLY(8, 2) QA_BFLY(4, 4) QA_BFLY(2, 8) QA_BFLY(1, 16)
#undef QA_BFLY
    w[0] += __shfl_xor(w[0], 32, 64);
    // lane l < 32 now holds the total of value index sum_s bit_s(l) * (16 >> s)
    if (lane < 32) {
      const int vi = ((lane & 1) << 4) | ((lane & 2) << 2) | (lane & 4) | ((lane & 8) >> 2) | ((lane & 16) >> 4);
      reinterpret_cast<float *>(&part[wave][it][0])[vi] = w[0];
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  }
  }
  __syncthreads();
  const int cpar = A >= a.out[0].Vh, xc = A - cpar * a.out[0].Vh;
  for (int e = threadIdx.x; e < NIT * 8; e += blockDim.x) {
    const int it = e >> 3, o8 = e & 7;
    float4 s = part[0][it][o8];
    for (int w2 = 1; w2 < nw; w2++) { const float4 t = part[w2][it][o8]; s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w; }
    const int q = o8 >> 1;
    const CoarseVec &o = (o8 & 1) ? a.out2[q] : a.out[q];
    float *ob = o.v[cpar];
    const int chi = it / (NVEC / 2), vp = it - chi * (NVEC / 2), c0 = chi * NVEC + 2 * vp;
    ob[((size_t)c0 * o.stride + xc) * 2] = s.x; ob[((size_t)c0 * o.stride + xc) * 2 + 1] = s.y;
    ob[((size_t)(c0 + 1) * o.stride + xc) * 2] = s.z; ob[((size_t)(c0 + 1) * o.stride + xc) * 2 + 1] = s.w;
  }
}

// phi = P e_j for the coarse unit vector j = (chirality, vector) at every coarse site: column j of V, copied out of the
// aggregate-major layout (1/12 of V is touched instead of the whole of it by a prolongation)
template <int NSF, int NCF, int NV>
__global__ void column_kernel(FineVec out, const float4 *V, const int *block_to_fine, int blockVol, int spin_bs, int NVEC, int j, long total) {
  constexpr int K = NSF * NCF;
  const long t = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (t >= total) return;
  const long A = t / blockVol;
  const int b = (int)(t - A * blockVol);
  const int f = block_to_fine[t];
  const int parity = f >= out.Vh, x = f - parity * out.Vh;
  float *base = out.v[parity];
  const int chi = j / NVEC, v = j - chi * NVEC;
  float2 c[K];
#pragma unroll
  for (int k = 0; k < K; k++) {
    c[k] = make_float2(0.f, 0.f);
    if ((k / NCF) / spin_bs == chi) {
      const float4 w = V[(((size_t)A * K + k) * (NVEC / 2) + (v >> 1)) * blockVol + b];
      c[k] = (v & 1) ? make_float2(w.z, w.w) : make_float2(w.x, w.y);
    }
  }
  if (NV == 4 && K % 2 == 0) {   // two components per plane entry: one 16-byte store instead of four 4-byte ones
#pragma unroll
    for (int m = 0; m < K / 2; m++) reinterpret_cast<float4 *>(base)[(size_t)m * out.stride + x] = make_float4(c[2 * m].x, c[2 * m].y, c[2 * m + 1].x, c[2 * m + 1].y);
  } else {
#pragma unroll
    for (int k = 0; k < K; k++) { const size_t i = fidx<NV>(out.stride, x, k); base[i] = c[k].x; base[i + 1] = c[k].y; }
  }
}

// ---- small aggregates (blockVol <= 32, the coarse levels: 2^4 = 16 sites).  One thread per site leaves most of a wave idle and
// runs the (chirality, vector pair) iterations one after the other behind block-wide reductions — 0.28 ms for a 37 MB
// transfer.  Here a wave holds 64 / GS groups of GS lanes (GS = blockVol rounded up to a power of two), every group takes its
// own (chi, vp) iteration for the whole aggregate and reduces with GS-wide shuffles: no LDS, no barriers, all iterations
// of an aggregate in flight at once across the 8 waves of its work-group. ----
template <int NSF, int NCF, int NVEC, int NV, bool DUAL, bool HALF>
__global__ void __launch_bounds__(512) restrict_small_kernel(CoarseVec out, CoarseVec out2, FineVec in, const void *V, const int *block_to_fine, int blockVol, int GS,
                                                             int spin_bs, MaskArg mask) {
  constexpr int K = NSF * NCF, NIT = NVEC;   // iterations = 2 chiralities x NVEC/2 vector pairs
  const int A = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int groupsPerWave = 64 / GS, b = lane % GS, grp = lane / GS;
  const int slot = wave * groupsPerWave + grp, nslots = (blockDim.x >> 6) * groupsPerWave;
  const bool site = b < blockVol;
  const bool active = site && (DUAL || mask_keep(mask, b));
  const bool outside = DUAL && site && mask_outside(mask, b);
  float2 r[K];
  bool have = false;
  if (active) {
    const int f = block_to_fine[(size_t)A * blockVol + b];
    const int parity = f >= in.Vh, x = f - parity * in.Vh;
    const float *base = in.v[parity];
    if (base) {
      have = true;
load_fine_site<NV, K>(r, base, in.stride, x);
    }
  }
  const int cpar = A >= out.Vh, xc = A - cpar * out.Vh;
  float *ob = out.v[cpar], *ob2 = DUAL ? out2.v[cpar] : nullptr;
  for (int it = slot; it < NIT; it += nslots) {
    const int chi = it / (NVEC / 2), vp = it - chi * (NVEC / 2);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (have) {
#pragma unroll
      for (int k = 0; k < K; k++) {
        if ((k / NCF) / spin_bs != chi) continue;
        const float4 w = load_v<HALF>(V, (((size_t)A * K + k) * (NVEC / 2) + vp) * blockVol + b);
        acc.x += w.x * r[k].x + w.y * r[k].y; acc.y += w.x * r[k].y - w.y * r[k].x;
        acc.z += w.z * r[k].x + w.w * r[k].y; acc.w += w.z * r[k].y - w.w * r[k].x;
      }
    }
    float4 so = acc, si = make_float4(0.f, 0.f, 0.f, 0.f);
    if (DUAL) { if (!outside) { si = acc; so = make_float4(0.f, 0.f, 0.f, 0.f); } }
    for (int off = GS >> 1; off > 0; off >>= 1) {
      so.x += __shfl_down(so.x, off, GS); so.y += __shfl_down(so.y, off, GS); so.z += __shfl_down(so.z, off, GS); so.w += __shfl_down(so.w, off, GS);
      if (DUAL) { si.x += __shfl_down(si.x, off, GS); si.y += __shfl_down(si.y, off, GS); si.z += __shfl_down(si.z, off, GS); si.w += __shfl_down(si.w, off, GS); }
    }
    if (b == 0) {
      const int c0 = chi * NVEC + 2 * vp;
      ob[((size_t)c0 * out.stride + xc) * 2] = so.x; ob[((size_t)c0 * out.stride + xc) * 2 + 1] = so.y;
      ob[((size_t)(c0 + 1) * out.stride + xc) * 2] = so.z; ob[((size_t)(c0 + 1) * out.stride + xc) * 2 + 1] = so.w;
      if (DUAL) {
        ob2[((size_t)c0 * out2.stride + xc) * 2] = si.x; ob2[((size_t)c0 * out2.stride + xc) * 2 + 1] = si.y;
        ob2[((size_t)(c0 + 1) * out2.stride + xc) * 2] = si.z; ob2[((size_t)(c0 + 1) * out2.stride + xc) * 2 + 1] = si.w;
      }
    }
  }
}

// prolongator for small aggregates: the K fine components of a site are spread over the groups / waves the same way
template <int NSF, int NCF, int NVEC, int NV, bool HALF>
__global__ void __launch_bounds__(512) prolong_small_kernel(FineVec out, CoarseVec in, const void *V, const int *block_to_fine, int blockVol, int GS, int spin_bs) {
  constexpr int K = NSF * NCF;
  __shared__ float2 xc_s[2 * NVEC];
  const int A = blockIdx.x;
  const int cpar = A >= in.Vh, xc = A - cpar * in.Vh;
  for (int j = threadIdx.x; j < 2 * NVEC; j += blockDim.x) {
    const float *p = in.v[cpar] + ((size_t)j * in.stride + xc) * 2;
    xc_s[j] = make_float2(p[0], p[1]);
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int groupsPerWave = 64 / GS, b = lane % GS, grp = lane / GS;
  const int slot = wave * groupsPerWave + grp, nslots = (blockDim.x >> 6) * groupsPerWave;
  if (b >= blockVol) return;
  const int f = block_to_fine[(size_t)A * blockVol + b];
  const int parity = f >= out.Vh, x = f - parity * out.Vh;
  float *base = out.v[parity];
  if (!base) return;
  for (int k = slot; k < K; k += nslots) {
    const int chi = (k / NCF) / spin_bs;
    float re = 0.f, im = 0.f;
#pragma unroll 4
    for (int vp = 0; vp < NVEC / 2; vp++) {
      const float4 w = load_v<HALF>(V, (((size_t)A * K + k) * (NVEC / 2) + vp) * blockVol + b);
      const float2 c0 = xc_s[chi * NVEC + 2 * vp], c1 = xc_s[chi * NVEC + 2 * vp + 1];
      re += w.x * c0.x - w.y * c0.y + w.z * c1.x - w.w * c1.y;
      im += w.x * c0.y + w.y * c0.x + w.z * c1.y + w.w * c1.x;
    }
    const size_t i = fidx<NV>(out.stride, x, k);
    base[i] = re; base[i + 1] = im;
  }
}

// ---- prolongator ----
template <int NSF, int NCF, int NVEC, int NV, bool HALF = false>
__global__ void __launch_bounds__(512) prolong_kernel(FineVec out, CoarseVec in, const void *V, const int *block_to_fine, int blockVol, int spin_bs, AggMap amap) {
  constexpr int K = NSF * NCF;
  __shared__ float2 xc_s[2 * NVEC];
  const int A = aggregate_of_block(amap), b = threadIdx.x;
  const int cpar = A >= in.Vh, xc = A - cpar * in.Vh;
  for (int j = threadIdx.x; j < 2 * NVEC; j += blockDim.x) {
    const float *p = in.v[cpar] + ((size_t)j * in.stride + xc) * 2;
    xc_s[j] = make_float2(p[0], p[1]);
  }
  __syncthreads();
  if (b >= blockVol) return;
  const int f = block_to_fine[(size_t)A * blockVol + b];
  const int parity = f >= out.Vh, x = f - parity * out.Vh;
  float *base = out.v[parity];
  if (!base) return;   // this parity is absent from a single-parity output field
  // vector pair outermost: its two coarse coefficients are read from LDS once and meet the K / 2 spin-colour rows of their
  // chirality (k-outer re-read them for every row: 288 LDS reads per thread next to 144 V loads); 16-byte stores where the
  // field order pairs two components (FLOAT4: spin-colour 2 m, 2 m + 1 share a plane entry)
  float2 acc[K];
#pragma unroll
  for (int k = 0; k < K; k++) acc[k] = make_float2(0.f, 0.f);
  if constexpr (NSF == 4) {
    // fine level: three steps of V requests in flight per thread (raw register images, converted at their use), every step fully
    // unrolled and fenced — as a rolled loop the 6 x unroll loads of an iteration were drained before the next ones were issued
    constexpr int KH = K / 2, NVP = NVEC / 2;
    typedef typename VRaw<HALF>::type raw_t;
#pragma unroll
    for (int chi = 0; chi < 2; chi++) {
      raw_t w0[KH], w1[KH], w2[KH];
      auto vload = [&](raw_t *dst, int vpl) {
        if (vpl >= NVP) return;
#pragma unroll
        for (int kk = 0; kk < KH; kk++) dst[kk] = load_v_raw<HALF>(V, (((size_t)A * K + chi * KH + kk) * NVP + vpl) * blockVol + b);
      };
      vload(w0, 0); vload(w1, 1); vload(w2, 2);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int vp = 0; vp < NVP; vp++) {
        const int ph = vp % 3;
        const float2 c0 = xc_s[chi * NVEC + 2 * vp], c1 = xc_s[chi * NVEC + 2 * vp + 1];
#pragma unroll
        for (int kk = 0; kk < KH; kk++) {
          const float4 w = v_unpack(ph == 0 ? w0[kk] : (ph == 1 ? w1[kk] : w2[kk]));
          acc[chi * KH + kk].x += w.x * c0.x - w.y * c0.y + w.z * c1.x - w.w * c1.y;
          acc[chi * KH + kk].y += w.x * c0.y + w.y * c0.x + w.z * c1.y + w.w * c1.x;
        }
        // pin the sums here: without it the multiply-adds sink below the last fence and every load stays live (256 registers + scratch)
#pragma unroll
        for (int kk = 0; kk < KH; kk++) asm volatile("" : "+v"(acc[chi * KH + kk].x), "+v"(acc[chi * KH + kk].y));
        __builtin_amdgcn_sched_barrier(0);
        if (ph == 0) vload(w0, vp + 3); else if (ph == 1) vload(w1, vp + 3); else vload(w2, vp + 3);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  } else {
#pragma unroll
  for (int chi = 0; chi < 2; chi++) {
#pragma unroll 2
    for (int vp = 0; vp < NVEC / 2; vp++) {
      const float2 c0 = xc_s[chi * NVEC + 2 * vp], c1 = xc_s[chi * NVEC + 2 * vp + 1];
#pragma unroll
      for (int k = 0; k < K; k++) {
        if (k / (K / 2) != chi) continue;   // two chiralities whatever the spin blocking: rows [0, K/2) and [K/2, K) (= (k / NCF) / spin_bs)
        const float4 w = load_v<HALF>(V, (((size_t)A * K + k) * (NVEC / 2) + vp) * blockVol + b);
        acc[k].x += w.x * c0.x - w.y * c0.y + w.z * c1.x - w.w * c1.y;
        acc[k].y += w.x * c0.y + w.y * c0.x + w.z * c1.y + w.w * c1.x;
      }
    }
  }
  }
  if (NV == 4) {
#pragma unroll
    for (int k = 0; k < K; k += 2) *reinterpret_cast<float4 *>(base + fidx<NV>(out.stride, x, k)) = make_float4(acc[k].x, acc[k].y, acc[k + 1].x, acc[k + 1].y);
  } else {
#pragma unroll
    for (int k = 0; k < K; k++) *reinterpret_cast<float2 *>(base + fidx<NV>(out.stride, x, k)) = acc[k];
  }
}

// ---- V fill + block Gram-Schmidt ----
struct VecList { const float *ev[kMaxVec]; const float *od[kMaxVec]; };

template <int NV> __global__ void fillv_kernel(float *V, VecList B, int stride, int Vh, const int *block_to_fine, int blockVol, int K, int nvec, long total) {
  const long t = blockIdx.x * (long)blockDim.x + threadIdx.x;  // over (A, b)
  if (t >= total) return;
  const long A = t / blockVol;
  const int b = (int)(t - A * blockVol);
  const int f = block_to_fine[t];
  const int parity = f >= Vh, x = f - parity * Vh;
  if (NV == 4 && K % 2 == 0) {
    // 16-byte reads (two components of one vector share a plane entry) and 16-byte writes (the two vectors of a pair share a V entry)
    float4 *V4 = reinterpret_cast<float4 *>(V);
    for (int vp = 0; vp < nvec / 2; vp++) {
      const float4 *b0 = reinterpret_cast<const float4 *>(parity ? B.od[2 * vp] : B.ev[2 * vp]);
      const float4 *b1 = reinterpret_cast<const float4 *>(parity ? B.od[2 * vp + 1] : B.ev[2 * vp + 1]);
      for (int m = 0; m < K / 2; m++) {
        const float4 u = b0[(size_t)m * stride + x], w = b1[(size_t)m * stride + x];
        V4[(((size_t)A * K + 2 * m) * (nvec / 2) + vp) * blockVol + b] = make_float4(u.x, u.y, w.x, w.y);
        V4[(((size_t)A * K + 2 * m + 1) * (nvec / 2) + vp) * blockVol + b] = make_float4(u.z, u.w, w.z, w.w);
      }
    }
    return;
  }
  for (int v = 0; v < nvec; v++) {
    const float *base = parity ? B.od[v] : B.ev[v];
    for (int k = 0; k < K; k++) {
      const size_t i = fidx<NV>(stride, x, k);
      const size_t o = ((((size_t)A * K + k) * (nvec / 2) + v / 2) * blockVol + b) * 4 + (v & 1) * 2;
      V[o] = base[i]; V[o + 1] = base[i + 1];
    }
  }
}

__device__ __forceinline__ double2 block_sum2d(double2 v, double2 *lds) {
  for (int off = 32; off > 0; off >>= 1) { v.x += __shfl_down(v.x, off, 64); v.y += __shfl_down(v.y, off, 64); }
  const int wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) lds[wave] = v;
  __syncthreads();
  double2 r = lds[0];
  for (int w = 1; w < nw; w++) { r.x += lds[w].x; r.y += lds[w].y; }
  return r;
}

// modified Gram-Schmidt over the nvec vectors of one (aggregate, chirality) block, sums in fp64
// (reference blockGramSchmidt, lib/transfer_util.cu:328-363)
__global__ void block_gs_kernel(float *V, int blockVol, int K, int ncf, int spin_bs, int nvec, const int *only) {
  if (only && !only[blockIdx.x]) return;   // fall-back of the CholeskyQR kernel: just the blocks it flagged
  __shared__ double2 lds[16];
  const int A = blockIdx.x >> 1, chi = blockIdx.x & 1;
  const int Kc = K / 2, n = Kc * blockVol;
  // element e of this block -> (k, b)
  auto addr = [&](int e, int v) -> size_t {
    const int kk = e / blockVol, b = e - kk * blockVol;
    // the kk-th fine spin-colour index whose chirality is chi
    const int nsc = spin_bs * ncf;             // spin-colour values per chirality are contiguous in k = s*ncf + c
    const int k = chi * nsc + kk;
    return ((((size_t)A * K + k) * (nvec / 2) + v / 2) * blockVol + b) * 4 + (v & 1) * 2;
  };
  for (int jc = 0; jc < nvec; jc++) {
    for (int ic = 0; ic < jc; ic++) {
      double2 dot = make_double2(0.0, 0.0);
      for (int e = threadIdx.x; e < n; e += blockDim.x) {
        const size_t ai = addr(e, ic), aj = addr(e, jc);
        const double ar = V[ai], aim = V[ai + 1], br = V[aj], bim = V[aj + 1];
        dot.x += ar * br + aim * bim; dot.y += ar * bim - aim * br;
      }
      dot = block_sum2d(dot, lds);
      for (int e = threadIdx.x; e < n; e += blockDim.x) {
        const size_t ai = addr(e, ic), aj = addr(e, jc);
        const double ar = V[ai], aim = V[ai + 1];
        V[aj] = (float)(V[aj] - (dot.x * ar - dot.y * aim));
        V[aj + 1] = (float)(V[aj + 1] - (dot.x * aim + dot.y * ar));
      }
      __syncthreads();
    }
    double2 nrm = make_double2(0.0, 0.0);
    for (int e = threadIdx.x; e < n; e += blockDim.x) { const size_t aj = addr(e, jc); nrm.x += (double)V[aj] * V[aj] + (double)V[aj + 1] * V[aj + 1]; }
    nrm = block_sum2d(nrm, lds);
    const float scale = nrm.x > 0.0 ? (float)(1.0 / sqrt(nrm.x)) : 0.f;
    for (int e = threadIdx.x; e < n; e += blockDim.x) { const size_t aj = addr(e, jc); V[aj] *= scale; V[aj + 1] *= scale; }
    __syncthreads();
  }
}

// ---- CholeskyQR: the same block orthonormalisation in two passes over the block instead of nvec (nvec - 1) / 2 -----------------
// Q R = V with R upper triangular and a positive diagonal is unique, so Q = V R^-1 with R from the Cholesky factorisation of the
// Gram matrix G = V^dagger V (= R^dagger R) IS the Gram-Schmidt result — but G needs one sweep over the block (all 300 inner
// products of 24 vectors at once, fp64 sums, the block staged through LDS in chunks of 256 elements) and V R^-1 a second one,
// where modified Gram-Schmidt makes 276 dependent sweeps with two block reductions each (0.44 - 0.6 s at 48^3 x 96).  Run twice
// (CholeskyQR2): the second round removes the orthogonality the first one loses to the conditioning of the block — as long as
// eps cond(V)^2 < 1 with the eps of the Gram matrix, which is fp32 here (V is fp32 and a chunk's 256-term sum is fp32: ~1e-6
// relative).  Two guards, both matched to that precision: a Cholesky pivot below kQrPivot x G_jj is round-off, not data, and a
// first-round result whose Gram matrix is further than kQrRound2 from the identity is beyond what the second round repairs.  A block
// that trips either one is flagged and goes to the Gram-Schmidt kernel — that block only; what the first round may already have
// written is V T with T upper triangular with a positive diagonal, which has the same Q.
// Reference semantics: blockGramSchmidt, lib/transfer_util.cu:328-363.
constexpr int kQrChunk = 256;
constexpr double kQrPivot = 1e-5, kQrRound2 = 5e-2;
template <int NVEC> __global__ void __launch_bounds__(256) block_cholqr_kernel(float *V, int blockVol, int K, int ncf, int spin_bs, int *failed, int *failedBlock) {
  constexpr int nvec = NVEC;
  extern __shared__ double smem[];
  // layout: tile [kQrChunk][nvec + 1] float2 | G / L [nvec][nvec] double2 | Rinv [nvec][nvec] float2 | flag
  float2 *tile = reinterpret_cast<float2 *>(smem);
  constexpr int tstride = nvec + 1;
  double2 *G = reinterpret_cast<double2 *>(tile + (size_t)kQrChunk * tstride + (kQrChunk * tstride & 1));
  float2 *Rinv = reinterpret_cast<float2 *>(G + nvec * nvec);
  int *bad = reinterpret_cast<int *>(Rinv + nvec * nvec);
  const int A = blockIdx.x >> 1, chi = blockIdx.x & 1;
  const int Kc = K / 2, n = Kc * blockVol, nsc = spin_bs * ncf;
  constexpr int nvp = nvec / 2, npairs = nvec * (nvec + 1) / 2;
  auto addr4 = [&](int e, int vp) -> size_t {   // float4 index of (element e, vector pair vp)
    const int kk = e / blockVol, b = e - kk * blockVol;
    return (((size_t)A * K + chi * nsc + kk) * nvp + vp) * blockVol + b;
  };
  float4 *V4 = reinterpret_cast<float4 *>(V);
  if (threadIdx.x == 0) *bad = 0;
  for (int round = 0; round < 2; round++) {
    // ---- Gram matrix: pair p = (i <= j) per thread (up to 3 pairs for nvec = 32) ----
    int pi[3], pj[3];
    double2 acc[3];
#pragma unroll
    for (int q = 0; q < 3; q++) {
      const int p = threadIdx.x + 256 * q;
      acc[q] = make_double2(0.0, 0.0);
      pi[q] = pj[q] = -1;
      if (p < npairs) {   // row-major enumeration of the upper triangle
        int i = 0, rem = p;
        while (rem >= nvec - i) { rem -= nvec - i; i++; }
        pi[q] = i; pj[q] = i + rem;
      }
    }
    for (int c0 = 0; c0 < n; c0 += kQrChunk) {
      __syncthreads();
      const int e = c0 + threadIdx.x;
#pragma unroll
      for (int vp = 0; vp < nvp; vp++) {
        const float4 w = e < n ? V4[addr4(e, vp)] : make_float4(0.f, 0.f, 0.f, 0.f);
        tile[threadIdx.x * tstride + 2 * vp] = make_float2(w.x, w.y);
        tile[threadIdx.x * tstride + 2 * vp + 1] = make_float2(w.z, w.w);
      }
      __syncthreads();
#pragma unroll
      for (int q = 0; q < 3; q++) {
        if (pi[q] < 0) continue;
        // fp32 inside a chunk (packed: conj(a) b = two v_pk_fma_f32), fp64 across the chunks: the per-term conversions and fp64
        // multiply-adds made this loop the most expensive part of the kernel (59.5 ms per pass at 48^3 x 96); a chunk's 256-term
        // sum is good to ~1e-6 relative, the second CholeskyQR round and the fp64 chunk sums keep the result at fp32 round-off
        pkf2 part = {0.f, 0.f};
        for (int t = 0; t < kQrChunk; t++) {
          const float2 a = tile[t * tstride + pi[q]], b = tile[t * tstride + pj[q]];
          part = pk::cmac_conj(part, (pkf2){a.x, a.y}, (pkf2){b.x, b.y});
        }
        acc[q].x += (double)part.x; acc[q].y += (double)part.y;
      }
    }
#pragma unroll
    for (int q = 0; q < 3; q++)
      if (pi[q] >= 0) {
        G[pi[q] * nvec + pj[q]] = acc[q]; G[pj[q] * nvec + pi[q]] = make_double2(acc[q].x, -acc[q].y);
        // second round: the first one must have left something close to orthonormal
        if (round == 1 && (fabs(acc[q].x - (pi[q] == pj[q] ? 1.0 : 0.0)) > kQrRound2 || fabs(acc[q].y) > kQrRound2)) *bad = 1;
      }
    __syncthreads();
    if (*bad) break;
    // ---- Cholesky G = L L^dagger in place (lower triangle), column by column; thread i owns row i ----
    for (int j = 0; j < nvec; j++) {
      if ((int)threadIdx.x == j) {
        double d = G[j * nvec + j].x;
        for (int k = 0; k < j; k++) { const double2 l = G[j * nvec + k]; d -= l.x * l.x + l.y * l.y; }
        if (!(d > kQrPivot * G[j * nvec + j].x) || !(d > 0.0)) { *bad = 1; d = 1.0; }
        G[j * nvec + j] = make_double2(sqrt(d), 0.0);
      }
      __syncthreads();
      const int i = threadIdx.x;
      if (i > j && i < nvec) {
        double2 v = G[i * nvec + j];
        for (int k = 0; k < j; k++) {   // -= L[i][k] conj(L[j][k])
          const double2 a = G[i * nvec + k], b = G[j * nvec + k];
          v.x -= a.x * b.x + a.y * b.y; v.y -= a.y * b.x - a.x * b.y;
        }
        const double inv = 1.0 / G[j * nvec + j].x;
        G[i * nvec + j] = make_double2(v.x * inv, v.y * inv);
      }
      __syncthreads();
    }
    if (*bad) break;
    // ---- R = L^dagger (upper); column j of R^-1 by back substitution, thread j ----
    if ((int)threadIdx.x < nvec) {
      const int j = threadIdx.x;
      double2 x[NVEC];
#pragma unroll
      for (int i = nvec - 1; i >= 0; i--) {
        double2 v = make_double2(i == j ? 1.0 : 0.0, 0.0);
#pragma unroll
        for (int k = i + 1; k < nvec; k++) {   // R[i][k] = conj(L[k][i])
          if (k > j) continue;
          const double2 r = G[k * nvec + i];
          v.x -= r.x * x[k].x + r.y * x[k].y; v.y -= r.x * x[k].y - r.y * x[k].x;
        }
        const double inv = 1.0 / G[i * nvec + i].x;
        x[i] = i <= j ? make_double2(v.x * inv, v.y * inv) : make_double2(0.0, 0.0);
      }
#pragma unroll
      for (int i = 0; i < nvec; i++) Rinv[i * nvec + j] = make_float2((float)x[i].x, (float)x[i].y);
    }
    __syncthreads();
    // ---- V <- V R^-1: thread per element ----
    for (int c0 = 0; c0 < n; c0 += 256) {
      const int e = c0 + threadIdx.x;
      if (e >= n) continue;
      float2 v[NVEC];
#pragma unroll
      for (int vp = 0; vp < nvp; vp++) { const float4 w = V4[addr4(e, vp)]; v[2 * vp] = make_float2(w.x, w.y); v[2 * vp + 1] = make_float2(w.z, w.w); }
#pragma unroll
      for (int vp = nvp - 1; vp >= 0; vp--) {   // columns from the right: column j only needs v[0..j], so v can be overwritten in place
        float2 o[2];
#pragma unroll
        for (int h = 1; h >= 0; h--) {
          const int j = 2 * vp + h;
          pkf2 a = {0.f, 0.f};
#pragma unroll
          for (int i = 0; i <= j; i++) { const float2 r = Rinv[i * nvec + j]; a = pk::cmac(a, (pkf2){v[i].x, v[i].y}, (pkf2){r.x, r.y}); }
          o[h] = make_float2(a.x, a.y);
        }
        v[2 * vp] = o[0]; v[2 * vp + 1] = o[1];
        V4[addr4(e, vp)] = make_float4(o[0].x, o[0].y, o[1].x, o[1].y);
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0 && *bad) { atomicAdd(failed, 1); failedBlock[blockIdx.x] = 1; }
}

// ================================================================================================
// Direct Galerkin construction, step 2 ("VUV", reference ComputeVUV lib/coarse_op.cuh:487-600) on the matrix cores: for one forward
// direction mu and every aggregate A
//     Y[(chi, v)][(chi', v')] = sum_{x in A} sum_{s in chi, c} conj(V(x)[s, c; v]) [ (1 - gamma_mu) UV(x) ]^{chi'}[s, c; v']      (UV from galerkin_uv_kernel)
// — the sites whose mu-neighbour lies in the next aggregate give the coarse link Y_{2 mu}(A), all others the part S(A) of the local
// matrix (hermitian completion by the caller, coarse.hip).  The probing path gets the same numbers from 2 Nvec passes over V with a
// block reduction per coefficient; here the sum over the 256 sites of the aggregate IS the K dimension of a GEMM:
// per row block chi a real GEMM  M = Nvec rows (v; padded to 16 / 32), N = 4 Nvec real columns ((chi', v') x re / im), K = 256 x 6 x 2
// on v_mfma_f32_16x16x4_f32 (exact fp32).  The projector is applied on the fly: a column of chirality chi' = chi sees UV[s, c] itself, one of the
// other chirality phi(mu, s) UV[partner(mu, s), c] with phi in {+-1, +-i} (table below, read off spin_project / spin_reconstruct of dslash.hip).
// One work-group per aggregate, 8 waves: wave = 4 chi + w takes row block chi of the 64 sites with block coordinate y_mu = w, so the
// class-3 waves hold the link and classes 0..2 the local part without any masking.  A k-step is 4 sites of one (spin-colour, re/im):
//     A operand  lane (row v, kq): Re or Im of V(x_kq)[s, c; v]                      one 8-byte load serves the re and the im step
//     B operand  lane (col (j, o), kq): from w = phi UV(x_kq)[s', c; v'(j)] = (wr, wi):   re step: o ? wi : wr     im step: o ? -wr : wi
// so that column (j, re) collects Vr Wr + Vi Wi and (j, im) Vr Wi - Vi Wr, i.e. conj(V) W.  Operands come straight from global memory
// (every 16-byte word of the aggregate's V / UV rows is used by lanes of the work-group within microseconds: L1 / L2 absorb the pieces);
// partial tiles are summed through LDS and written in the link layout [site][matrix][column pair][row] float4.
// local = 1: UV is a chirality-diagonal site term (twisted clover, galerkin_local_uv_kernel): no cross columns, every class is local.
// Fine level only: 4 x 3 spin-colour, 4^4 aggregates, Nvec 8, 24 or 32, unpartitioned.
// ================================================================================================
typedef float gf32x4 __attribute__((ext_vector_type(4)));
// forward hop (1 - gamma_mu) in the chiral basis of dslash.hip: row s of the OTHER chirality's contribution is i^k UV[partner]
__device__ __constant__ int kGalPartner[4][4] = {{3, 2, 1, 0}, {3, 2, 1, 0}, {2, 3, 0, 1}, {2, 3, 0, 1}};
__device__ __constant__ int kGalPhase[4][4] = {{3, 3, 1, 1}, {0, 2, 2, 0}, {3, 1, 1, 3}, {2, 2, 2, 2}};
// the K loop of one wave (row block CHI, its class of 64 sites).  Branch-free on purpose: the first version selected same / cross rows, the
// phase and the padding rows with conditionals, which the compiler turned into ~300 branches with an s_waitcnt vmcnt(0) behind every load —
// 37 ms per direction at 48^3 x 96, no faster with two work-groups per CU.  Here every lane multiplies what it loads by a complex
// constant fixed before the loop: i^(3 o) for a column of the wave's own chirality (o: the lane's real / imaginary column), i^(phase + 3 o)
// for the other chirality (0 in local mode), 0 for a padding row of A — so that (bre, bim) = c w and the two MFMA steps follow.
template <int NVEC, int CHI, typename SiteB>
__device__ __forceinline__ void galerkin_vuv_accumulate(gf32x4 (&acc)[(NVEC + 15) / 16][NVEC / 4], const float2 *V, const float2 *UV, int A, int Aloc, int mu, int local, int row16, int kq, SiteB site_b, int cmBase) {
  constexpr int NVP = NVEC / 2, MT = (NVEC + 15) / 16, NT = NVEC / 4, BV = 256;
#pragma unroll
  for (int mt = 0; mt < MT; mt++)
#pragma unroll
    for (int nt = 0; nt < NT; nt++) acc[mt][nt] = (gf32x4){0.f, 0.f, 0.f, 0.f};
  const int o = row16 & 1;
  // i^k as (re, im)
  auto ipow = [](int k) -> float2 { k &= 3; return make_float2(k == 0 ? 1.f : (k == 2 ? -1.f : 0.f), k == 1 ? 1.f : (k == 3 ? -1.f : 0.f)); };
  const float2 cSame = ipow(3 * o);
  float2 cCross[2];
#pragma unroll
  for (int h = 0; h < 2; h++) {
    cCross[h] = ipow(kGalPhase[mu][2 * CHI + h] + 3 * o);
    if (local) cCross[h] = make_float2(0.f, 0.f);
  }
  // per-lane element offsets (float2 units) inside a (spin-colour) row of the aggregate: A rows v = 16 mt + row16 (padding rows: clamped, scaled by 0)
  int aOff[MT]; float aScale[MT];
#pragma unroll
  for (int mt = 0; mt < MT; mt++) {
    const int v = 16 * mt + row16, vv = v < NVEC ? v : NVEC - 1;
    aOff[mt] = (vv >> 1) * BV * 2 + (vv & 1);
    aScale[mt] = v < NVEC ? 1.f : 0.f;
  }
  int bOff[NT];
#pragma unroll
  for (int nt = 0; nt < NT; nt++) {
    const int j = 8 * nt + (row16 >> 1), vc = j - (nt >= NT / 2 ? NVEC : 0);
    bOff[nt] = (vc >> 1) * BV * 2 + (vc & 1);
  }
  const float2 *Va = V + (size_t)A * 12 * NVP * BV * 2, *Ua = UV + (size_t)Aloc * 12 * NVP * BV * 2;
  constexpr int ROW = NVP * BV * 2;   // float2 elements per (spin-colour) row
  // spin-colour row outermost, the wave's 64 sites (16 k-steps of 4) inside: with UV in class-major order (cmBase >= 0: the wave's sites are entries
  // cmBase .. cmBase + 63 of every UV row) consecutive k-steps read consecutive 64-byte pieces — every 128-byte line of UV is fetched once
#pragma unroll
  for (int s6 = 0; s6 < 6; s6++) {
    for (int g = 0; g < 16; g++) {
      const int b2 = 2 * site_b(4 * g + kq), u2 = cmBase >= 0 ? 2 * (cmBase + 4 * g + kq) : b2;
      {
      const int spin = 2 * CHI + s6 / 3, col = s6 % 3;
      const int rowSame = (3 * spin + col) * ROW, rowCross = (3 * kGalPartner[mu][spin] + col) * ROW;
      float are[MT], aim[MT], bre[NT], bim[NT];
#pragma unroll
      for (int mt = 0; mt < MT; mt++) {
        const float2 a = Va[rowSame + aOff[mt] + b2];
        are[mt] = aScale[mt] * a.x; aim[mt] = aScale[mt] * a.y;
      }
#pragma unroll
      for (int nt = 0; nt < NT; nt++) {
        constexpr int dummy = 0; (void)dummy;
        const bool same = (nt >= NT / 2) == (CHI == 1);
        const float2 w = Ua[(same ? rowSame : rowCross) + bOff[nt] + u2];
        const float2 c = same ? cSame : cCross[s6 / 3];
        bre[nt] = c.x * w.x - c.y * w.y;
        bim[nt] = c.x * w.y + c.y * w.x;
      }
      // all re steps, then all im steps: consecutive matrix instructions on different accumulators
#pragma unroll
      for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int nt = 0; nt < NT; nt++) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(are[mt], bre[nt], acc[mt][nt], 0, 0, 0);
#pragma unroll
      for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int nt = 0; nt < NT; nt++) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(aim[mt], bim[nt], acc[mt][nt], 0, 0, 0);
      }
    }
  }
}
template <int NVEC> __global__ void __launch_bounds__(512) galerkin_vuv_kernel(float *G, const float2 *V, const float2 *UV, int mu, int accumulateLocal, int pm, int local, int aggOffset, int classMajor) {
  constexpr int NVP = NVEC / 2, MT = (NVEC + 15) / 16, NT = NVEC / 4, n = 2 * NVEC, BV = 256;
  extern __shared__ float glds[];   // [class][mt][nt][reg][lane], one row block at a time
  const int A = blockIdx.x + aggOffset, lane = threadIdx.x & 63;   // UV holds the aggregates [aggOffset, aggOffset + gridDim) only
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int cls = wave & 3, chiR = wave >> 2;
  const int row16 = lane & 15, kq = lane >> 4;
  // site list of this wave: y_mu = cls, the other three coordinates from pos = 0 .. 63; b = place of the site inside the aggregate
  auto site_b = [&](int pos) -> int {
    int y[4], q = pos;
    for (int d = 0; d < 4; d++) if (d != mu) { y[d] = q & 3; q >>= 2; }
    y[mu] = cls;
    const int lex = ((y[3] * 4 + y[2]) * 4 + y[1]) * 4 + y[0];
    return pm ? ((y[0] + y[1] + y[2] + y[3]) & 1) * (BV / 2) + (lex >> 1) : lex;
  };
  gf32x4 acc[MT][NT];
  const int cmBase = classMajor ? cls * 64 : -1;
  if (chiR == 0) galerkin_vuv_accumulate<NVEC, 0>(acc, V, UV, A, (int)blockIdx.x, mu, local, row16, kq, site_b, cmBase);
  else galerkin_vuv_accumulate<NVEC, 1>(acc, V, UV, A, (int)blockIdx.x, mu, local, row16, kq, site_b, cmBase);
  // ---- partial tiles -> LDS, one row block (chi) at a time so that the buffer is 4 waves x 12 KB and two work-groups share a CU;
  // classes 0..2 summed = local part, class 3 = link ----
  constexpr int TILE = MT * NT * 4 * 64;   // floats per wave
  float4 *G4 = reinterpret_cast<float4 *>(G);
  for (int chi = 0; chi < 2; chi++) {
    if (chi) __syncthreads();
    if (chiR == chi) {
#pragma unroll
      for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int nt = 0; nt < NT; nt++)
#pragma unroll
          for (int r = 0; r < 4; r++) glds[cls * TILE + ((mt * NT + nt) * 4 + r) * 64 + lane] = acc[mt][nt][r];
    }
    __syncthreads();
    // output element (row i = (chi, v), complex column j): tile (v / 16, j / 8); D layout: row = 4 (lane / 16) + reg, col = lane & 15
    // local (site-diagonal term of the fine operator): every class belongs to the local matrix, no link is written
    constexpr int HALF = NVEC * (n / 2);   // (row v, column pair) elements of one row block
    for (int e = threadIdx.x + (local ? HALF : 0); e < 2 * HALF; e += blockDim.x) {
      const int which = e / HALF, r2 = e - which * HALF;   // 0: link (class 3), 1: local (classes 0..2)
      const int jp = r2 / NVEC, v = r2 - jp * NVEC, i = chi * NVEC + v, mt = v >> 4, rr = v & 15;
      float val[4];
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const int j = 2 * jp + h, nt = j >> 3;
#pragma unroll
        for (int oo = 0; oo < 2; oo++) {
          const int col = 2 * (j & 7) + oo;
          const int off = ((mt * NT + nt) * 4 + (rr & 3)) * 64 + (rr >> 2) * 16 + col;
          val[2 * h + oo] = which ? glds[off] + glds[TILE + off] + glds[2 * TILE + off] + (local ? glds[3 * TILE + off] : 0.f) : glds[3 * TILE + off];
        }
      }
      const size_t dst = (((size_t)A * 9 + (which ? 8 : 2 * mu)) * (n / 2) + jp) * n + i;
      if (which && accumulateLocal) { const float4 old = G4[dst]; val[0] += old.x; val[1] += old.y; val[2] += old.z; val[3] += old.w; }
      G4[dst] = make_float4(val[0], val[1], val[2], val[3]);
    }
  }
}

bool Transfer::canDirectGalerkin() const {
  static int off = -1;
  if (off < 0) { const char *e = getenv("QUDA_AMD_GALERKIN_DIRECT"); off = (e && !atoi(e)) ? 1 : 0; }
  if (off || fineSpin != 4 || fineColor != 3 || spin_bs != 2 || (Nvec != 8 && Nvec != 24 && Nvec != 32)) return false;
  for (int d = 0; d < 4; d++) if (geo_bs[d] != 4 || Xc[d] == 1 || commGrid().partitioned(d)) return false;
  return true;
}
// forward link Y_{2 mu} and the in-aggregate part S of all coarse sites from UV = galerkinUV(V): slots 2 mu and 8 of the coarse links
void Transfer::directGalerkinVUV(float *links, const float *UV, int mu, bool accumulateLocal, bool local, int aggOffset, int nAggChunk, bool classMajor) const {
  if (!canDirectGalerkin()) errorQuda("direct Galerkin construction not available for this transfer operator");
  const size_t lds = (size_t)4 * ((Nvec + 15) / 16) * (Nvec / 4) * 4 * 64 * sizeof(float);
#define QA_VUV(NV) { static bool attr = false; \
    if (!attr) { HIP_CHECK(hipFuncSetAttribute((const void *)galerkin_vuv_kernel<NV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); attr = true; } \
    hipLaunchKernelGGL((galerkin_vuv_kernel<NV>), dim3(nAggChunk > 0 ? nAggChunk : (int)nAgg), dim3(512), lds, computeStream(), links, (const float2 *)V, (const float2 *)UV, mu, accumulateLocal ? 1 : 0, parityMajor ? 1 : 0, local ? 1 : 0, aggOffset, classMajor ? 1 : 0); }
  if (Nvec == 24) QA_VUV(24) else if (Nvec == 32) QA_VUV(32) else QA_VUV(8)
#undef QA_VUV
  HIP_CHECK(hipGetLastError());
}

// ---- random source ----
__device__ __forceinline__ float hash_uniform(unsigned long long seed, unsigned long long i) {
  unsigned long long z = seed + 0x9E3779B97F4A7C15ULL * (i + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  z ^= z >> 31;
  return (float)((z >> 40) * (1.0 / 16777216.0));
}
// vec reals per 16-byte (8-byte) plane element: the padding sites [Vh, stride) of every plane stay zero — the flat BLAS kernels run over them
template <typename real> __global__ void random_kernel(real *v, long n, unsigned long long seed, unsigned long long offset, int vec, int stride, int Vh) {
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i < n) v[i] = (int)((i / vec) % stride) < Vh ? (real)hash_uniform(seed, offset + i) : (real)0;
}
void spinorRandom(ColorSpinorField &f, unsigned long long seed) {
  if (f.Location() != QUDA_CUDA_FIELD_LOCATION) errorQuda("device field required");
  const int nseg = f.SiteSubset() == QUDA_FULL_SITE_SUBSET ? 2 : 1;
  const long n = (long)f.Stride() * f.Nspin() * f.Ncolor() * 2;
  const unsigned long long rank_off = (unsigned long long)commGrid().rank << 40;
  for (int s = 0; s < nseg; s++) {
    void *p = nseg == 2 ? (s ? f.Odd().V() : f.Even().V()) : f.V();
    const int bs = 256; const long nb = (n + bs - 1) / bs;
    const int vec = f.Nspin() == 4 && f.Precision() == QUDA_SINGLE_PRECISION ? 4 : 2;
    if (f.Precision() == QUDA_DOUBLE_PRECISION) hipLaunchKernelGGL((random_kernel<double>), dim3(nb), dim3(bs), 0, computeStream(), (double *)p, n, seed, rank_off + (unsigned long long)s * n, vec, f.Stride(), f.VolumeCB());
    else if (f.Precision() == QUDA_SINGLE_PRECISION) hipLaunchKernelGGL((random_kernel<float>), dim3(nb), dim3(bs), 0, computeStream(), (float *)p, n, seed, rank_off + (unsigned long long)s * n, vec, f.Stride(), f.VolumeCB());
    else errorQuda("random source needs fp64/fp32 storage");
  }
  HIP_CHECK(hipGetLastError());
}

// ================================================================================================
Transfer::Transfer(const std::vector<ColorSpinorField *> &B, int Nvec_, int *gbs, int spin_bs_)
    : Nvec(Nvec_), spin_bs(spin_bs_), V(nullptr), V_h(nullptr), block_to_fine(nullptr), fine_to_block(nullptr), flops_(0),
      site_subset(QUDA_FULL_SITE_SUBSET), subset_parity(QUDA_INVALID_PARITY) {
  if ((int)B.size() < Nvec) errorQuda("need %d null vectors, got %zu", Nvec, B.size());
  if (Nvec % 2 || Nvec > kMaxVec) errorQuda("Nvec = %d must be even and <= %d", Nvec, kMaxVec);
  const ColorSpinorField &b0 = *B[0];
  if (b0.SiteSubset() != QUDA_FULL_SITE_SUBSET) errorQuda("null vectors must be full fields");
  fineSpin = b0.Nspin(); fineColor = b0.Ncolor();
  if (fineSpin % spin_bs || fineSpin / spin_bs != 2) errorQuda("spin block %d does not leave two chiralities from %d spins", spin_bs, fineSpin);
  for (int d = 0; d < 4; d++) Xf[d] = b0.X(d);
  // block-size fallback, reference lib/transfer.cpp:31-44
  for (int d = 0; d < 4; d++) {
    while (gbs[d] > 0) {
      if (d == 0 && Xf[0] == gbs[0]) warningQuda("X-dimension length %d cannot block length %d", Xf[0], gbs[0]);
      else if ((Xf[d] / gbs[d] + 1) % 2 == 0) warningQuda("Indexing does not (yet) support odd coarse dimensions: X(%d) = %d", d, Xf[d] / gbs[d]);
      else if ((Xf[d] / gbs[d]) * gbs[d] != Xf[d]) warningQuda("cannot block dim[%d]=%d with block size = %d", d, Xf[d], gbs[d]);
      else break;
      gbs[d] /= 2;
    }
    if (gbs[d] == 0) errorQuda("Unable to block dimension %d", d);
  }
  blockVol = 1; nAgg = 1; fineVol = 1;
  for (int d = 0; d < 4; d++) { geo_bs[d] = gbs[d]; Xc[d] = Xf[d] / gbs[d]; blockVol *= gbs[d]; nAgg *= Xc[d]; fineVol *= Xf[d]; }
  if (blockVol == 1) errorQuda("Total geometric block size is 1");
  if (blockVol > 1024) errorQuda("aggregate of %d sites exceeds one work-group", blockVol);
  if (getVerbosity() >= QUDA_VERBOSE) printfQuda("Transfer: using block size %d x %d x %d x %d\n", geo_bs[0], geo_bs[1], geo_bs[2], geo_bs[3]);
  // order of the sites inside an aggregate: parity-major on the finest level when every block extent is even (block_coords above)
  {
    static int pmEnv = -1;
    if (pmEnv < 0) { const char *e = getenv("QUDA_AMD_V_PARITY_MAJOR"); pmEnv = e ? atoi(e) : 1; }
    parityMajor = pmEnv && fineSpin == 4 && geo_bs[0] % 2 == 0 && geo_bs[1] % 2 == 0 && geo_bs[2] % 2 == 0 && geo_bs[3] % 2 == 0;
  }
  createGeoMap();
  // through the device pool: the next hierarchy (the other twist flavour, a refinement pass) takes the same 24.5 GB at 48^3 x 96 instead of
  // a fresh hipMalloc + hipFree pair (~0.1 s, and the first touch of fresh pages slows every kernel that meets them)
  V = (float *)poolDeviceMalloc(vBytes());
  fillAndOrthonormalise(B);
}

__global__ void v_to_half_kernel(vhalf4_t *out, const float4 *in, size_t n) {
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 v = in[i];
  vhalf4_t h;
  h.x = (_Float16)v.x; h.y = (_Float16)v.y; h.z = (_Float16)v.z; h.w = (_Float16)v.w;
  out[i] = h;
}
void Transfer::makeHalf() const {
  if (V_h) return;
  const size_t n4 = vBytes() / sizeof(float4);
  V_h = poolDeviceMalloc(n4 * sizeof(vhalf4_t));
  hipLaunchKernelGGL(v_to_half_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, computeStream(), (vhalf4_t *)V_h, (const float4 *)V, n4);
  HIP_CHECK(hipGetLastError());
}

Transfer::~Transfer() {
  if (V_h) poolDeviceFree(V_h, 0);
  if (V) poolDeviceFree(V, 0);
  if (block_to_fine) poolDeviceFree(block_to_fine, 0);
  if (fine_to_block) poolDeviceFree(fine_to_block, 0);
}

// reference createGeoMap lib/transfer.cpp:220-258 (fine site -> coarse site), plus the position inside the aggregate
struct GeoMapArg { int Xf[4], Xc[4], bs[4], blockVol, pm; long Vh, Vhc; };
__global__ void geo_map_kernel(int *b2f, int *f2b, GeoMapArg a) {
  const long f = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (f >= 2 * a.Vh) return;
  const int parity = f >= a.Vh;
  const long i = f - parity * a.Vh;
  const long za = i / (a.Xf[0] / 2); const int xh = (int)(i - za * (a.Xf[0] / 2));
  const long zb = za / a.Xf[1]; const int y = (int)(za - zb * a.Xf[1]);
  const int t = (int)(zb / a.Xf[2]), z = (int)(zb - (long)t * a.Xf[2]);
  const int x[4] = {2 * xh + ((y + z + t + parity) & 1), y, z, t};
  int xc[4], yb[4];
  for (int d = 0; d < 4; d++) { xc[d] = x[d] / a.bs[d]; yb[d] = x[d] % a.bs[d]; }
  const int cpar = (xc[0] + xc[1] + xc[2] + xc[3]) & 1;
  const long clex = ((long)(xc[3] * a.Xc[2] + xc[2]) * a.Xc[1] + xc[1]) * a.Xc[0] + xc[0];
  const long A = cpar * a.Vhc + clex / 2;
  int b = ((yb[3] * a.bs[2] + yb[2]) * a.bs[1] + yb[1]) * a.bs[0] + yb[0];
  if (a.pm) b = ((yb[0] + yb[1] + yb[2] + yb[3]) & 1) * (a.blockVol / 2) + b / 2;   // block extents even: site parity = block-local parity
  b2f[A * a.blockVol + b] = (int)f;
  f2b[f] = (int)(A * a.blockVol + b);
}
void Transfer::createGeoMap() {
  // on the device (round 3): the host loop over 10.6 M sites and its two 42 MB copies cost ~0.1 s of every build at 48^3 x 96
  GeoMapArg a;
  for (int d = 0; d < 4; d++) { a.Xf[d] = Xf[d]; a.Xc[d] = Xc[d]; a.bs[d] = geo_bs[d]; }
  a.blockVol = blockVol; a.pm = parityMajor ? 1 : 0; a.Vh = fineVol / 2; a.Vhc = nAgg / 2;
  block_to_fine = (int *)poolDeviceMalloc(fineVol * sizeof(int));
  fine_to_block = (int *)poolDeviceMalloc(fineVol * sizeof(int));
  hipLaunchKernelGGL(geo_map_kernel, dim3((unsigned)((fineVol + 255) / 256)), dim3(256), 0, computeStream(), block_to_fine, fine_to_block, a);
  HIP_CHECK(hipGetLastError());
}

void Transfer::fillAndOrthonormalise(const std::vector<ColorSpinorField *> &B) {
  VecList vl;
  memset(&vl, 0, sizeof(vl));
  for (int v = 0; v < Nvec; v++) { const FineVec f = fineVec(*B[v]); vl.ev[v] = f.v[0]; vl.od[v] = f.v[1]; }
  const FineVec f0 = fineVec(*B[0]);
  const long total = fineVol;
  const int K = fineSpin * fineColor;
  if (fineSpin == 4) hipLaunchKernelGGL((fillv_kernel<4>), dim3((total + 255) / 256), dim3(256), 0, computeStream(), V, vl, f0.stride, f0.Vh, block_to_fine, blockVol, K, Nvec, total);
  else hipLaunchKernelGGL((fillv_kernel<2>), dim3((total + 255) / 256), dim3(256), 0, computeStream(), V, vl, f0.stride, f0.Vh, block_to_fine, blockVol, K, Nvec, total);
  HIP_CHECK(hipGetLastError());
  static int useQr = -1;
  if (useQr < 0) { const char *e = getenv("QUDA_AMD_BLOCK_ORTHO"); useQr = (e && !strcmp(e, "gs")) ? 0 : 1; }
  int nfail = useQr ? 0 : 1;
  lastGsFallbackBlocks = 0;
  if (useQr) {
    int *d_fail = nullptr;   // [0] number of flagged blocks, [1 + b] flag of (aggregate, chirality) block b
    const size_t failBytes = (1 + 2 * (size_t)nAgg) * sizeof(int);
    d_fail = (int *)poolDeviceMalloc(failBytes);
    HIP_CHECK(hipMemsetAsync(d_fail, 0, failBytes, computeStream()));
    const size_t tileFloats2 = (size_t)kQrChunk * (Nvec + 1) + 1;
    const size_t lds = tileFloats2 * sizeof(float2) + (size_t)Nvec * Nvec * (sizeof(double2) + sizeof(float2)) + 64;
#define QA_QR(NV)                                                                                                                  \
  {                                                                                                                                \
    HIP_CHECK(hipFuncSetAttribute((const void *)block_cholqr_kernel<NV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));   \
    hipLaunchKernelGGL((block_cholqr_kernel<NV>), dim3(2 * nAgg), dim3(256), lds, computeStream(), V, blockVol, K, fineColor, spin_bs, d_fail, d_fail + 1); \
  }
    switch (Nvec) {
      case 4: QA_QR(4) break;
      case 8: QA_QR(8) break;
      case 24: QA_QR(24) break;
      case 32: QA_QR(32) break;
      default: nfail = 1;   // not instantiated: Gram-Schmidt
    }
#undef QA_QR
    HIP_CHECK(hipGetLastError());
    if (!nfail) HIP_CHECK(hipMemcpyAsync(&nfail, d_fail, sizeof(int), hipMemcpyDeviceToHost, computeStream()));
    HIP_CHECK(hipStreamSynchronize(computeStream()));
    if (nfail && (Nvec == 4 || Nvec == 8 || Nvec == 24 || Nvec == 32)) {
      // flagged blocks hold their original vectors or those times an upper-triangular matrix with a positive diagonal (first round done,
      // second refused): the same Q either way, so the sequential kernel runs on exactly those blocks, in place
      if (getVerbosity() >= QUDA_SUMMARIZE) printfQuda("block orthonormalisation: %d of %ld blocks too ill-conditioned for CholeskyQR2 in fp32, Gram-Schmidt on those\n", nfail, 2 * (long)nAgg);
      hipLaunchKernelGGL(block_gs_kernel, dim3(2 * nAgg), dim3(256), 0, computeStream(), V, blockVol, K, fineColor, spin_bs, Nvec, (const int *)(d_fail + 1));
      HIP_CHECK(hipGetLastError());
      HIP_CHECK(hipStreamSynchronize(computeStream()));
      lastGsFallbackBlocks = nfail;
      nfail = 0;
    }
    poolDeviceFree(d_fail, failBytes);
  }
  if (nfail) hipLaunchKernelGGL(block_gs_kernel, dim3(2 * nAgg), dim3(256), 0, computeStream(), V, blockVol, K, fineColor, spin_bs, Nvec, (const int *)nullptr);
  HIP_CHECK(hipGetLastError());
  HIP_CHECK(hipStreamSynchronize(computeStream()));
}

ColorSpinorField *Transfer::createCoarseField() const {
  ColorSpinorParam p;
  p.location = QUDA_CUDA_FIELD_LOCATION;
  p.nSpin = 2; p.nColor = Nvec;
  for (int d = 0; d < 4; d++) p.x[d] = Xc[d];
  p.siteSubset = QUDA_FULL_SITE_SUBSET;
  p.precision = QUDA_SINGLE_PRECISION;
  p.create = QUDA_ZERO_FIELD_CREATE;
  return new ColorSpinorField(p);
}
ColorSpinorField *Transfer::createFineField() const {
  ColorSpinorParam p;
  p.location = QUDA_CUDA_FIELD_LOCATION;
  p.nSpin = fineSpin; p.nColor = fineColor;
  for (int d = 0; d < 4; d++) p.x[d] = Xf[d];
  p.siteSubset = QUDA_FULL_SITE_SUBSET;
  p.precision = QUDA_SINGLE_PRECISION;
  p.create = QUDA_ZERO_FIELD_CREATE;
  return new ColorSpinorField(p);
}

static CoarseVec coarseVec(const ColorSpinorField &c) {
  FineVec f = fineVec(c);
  CoarseVec r;
  r.v[0] = f.v[0]; r.v[1] = f.v[1]; r.stride = f.stride; r.Vh = f.Vh;
  return r;
}

#define QA_TRANSFER_DISPATCH(CALL)                                                             \
  if (fineSpin == 4 && fineColor == 3) {                                                       \
    switch (Nvec) {                                                                            \
      case 4: { CALL(4, 3, 4, 4); } break;                                                     \
      case 8: { CALL(4, 3, 8, 4); } break;                                                     \
      case 24: { CALL(4, 3, 24, 4); } break;                                                   \
      case 32: { CALL(4, 3, 32, 4); } break;                                                   \
      default: errorQuda("Nvec = %d not instantiated for the fine level (4, 8, 24, 32)", Nvec); \
    }                                                                                          \
  } else if (fineSpin == 2 && fineColor == 4 && Nvec == 4) { CALL(2, 4, 4, 2);                 \
  } else if (fineSpin == 2 && fineColor == 8 && Nvec == 8) { CALL(2, 8, 8, 2);                 \
  } else if (fineSpin == 2 && fineColor == 8 && Nvec == 4) { CALL(2, 8, 4, 2);                 \
  } else if (fineSpin == 2 && fineColor == 24 && Nvec == 24) { CALL(2, 24, 24, 2);             \
  } else if (fineSpin == 2 && fineColor == 24 && Nvec == 32) { CALL(2, 24, 32, 2);             \
  } else if (fineSpin == 2 && fineColor == 32 && Nvec == 32) { CALL(2, 32, 32, 2);             \
  } else errorQuda("transfer %d x %d -> Nvec %d not instantiated", fineSpin, fineColor, Nvec);

void Transfer::setSiteSubset(QudaSiteSubset subset, QudaParity parity) {
  if (subset == QUDA_PARITY_SITE_SUBSET && parity != QUDA_EVEN_PARITY && parity != QUDA_ODD_PARITY) errorQuda("Undefined parity %d", parity);
  site_subset = subset;
  subset_parity = parity;
}

// the x-neighbour aggregate order (aggregate_of_block) where the blocking allows it; QUDA_AMD_PROLONG_XGROUP=0 switches it off
static AggMap aggMapOf(const Transfer &T) {
  AggMap amap = {0, T.Xc[0], T.Xc[1], T.Xc[2], T.nAgg / 2};
  static int off = -1;
  if (off < 0) { const char *e = getenv("QUDA_AMD_PROLONG_XGROUP"); off = (e && !atoi(e)) ? 1 : 0; }
  if (!off && T.geo_bs[0] == 4 && T.Xc[0] % 4 == 0 && T.nAgg % 32 == 0) amap.xgroup = T.Xc[0] / 4;
  return amap;
}

void Transfer::R(ColorSpinorField &coarse, const ColorSpinorField &fine, int dir, int boundary) const {
  const bool sub = fine.SiteSubset() == QUDA_PARITY_SITE_SUBSET;
  if (sub && site_subset != QUDA_PARITY_SITE_SUBSET) errorQuda("single-parity fine field but the transfer is set to full fields");
  if (fine.Nspin() != fineSpin || fine.Ncolor() != fineColor || (long)fine.VolumeCB() * 2 != fineVol) errorQuda("fine field does not match the transfer operator");
  if (coarse.Nspin() != 2 || coarse.Ncolor() != Nvec || coarse.Volume() != nAgg) errorQuda("coarse field does not match the transfer operator");
  const FineVec in = fineVec(fine, sub ? (int)subset_parity : -1);
  const CoarseVec out = coarseVec(coarse);
  MaskArg m;
  m.dir = dir; m.boundary = boundary; m.pm = parityMajor ? 1 : 0;
  for (int d = 0; d < 4; d++) { m.bs[d] = geo_bs[d]; m.single[d] = Xc[d] == 1; }
  const int threads = (blockVol + 63) / 64 * 64;
  int gs = 1; while (gs < blockVol) gs <<= 1;
  const bool small = blockVol <= 32;
  const bool half = coarseHalfStorage() && V_h != nullptr && dir < 0;   // the Galerkin-split variants always use the fp32 master
  // the barrier-free restrictor (restrict_stream_kernel): fine level with two chiralities of K / 2 rows, at most four waves per aggregate
  static int streamOff = -1;
  if (streamOff < 0) { const char *e = getenv("QUDA_AMD_RESTRICT_STREAM"); streamOff = (e && !atoi(e)) ? 1 : 0; }
  const bool stream = !streamOff && !small && threads <= 256 && spin_bs == 2;
#define QA_R(NSF, NCF, NVEC, NV) \
  if (small && half) hipLaunchKernelGGL((restrict_small_kernel<NSF, NCF, NVEC, NV, false, true>), dim3(nAgg), dim3(512), 0, computeStream(), out, out, in, (const void *)V_h, block_to_fine, blockVol, gs, spin_bs, m); \
  else if (small) hipLaunchKernelGGL((restrict_small_kernel<NSF, NCF, NVEC, NV, false, false>), dim3(nAgg), dim3(512), 0, computeStream(), out, out, in, (const void *)V, block_to_fine, blockVol, gs, spin_bs, m); \
  else if constexpr (NSF == 4 && NVEC % 8 == 0) { \
    if (stream && half) hipLaunchKernelGGL((restrict_stream_kernel<NSF, NCF, NVEC, NV, true>), dim3(nAgg), dim3(threads), 0, computeStream(), out, in, (const void *)V_h, block_to_fine, blockVol, m, aggMapOf(*this)); \
    else if (stream) hipLaunchKernelGGL((restrict_stream_kernel<NSF, NCF, NVEC, NV, false>), dim3(nAgg), dim3(threads), 0, computeStream(), out, in, (const void *)V, block_to_fine, blockVol, m, aggMapOf(*this)); \
    else if (half) hipLaunchKernelGGL((restrict_kernel<NSF, NCF, NVEC, NV, false, true>), dim3(nAgg), dim3(threads), 0, computeStream(), out, out, in, (const void *)V_h, block_to_fine, blockVol, spin_bs, m, aggMapOf(*this)); \
    else hipLaunchKernelGGL((restrict_kernel<NSF, NCF, NVEC, NV, false, false>), dim3(nAgg), dim3(threads), 0, computeStream(), out, out, in, (const void *)V, block_to_fine, blockVol, spin_bs, m, aggMapOf(*this)); \
  } \
  else if (half) hipLaunchKernelGGL((restrict_kernel<NSF, NCF, NVEC, NV, false, true>), dim3(nAgg), dim3(threads), 0, computeStream(), out, out, in, (const void *)V_h, block_to_fine, blockVol, spin_bs, m, aggMapOf(*this)); \
  else hipLaunchKernelGGL((restrict_kernel<NSF, NCF, NVEC, NV, false, false>), dim3(nAgg), dim3(threads), 0, computeStream(), out, out, in, (const void *)V, block_to_fine, blockVol, spin_bs, m, aggMapOf(*this))
  if (g_acctOn) {   // V once (fp32 or its fp16 mirror; a single-parity source touches half of it) + fine vector + coarse vector
    const double frac = sub ? 0.5 : 1.0;
    char tag[48]; snprintf(tag, sizeof(tag), "level %s -> coarse%s", fineSpin == 4 ? "0" : "c", half ? " fp16 V" : "");
    const bool streamed = stream && fineSpin == 4 && Nvec % 8 == 0;
    acct(small ? "restrict_small_kernel" : (streamed ? "restrict_stream_kernel" : "restrict_kernel"), frac * fineVol * ((double)fineSpin * fineColor * Nvec * (half ? 4 : 8) + fineSpin * fineColor * 8.0) + (double)nAgg * 2 * Nvec * 8, tag);
  }
  QA_TRANSFER_DISPATCH(QA_R)
#undef QA_R
  HIP_CHECK(hipGetLastError());
  flops_ += 8ull * fineSpin * fineColor * Nvec * fineVol;  // reference lib/restrictor.cu:405
}

void Transfer::RSplit(ColorSpinorField &leaving, ColorSpinorField &staying, const ColorSpinorField &fine, int dir) const {
  if (fine.SiteSubset() != QUDA_FULL_SITE_SUBSET) errorQuda("the split restriction works on full fields");
  if (fine.Nspin() != fineSpin || fine.Ncolor() != fineColor || fine.Volume() != fineVol) errorQuda("fine field does not match the transfer operator");
  if (leaving.Nspin() != 2 || leaving.Ncolor() != Nvec || leaving.Volume() != nAgg || staying.Volume() != nAgg || staying.Ncolor() != Nvec) errorQuda("coarse field does not match the transfer operator");
  if (dir < 0 || dir > 7) errorQuda("direction %d", dir);
  const FineVec in = fineVec(fine);
  const CoarseVec out = coarseVec(leaving), out2 = coarseVec(staying);
  MaskArg m;
  m.dir = dir; m.boundary = 1; m.pm = parityMajor ? 1 : 0;
  for (int d = 0; d < 4; d++) { m.bs[d] = geo_bs[d]; m.single[d] = Xc[d] == 1; }
  const int threads = (blockVol + 63) / 64 * 64;
  int gs = 1; while (gs < blockVol) gs <<= 1;
  const bool small = blockVol <= 32;
#define QA_R2(NSF, NCF, NVEC, NV) \
  if (small) hipLaunchKernelGGL((restrict_small_kernel<NSF, NCF, NVEC, NV, true, false>), dim3(nAgg), dim3(512), 0, computeStream(), out, out2, in, (const void *)V, block_to_fine, blockVol, gs, spin_bs, m); \
  else hipLaunchKernelGGL((restrict_kernel<NSF, NCF, NVEC, NV, true, false>), dim3(nAgg), dim3(threads), 0, computeStream(), out, out2, in, (const void *)V, block_to_fine, blockVol, spin_bs, m, aggMapOf(*this))
  QA_TRANSFER_DISPATCH(QA_R2)
#undef QA_R2
  HIP_CHECK(hipGetLastError());
  flops_ += 8ull * fineSpin * fineColor * Nvec * fineVol;
}

void Transfer::P(ColorSpinorField &fine, const ColorSpinorField &coarse) const {
  const bool sub = fine.SiteSubset() == QUDA_PARITY_SITE_SUBSET;
  if (sub && site_subset != QUDA_PARITY_SITE_SUBSET) errorQuda("single-parity fine field but the transfer is set to full fields");
  if (fine.Nspin() != fineSpin || fine.Ncolor() != fineColor || (long)fine.VolumeCB() * 2 != fineVol) errorQuda("fine field does not match the transfer operator");
  if (coarse.Nspin() != 2 || coarse.Ncolor() != Nvec || coarse.Volume() != nAgg) errorQuda("coarse field does not match the transfer operator");
  const FineVec out = fineVec(fine, sub ? (int)subset_parity : -1);
  const CoarseVec in = coarseVec(coarse);
  const int threads = (blockVol + 63) / 64 * 64;
  if (threads > 512) errorQuda("aggregates of %d sites: the prolongator runs one work-group of at most 512 threads per aggregate", blockVol);
  int gs = 1; while (gs < blockVol) gs <<= 1;
  const bool small = blockVol <= 32;
  const bool half = coarseHalfStorage() && V_h != nullptr;
#define QA_P(NSF, NCF, NVEC, NV) \
  if (small && half) hipLaunchKernelGGL((prolong_small_kernel<NSF, NCF, NVEC, NV, true>), dim3(nAgg), dim3(512), 0, computeStream(), out, in, (const void *)V_h, block_to_fine, blockVol, gs, spin_bs); \
  else if (small) hipLaunchKernelGGL((prolong_small_kernel<NSF, NCF, NVEC, NV, false>), dim3(nAgg), dim3(512), 0, computeStream(), out, in, (const void *)V, block_to_fine, blockVol, gs, spin_bs); \
  else if (half) hipLaunchKernelGGL((prolong_kernel<NSF, NCF, NVEC, NV, true>), dim3(nAgg), dim3(threads), 0, computeStream(), out, in, (const void *)V_h, block_to_fine, blockVol, spin_bs, amap); \
  else hipLaunchKernelGGL((prolong_kernel<NSF, NCF, NVEC, NV, false>), dim3(nAgg), dim3(threads), 0, computeStream(), out, in, (const void *)V, block_to_fine, blockVol, spin_bs, amap)
  const AggMap amap = aggMapOf(*this);
  if (g_acctOn) {
    const double frac = sub ? 0.5 : 1.0;
    char tag[48]; snprintf(tag, sizeof(tag), "coarse -> level %s%s", fineSpin == 4 ? "0" : "c", half ? " fp16 V" : "");
    acct(small ? "prolong_small_kernel" : "prolong_kernel", frac * fineVol * ((double)fineSpin * fineColor * Nvec * (half ? 4 : 8) + fineSpin * fineColor * 8.0) + (double)nAgg * 2 * Nvec * 8, tag);
  }
  QA_TRANSFER_DISPATCH(QA_P)
#undef QA_P
  HIP_CHECK(hipGetLastError());
  flops_ += 8ull * fineSpin * fineColor * Nvec * fineVol;  // reference lib/prolongator.cu:228
}

// four sources per pass over V (fine level, fp32 V): see restrict_stream4_kernel / prolong4_kernel
bool Transfer::canQuad() const { return fineSpin == 4 && fineColor == 3 && spin_bs == 2 && blockVol > 32 && blockVol <= 256 && (Nvec == 8 || Nvec == 24 || Nvec == 32); }
void Transfer::R4(ColorSpinorField *const coarse[4], const ColorSpinorField *const fine[4]) const {
  if (!canQuad()) errorQuda("four-source restrictor: fine level with 4^4-type aggregates only");
  const bool sub = fine[0]->SiteSubset() == QUDA_PARITY_SITE_SUBSET;
  if (sub && site_subset != QUDA_PARITY_SITE_SUBSET) errorQuda("single-parity fine field but the transfer is set to full fields");
  Src4 a;
  for (int s = 0; s < 4; s++) {
    if (fine[s]->Nspin() != fineSpin || fine[s]->Ncolor() != fineColor || (long)fine[s]->VolumeCB() * 2 != fineVol || fine[s]->SiteSubset() != fine[0]->SiteSubset() || fine[s]->Precision() != QUDA_SINGLE_PRECISION)
      errorQuda("fine field %d does not match the transfer operator", s);
    a.f[s] = fineVec(*fine[s], sub ? (int)subset_parity : -1);
    a.c[s] = coarseVec(*coarse[s]);
  }
  MaskArg m;
  m.dir = -1; m.boundary = 0; m.pm = parityMajor ? 1 : 0;
  for (int d = 0; d < 4; d++) { m.bs[d] = geo_bs[d]; m.single[d] = Xc[d] == 1; }
  const int threads = (blockVol + 63) / 64 * 64;
  if (g_acctOn) {
    const double frac = sub ? 0.5 : 1.0;
    acct("restrict_stream4_kernel", frac * fineVol * ((double)fineSpin * fineColor * Nvec * 8 + 4.0 * fineSpin * fineColor * 8.0) + 4.0 * nAgg * 2 * Nvec * 8, "level 0 -> coarse, 4 sources");
  }
  switch (Nvec) {
    case 8: hipLaunchKernelGGL((restrict_stream4_kernel<3, 8, 4>), dim3(nAgg), dim3(threads), 0, computeStream(), a, (const void *)V, block_to_fine, blockVol, m, aggMapOf(*this), Blk4{}); break;
    case 24: hipLaunchKernelGGL((restrict_stream4_kernel<3, 24, 4>), dim3(nAgg), dim3(threads), 0, computeStream(), a, (const void *)V, block_to_fine, blockVol, m, aggMapOf(*this), Blk4{}); break;
    default: hipLaunchKernelGGL((restrict_stream4_kernel<3, 32, 4>), dim3(nAgg), dim3(threads), 0, computeStream(), a, (const void *)V, block_to_fine, blockVol, m, aggMapOf(*this), Blk4{}); break;
  }
  HIP_CHECK(hipGetLastError());
  flops_ += 4 * 8ull * fineSpin * fineColor * Nvec * fineVol;
}
void Transfer::P4(ColorSpinorField *const fine[4], const ColorSpinorField *const coarse[4]) const {
  if (!canQuad()) errorQuda("four-source prolongator: fine level with 4^4-type aggregates only");
  const bool sub = fine[0]->SiteSubset() == QUDA_PARITY_SITE_SUBSET;
  if (sub && site_subset != QUDA_PARITY_SITE_SUBSET) errorQuda("single-parity fine field but the transfer is set to full fields");
  Src4 a;
  for (int s = 0; s < 4; s++) {
    if (fine[s]->Nspin() != fineSpin || fine[s]->Ncolor() != fineColor || (long)fine[s]->VolumeCB() * 2 != fineVol || fine[s]->SiteSubset() != fine[0]->SiteSubset() || fine[s]->Precision() != QUDA_SINGLE_PRECISION)
      errorQuda("fine field %d does not match the transfer operator", s);
    a.f[s] = fineVec(*fine[s], sub ? (int)subset_parity : -1);
    a.c[s] = coarseVec(*coarse[s]);
  }
  const int threads = (blockVol + 63) / 64 * 64;
  const AggMap amap = aggMapOf(*this);
  if (g_acctOn) {
    const double frac = sub ? 0.5 : 1.0;
    acct("prolong4_kernel", frac * fineVol * ((double)fineSpin * fineColor * Nvec * 8 + 4.0 * fineSpin * fineColor * 8.0) + 4.0 * nAgg * 2 * Nvec * 8, "coarse -> level 0, 4 sources");
  }
  switch (Nvec) {
    case 8: hipLaunchKernelGGL((prolong4_kernel<3, 8, 4>), dim3(nAgg), dim3(threads), 0, computeStream(), a, (const void *)V, block_to_fine, blockVol, amap, Blk4{}); break;
    case 24: hipLaunchKernelGGL((prolong4_kernel<3, 24, 4>), dim3(nAgg), dim3(threads), 0, computeStream(), a, (const void *)V, block_to_fine, blockVol, amap, Blk4{}); break;
    default: hipLaunchKernelGGL((prolong4_kernel<3, 32, 4>), dim3(nAgg), dim3(threads), 0, computeStream(), a, (const void *)V, block_to_fine, blockVol, amap, Blk4{}); break;
  }
  HIP_CHECK(hipGetLastError());
  flops_ += 4 * 8ull * fineSpin * fineColor * Nvec * fineVol;
}

// ... with the fine vectors in four columns of a pair-major block field of one parity (block.h): the multi-source smoother's residuals in, its solutions out
void Transfer::R4Block(ColorSpinorField *const coarse[4], const float2 *panel, int nrhs, int col0) const {
  if (!canQuad() || site_subset != QUDA_PARITY_SITE_SUBSET) errorQuda("four-source restrictor on block fields: fine level, single-parity transfers");
  Src4 a;
  memset(&a, 0, sizeof(a));
  for (int s = 0; s < 4; s++) a.c[s] = coarseVec(*coarse[s]);
  const Blk4 blk = {reinterpret_cast<float4 *>(const_cast<float2 *>(panel)), nrhs, col0, subset_parity == QUDA_ODD_PARITY ? 1 : 0, (int)(fineVol / 2), 0};
  MaskArg m;
  m.dir = -1; m.boundary = 0; m.pm = parityMajor ? 1 : 0;
  for (int d = 0; d < 4; d++) { m.bs[d] = geo_bs[d]; m.single[d] = Xc[d] == 1; }
  const int threads = (blockVol + 63) / 64 * 64;
  acct("restrict_stream4_kernel", 0.5 * fineVol * ((double)fineSpin * fineColor * Nvec * 8 + 4.0 * fineSpin * fineColor * 8.0) + 4.0 * nAgg * 2 * Nvec * 8, "level 0 (block columns) -> coarse, 4 sources");
  switch (Nvec) {
    case 8: hipLaunchKernelGGL((restrict_stream4_kernel<3, 8, 4, true>), dim3(nAgg), dim3(threads), 0, computeStream(), a, (const void *)V, block_to_fine, blockVol, m, aggMapOf(*this), blk); break;
    case 24: hipLaunchKernelGGL((restrict_stream4_kernel<3, 24, 4, true>), dim3(nAgg), dim3(threads), 0, computeStream(), a, (const void *)V, block_to_fine, blockVol, m, aggMapOf(*this), blk); break;
    default: hipLaunchKernelGGL((restrict_stream4_kernel<3, 32, 4, true>), dim3(nAgg), dim3(threads), 0, computeStream(), a, (const void *)V, block_to_fine, blockVol, m, aggMapOf(*this), blk); break;
  }
  HIP_CHECK(hipGetLastError());
  flops_ += 4 * 8ull * fineSpin * fineColor * Nvec * fineVol / 2;
}
void Transfer::P4Block(float2 *panel, int nrhs, int col0, bool accumulate, const ColorSpinorField *const coarse[4]) const {
  if (!canQuad() || site_subset != QUDA_PARITY_SITE_SUBSET) errorQuda("four-source prolongator on block fields: fine level, single-parity transfers");
  Src4 a;
  memset(&a, 0, sizeof(a));
  for (int s = 0; s < 4; s++) a.c[s] = coarseVec(*coarse[s]);
  const Blk4 blk = {reinterpret_cast<float4 *>(panel), nrhs, col0, subset_parity == QUDA_ODD_PARITY ? 1 : 0, (int)(fineVol / 2), accumulate ? 1 : 0};
  const int threads = (blockVol + 63) / 64 * 64;
  const AggMap amap = aggMapOf(*this);
  acct("prolong4_kernel", 0.5 * fineVol * ((double)fineSpin * fineColor * Nvec * 8 + (accumulate ? 8.0 : 4.0) * fineSpin * fineColor * 8.0) + 4.0 * nAgg * 2 * Nvec * 8, "coarse -> level 0 (block columns), 4 sources");
  switch (Nvec) {
    case 8: hipLaunchKernelGGL((prolong4_kernel<3, 8, 4, true>), dim3(nAgg), dim3(threads), 0, computeStream(), a, (const void *)V, block_to_fine, blockVol, amap, blk); break;
    case 24: hipLaunchKernelGGL((prolong4_kernel<3, 24, 4, true>), dim3(nAgg), dim3(threads), 0, computeStream(), a, (const void *)V, block_to_fine, blockVol, amap, blk); break;
    default: hipLaunchKernelGGL((prolong4_kernel<3, 32, 4, true>), dim3(nAgg), dim3(threads), 0, computeStream(), a, (const void *)V, block_to_fine, blockVol, amap, blk); break;
  }
  HIP_CHECK(hipGetLastError());
  flops_ += 4 * 8ull * fineSpin * fineColor * Nvec * fineVol / 2;
}

void Transfer::column(ColorSpinorField &fine, int j) const {
  if (fine.SiteSubset() != QUDA_FULL_SITE_SUBSET || fine.Nspin() != fineSpin || fine.Ncolor() != fineColor || fine.Volume() != fineVol) errorQuda("fine field does not match the transfer operator");
  if (j < 0 || j >= 2 * Nvec) errorQuda("coarse component %d of %d", j, 2 * Nvec);
  const FineVec out = fineVec(fine);
  const int bs = 256;
  const long total = fineVol;
  if (fineSpin == 4) hipLaunchKernelGGL((column_kernel<4, 3, 4>), dim3((unsigned)((total + bs - 1) / bs)), dim3(bs), 0, computeStream(), out, (const float4 *)V, block_to_fine, blockVol, spin_bs, Nvec, j, total);
  else {
    switch (fineColor) {
      case 4: hipLaunchKernelGGL((column_kernel<2, 4, 2>), dim3((unsigned)((total + bs - 1) / bs)), dim3(bs), 0, computeStream(), out, (const float4 *)V, block_to_fine, blockVol, spin_bs, Nvec, j, total); break;
      case 8: hipLaunchKernelGGL((column_kernel<2, 8, 2>), dim3((unsigned)((total + bs - 1) / bs)), dim3(bs), 0, computeStream(), out, (const float4 *)V, block_to_fine, blockVol, spin_bs, Nvec, j, total); break;
      case 24: hipLaunchKernelGGL((column_kernel<2, 24, 2>), dim3((unsigned)((total + bs - 1) / bs)), dim3(bs), 0, computeStream(), out, (const float4 *)V, block_to_fine, blockVol, spin_bs, Nvec, j, total); break;
      case 32: hipLaunchKernelGGL((column_kernel<2, 32, 2>), dim3((unsigned)((total + bs - 1) / bs)), dim3(bs), 0, computeStream(), out, (const float4 *)V, block_to_fine, blockVol, spin_bs, Nvec, j, total); break;
      default: errorQuda("column extraction for %d fine colours not instantiated", fineColor);
    }
  }
  HIP_CHECK(hipGetLastError());
}

bool Transfer::canSplit4() const { return fineSpin == 4 && fineColor == 3 && blockVol > 32 && blockVol <= 256; }

void Transfer::RSplit4(ColorSpinorField *const leaving[4], ColorSpinorField *const staying[4], ColorSpinorField *const fine[4], const int dir[4]) const {
  if (!canSplit4()) errorQuda("the four-way split restriction is built for the fine level");
  Multi4 a;
  for (int q = 0; q < 4; q++) {
    if (fine[q]->SiteSubset() != QUDA_FULL_SITE_SUBSET || fine[q]->Volume() != fineVol) errorQuda("fine field does not match the transfer operator");
    a.in[q] = fineVec(*fine[q]); a.out[q] = coarseVec(*leaving[q]); a.out2[q] = coarseVec(*staying[q]); a.dir[q] = dir[q];
  }
  MaskArg m;
  m.dir = 0; m.boundary = 1; m.pm = parityMajor ? 1 : 0;
  for (int d = 0; d < 4; d++) { m.bs[d] = geo_bs[d]; m.single[d] = Xc[d] == 1; }
  const int threads = (blockVol + 63) / 64 * 64;
  switch (Nvec) {
    case 4: hipLaunchKernelGGL((restrict4_kernel<4, 3, 4, 4>), dim3(nAgg), dim3(threads), 0, computeStream(), a, (const float4 *)V, block_to_fine, blockVol, spin_bs, m, aggMapOf(*this)); break;
    case 8: hipLaunchKernelGGL((restrict4_kernel<4, 3, 8, 4>), dim3(nAgg), dim3(threads), 0, computeStream(), a, (const float4 *)V, block_to_fine, blockVol, spin_bs, m, aggMapOf(*this)); break;
    case 24: hipLaunchKernelGGL((restrict4_kernel<4, 3, 24, 4>), dim3(nAgg), dim3(threads), 0, computeStream(), a, (const float4 *)V, block_to_fine, blockVol, spin_bs, m, aggMapOf(*this)); break;
    case 32: hipLaunchKernelGGL((restrict4_kernel<4, 3, 32, 4>), dim3(nAgg), dim3(threads), 0, computeStream(), a, (const float4 *)V, block_to_fine, blockVol, spin_bs, m, aggMapOf(*this)); break;
    default: errorQuda("Nvec = %d not instantiated", Nvec);
  }
  HIP_CHECK(hipGetLastError());
  flops_ += 4 * 8ull * fineSpin * fineColor * Nvec * fineVol;
}

}  // namespace quda
