// device_io.h — register <-> HBM accessors for the planar field layouts (gfx950).
//
// All loads/stores are 16 bytes per lane where the layout allows it (double2 / float4; short4 = 8 B for
// the 16-bit formats), unit stride across the wave, so one wave instruction moves 1 KiB (or 512 B) of
// consecutive addresses.  16-bit storage is int16 fixed point with an fp32 per-site (spinor) / per-chiral-
// block (clover) scale, as in the reference (lib/io_spinor.h:49-62, :282-319; MAX_SHORT quda_internal.h:30);
// links use a fixed scale (|U_ij| <= 1).
#pragma once

#include <hip/hip_runtime.h>

namespace quda {

constexpr float kShortMax = 32767.0f;
constexpr float kShortInv = 1.0f / 32767.0f;

template <typename T> struct Store;
template <> struct Store<double> {
  using real = double;
  static constexpr int N = 2;       // reals per 16-byte vector
  static constexpr bool fixed = false;
};
template <> struct Store<float> {
  using real = float;
  static constexpr int N = 4;
  static constexpr bool fixed = false;
};
template <> struct Store<short> {
  using real = float;
  static constexpr int N = 4;
  static constexpr bool fixed = true;
};

// ---- generic planar vector field of NR reals per site ----
template <typename T, int NR> struct Planar;

// Addressing: every planar block is accessed through a raw buffer descriptor (4 SGPRs: base, bytes) with
// ONE 32-bit per-lane byte offset (VGPR) and the plane offset k*stride*16 as the instruction's scalar
// soffset -> `buffer_load_dwordx4 v[..], v_off, s[rsrc], s_plane offen`.  Compared with flat global loads
// this removes a 64-bit address pair per plane (21 planes per hop in fp64), which is what lets the stencil
// double-buffer two directions in registers without spilling; out-of-range lanes read 0 instead of faulting.
// One block (NR reals x stride sites) must stay below 4 GiB, true for every local volume of interest.
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));

template <typename T, int NR> __device__ __forceinline__ __amdgpu_buffer_rsrc_t planar_rsrc(const void *base, int stride) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)((unsigned)stride * (unsigned)(NR * sizeof(T))), 0x00020000);
}

template <int NR> struct Planar<double, NR> {
  template <int AUX = 0> static __device__ __forceinline__ void load(double *r, const void *base, int stride, int x, const float *, int) {
    const __amdgpu_buffer_rsrc_t rs = planar_rsrc<double, NR>(base, stride);
    const int off = x * 16;
#pragma unroll
    for (int k = 0; k < NR / 2; k++) {
      const double2 t = __builtin_bit_cast(double2, __builtin_amdgcn_raw_buffer_load_b128(rs, off, k * stride * 16, AUX));
      r[2 * k] = t.x;
      r[2 * k + 1] = t.y;
    }
  }
  template <int AUX = 0> static __device__ __forceinline__ void store(const double *r, void *base, int stride, int x, float *, int) {
    const __amdgpu_buffer_rsrc_t rs = planar_rsrc<double, NR>(base, stride);
    const int off = x * 16;
#pragma unroll
    for (int k = 0; k < NR / 2; k++)
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, make_double2(r[2 * k], r[2 * k + 1])), rs, off, k * stride * 16, AUX);
  }
};

template <int NR> struct Planar<float, NR> {
  template <int AUX = 0> static __device__ __forceinline__ void load(float *r, const void *base, int stride, int x, const float *, int) {
    const __amdgpu_buffer_rsrc_t rs = planar_rsrc<float, NR>(base, stride);
    const int off = x * 16;
#pragma unroll
    for (int k = 0; k < NR / 4; k++) {
      const float4 t = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, k * stride * 16, AUX));
      r[4 * k] = t.x; r[4 * k + 1] = t.y; r[4 * k + 2] = t.z; r[4 * k + 3] = t.w;
    }
    if (NR % 4) {  // trailing float2 plane (18-real links)
      const float2 t = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(rs, x * 8, (NR / 4) * stride * 16, AUX));
      r[NR - 2] = t.x; r[NR - 1] = t.y;
    }
  }
  template <int AUX = 0> static __device__ __forceinline__ void store(const float *r, void *base, int stride, int x, float *, int) {
    const __amdgpu_buffer_rsrc_t rs = planar_rsrc<float, NR>(base, stride);
    const int off = x * 16;
#pragma unroll
    for (int k = 0; k < NR / 4; k++)
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, make_float4(r[4 * k], r[4 * k + 1], r[4 * k + 2], r[4 * k + 3])), rs, off,
                                             k * stride * 16, AUX);
    if (NR % 4) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, make_float2(r[NR - 2], r[NR - 1])), rs, x * 8, (NR / 4) * stride * 16, AUX);
  }
};

// 16-bit fixed point.  norm == nullptr: fixed unit scale (links); else per-site scale at norm[nidx].
// Planes of EIGHT int16 (16 bytes per lane, like the fp32 / fp64 orders) followed, where NR is not a multiple of 8, by one
// narrower plane with the remaining 4 or 2 values: [NR/8 planes x stride x 16 B][tail plane].  The reference's 16-bit order
// is short4 (8 bytes per lane, lib/io_spinor.h:49-62); on gfx950 8-byte-per-lane loads run at 0.54-0.70 of the 16-byte
// rate (MI355X_MICROARCH.md), and the 16-bit twisted-clover stencil measured 4.9 TB/s of real traffic against 6.3 for the
// 16-byte formats with only 6 % traffic overhead (profiles/r02a_before_tmc_i16_32x4_*) — so the vector length, not the
// bytes, was what held it back.  Block size (NR x stride x 2 B), stride, norm array and alignment are unchanged.
typedef short short8_t __attribute__((ext_vector_type(8)));
template <int NR> struct Planar<short, NR> {
  static constexpr int NV = NR / 8, TAIL = NR % 8;   // TAIL: 0, 4 (one 8-byte plane) or 2 (one 4-byte plane)
  static_assert(TAIL == 0 || TAIL == 4 || TAIL == 2, "16-bit planar blocks hold 8k, 8k+4 or 8k+2 values");
  template <int AUX = 0> static __device__ __forceinline__ void load(float *r, const void *base, int stride, int x, const float *norm, int nidx) {
    // AUX with sc1 (bit 4): the block may have been written by another GPU / process -> the scale is read at system scope too
    const float s = norm ? ((AUX & 16) ? __hip_atomic_load(&norm[nidx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : norm[nidx]) * kShortInv : kShortInv;
    const __amdgpu_buffer_rsrc_t rs = planar_rsrc<short, NR>(base, stride);
    const int off = x * 16;
#pragma unroll
    for (int k = 0; k < NV; k++) {
      const short8_t t = __builtin_bit_cast(short8_t, __builtin_amdgcn_raw_buffer_load_b128(rs, off, k * stride * 16, AUX));
#pragma unroll
      for (int j = 0; j < 8; j++) r[8 * k + j] = t[j] * s;
    }
    if (TAIL == 4) {
      const short4 t = __builtin_bit_cast(short4, __builtin_amdgcn_raw_buffer_load_b64(rs, x * 8, NV * stride * 16, AUX));
      r[NR - 4] = t.x * s; r[NR - 3] = t.y * s; r[NR - 2] = t.z * s; r[NR - 1] = t.w * s;
    } else if (TAIL == 2) {
      const short2 t = __builtin_bit_cast(short2, __builtin_amdgcn_raw_buffer_load_b32(rs, x * 4, NV * stride * 16, AUX));
      r[NR - 2] = t.x * s; r[NR - 1] = t.y * s;
    }
  }
  static __device__ __forceinline__ short q16(float v) { return (short)__float2int_rn(fminf(fmaxf(v, -kShortMax), kShortMax)); }
  template <int AUX = 0> static __device__ __forceinline__ void store(const float *r, void *base, int stride, int x, float *norm, int nidx) {
    float s = kShortMax;
    if (norm) {
      float m = 0.f;
#pragma unroll
      for (int k = 0; k < NR; k++) m = fmaxf(m, fabsf(r[k]));
      if (AUX & 16) __hip_atomic_store(&norm[nidx], m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);  // write-through like the payload
      else norm[nidx] = m;
      s = m > 0.f ? kShortMax / m : 0.f;
    }
    const __amdgpu_buffer_rsrc_t rs = planar_rsrc<short, NR>(base, stride);
    const int off = x * 16;
#pragma unroll
    for (int k = 0; k < NV; k++) {
      short8_t t;
#pragma unroll
      for (int j = 0; j < 8; j++) t[j] = q16(r[8 * k + j] * s);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, t), rs, off, k * stride * 16, AUX);
    }
    if (TAIL == 4)
      __builtin_amdgcn_raw_buffer_store_b64(
          __builtin_bit_cast(u32x2_t, make_short4(q16(r[NR - 4] * s), q16(r[NR - 3] * s), q16(r[NR - 2] * s), q16(r[NR - 1] * s))), rs, x * 8,
          NV * stride * 16, AUX);
    else if (TAIL == 2)
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, make_short2(q16(r[NR - 2] * s), q16(r[NR - 1] * s))), rs, x * 4,
                                            NV * stride * 16, AUX);
  }
};

// ---- raw register image of one site of a planar block: the LOADS of Planar<T, NR>::load without anything that depends on the
// loaded data.  The stencil requests the operands of hop d + 1 before the arithmetic of hop d and fences the two phases against the
// scheduler; a conversion (16-bit -> fp32: 24 + 18 v_cvt + scale per hop) or a row reconstruction inside the request phase sits on
// the wrong side of that fence — it waits for the data right there and the hop is not overlapped at all.  load() only requests;
// unpack() is called at the start of the arithmetic phase.  Register cost of a 16-bit block in flight: NR / 2 + 1 instead of NR. ----
template <typename T, int NR> struct RawBlock {
  using real = typename Store<T>::real;
  real r[NR];
  template <int AUX = 0> __device__ __forceinline__ void load(const void *base, int stride, int x, const float *norm, int nidx) {
    Planar<T, NR>::template load<AUX>(r, base, stride, x, norm, nidx);
  }
  // first 12 reals only, from a 12-real block (half spinor of a ghost zone)
  template <int AUX = 0> __device__ __forceinline__ void load12(const void *base, int stride, int x, const float *norm, int nidx) {
    Planar<T, 12>::template load<AUX>(r, base, stride, x, norm, nidx);
  }
  __device__ __forceinline__ void unpack(real *out) const {
#pragma unroll
    for (int k = 0; k < NR; k++) out[k] = r[k];
  }
};
template <int NR> struct RawBlock<short, NR> {
  static constexpr int NV = NR / 8, TAIL = NR % 8, NW = NR / 2;
  unsigned w[NW];   // plane k (8 values) in w[4k .. 4k+3], then the tail plane; value j of a dword pair in the low / high half
  float nrm;        // per-site scale (spinors, clover blocks); links: fixed unit scale
  template <int AUX = 0> __device__ __forceinline__ void load(const void *base, int stride, int x, const float *norm, int nidx) {
    nrm = norm ? ((AUX & 16) ? __hip_atomic_load(&norm[nidx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : norm[nidx]) : 1.0f;
    const __amdgpu_buffer_rsrc_t rs = planar_rsrc<short, NR>(base, stride);
    const int off = x * 16;
#pragma unroll
    for (int k = 0; k < NV; k++) {
      const u32x4_t t = __builtin_amdgcn_raw_buffer_load_b128(rs, off, k * stride * 16, AUX);
      w[4 * k] = t.x; w[4 * k + 1] = t.y; w[4 * k + 2] = t.z; w[4 * k + 3] = t.w;
    }
    if (TAIL == 4) {
      const u32x2_t t = __builtin_amdgcn_raw_buffer_load_b64(rs, x * 8, NV * stride * 16, AUX);
      w[4 * NV] = t.x; w[4 * NV + 1] = t.y;
    } else if (TAIL == 2) {
      w[4 * NV] = __builtin_amdgcn_raw_buffer_load_b32(rs, x * 4, NV * stride * 16, AUX);
    }
  }
  // a 12-value block has its values 0..11 in the same dwords as the first 12 of a longer one (plane 0, then 4 values)
  template <int AUX = 0> __device__ __forceinline__ void load12(const void *base, int stride, int x, const float *norm, int nidx) {
    static_assert(NR >= 12, "half-spinor image needs 6 dwords");
    nrm = norm ? ((AUX & 16) ? __hip_atomic_load(&norm[nidx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : norm[nidx]) : 1.0f;
    const __amdgpu_buffer_rsrc_t rs = planar_rsrc<short, 12>(base, stride);
    const u32x4_t t = __builtin_amdgcn_raw_buffer_load_b128(rs, x * 16, 0, AUX);
    w[0] = t.x; w[1] = t.y; w[2] = t.z; w[3] = t.w;
    const u32x2_t u = __builtin_amdgcn_raw_buffer_load_b64(rs, x * 8, stride * 16, AUX);
    w[4] = u.x; w[5] = u.y;
  }
  __device__ __forceinline__ void unpack(float *out) const {
    const float s = nrm * kShortInv;
#pragma unroll
    for (int k = 0; k < NW; k++) {
      out[2 * k] = (float)(short)(w[k] & 0xffffu) * s;
      out[2 * k + 1] = (float)((int)w[k] >> 16) * s;
    }
  }
  // the integers as they are stored; the caller applies scale() to the (fewer) results of its linear arithmetic
  __device__ __forceinline__ void unpack_unscaled(float *out) const {
#pragma unroll
    for (int k = 0; k < NW; k++) {
      out[2 * k] = (float)(short)(w[k] & 0xffffu);
      out[2 * k + 1] = (float)((int)w[k] >> 16);
    }
  }
  __device__ __forceinline__ float scale() const { return nrm * kShortInv; }
};

// bytes of one planar block of NR reals x stride sites
template <typename T> __host__ __device__ constexpr size_t storeSize() { return sizeof(T); }

// ---- packed fp32 complex arithmetic: a complex number is an aligned register pair (re, im); a complex multiply-add is TWO
// v_pk_fma_f32 — op_sel / op_sel_hi choose which half of each source feeds the low / high result lane, neg_lo / neg_hi the sign —
// instead of four scalar multiply-adds.  The compiler's SLP vectoriser builds the same instructions but pays for every pair with
// v_mov shuffles (it is switched off for the stencil); written by hand the operands are already where they have to be. ----
typedef float pkf2 __attribute__((ext_vector_type(2)));
namespace pk {
__device__ __forceinline__ pkf2 cmul(pkf2 u, pkf2 h) {   // u h
  pkf2 r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(r) : "v"(u), "v"(h));
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "+v"(r) : "v"(u), "v"(h));
  return r;
}
__device__ __forceinline__ pkf2 cmac(pkf2 acc, pkf2 u, pkf2 h) {   // acc + u h
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(u), "v"(h));
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "+v"(acc) : "v"(u), "v"(h));
  return acc;
}
// y + s x, y - s x, y + i s x, y - i s x   with the real factor s in the LOW half of S
__device__ __forceinline__ pkf2 axpy(pkf2 S, pkf2 x, pkf2 y) {
  pkf2 r;
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "=v"(r) : "v"(S), "v"(x), "v"(y));
  return r;
}
__device__ __forceinline__ pkf2 axmy(pkf2 S, pkf2 x, pkf2 y) {
  pkf2 r;
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0] neg_hi:[0,1,0]" : "=v"(r) : "v"(S), "v"(x), "v"(y));
  return r;
}
__device__ __forceinline__ pkf2 aixpy(pkf2 S, pkf2 x, pkf2 y) {   // (y.re - s x.im, y.im + s x.re)
  pkf2 r;
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[0,0,1] neg_lo:[0,1,0]" : "=v"(r) : "v"(S), "v"(x), "v"(y));
  return r;
}
__device__ __forceinline__ pkf2 aixmy(pkf2 S, pkf2 x, pkf2 y) {   // (y.re + s x.im, y.im - s x.re)
  pkf2 r;
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[0,0,1] neg_hi:[0,1,0]" : "=v"(r) : "v"(S), "v"(x), "v"(y));
  return r;
}
__device__ __forceinline__ pkf2 add(pkf2 x, pkf2 y) {
  pkf2 r;
  asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
  return r;
}
__device__ __forceinline__ pkf2 cmac_conj(pkf2 acc, pkf2 u, pkf2 h) {   // acc + conj(u) h
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(u), "v"(h));
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_hi:[1,0,0]" : "+v"(acc) : "v"(u), "v"(h));
  return acc;
}
// (d x.re, d x.im) with the real d in the low (HI = false) or high half of D
template <bool HI> __device__ __forceinline__ pkf2 rscale(pkf2 D, pkf2 x) {
  pkf2 r;
  if (HI) asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(r) : "v"(D), "v"(x));
  else asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(r) : "v"(D), "v"(x));
  return r;
}
__device__ __forceinline__ pkf2 ld(const float *a, int i) { return (pkf2){a[i], a[i + 1]}; }
__device__ __forceinline__ void st(float *a, int i, pkf2 v) { a[i] = v.x; a[i + 1] = v.y; }
// conj(u) h
__device__ __forceinline__ pkf2 cmulc(pkf2 u, pkf2 h) {
  pkf2 r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(r) : "v"(u), "v"(h));
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_hi:[1,0,0]" : "+v"(r) : "v"(u), "v"(h));
  return r;
}
// acc + conj(u) h
__device__ __forceinline__ pkf2 cmacc(pkf2 acc, pkf2 u, pkf2 h) {
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(u), "v"(h));
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_hi:[1,0,0]" : "+v"(acc) : "v"(u), "v"(h));
  return acc;
}
// acc - u h
__device__ __forceinline__ pkf2 cmsub(pkf2 acc, pkf2 u, pkf2 h) {
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]" : "+v"(acc) : "v"(u), "v"(h));
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_hi:[1,0,0]" : "+v"(acc) : "v"(u), "v"(h));
  return acc;
}
// element by element: a b + c, a b - c, a b
__device__ __forceinline__ pkf2 efma(pkf2 a, pkf2 b, pkf2 c) { pkf2 r; asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ pkf2 efms(pkf2 a, pkf2 b, pkf2 c) { pkf2 r; asm("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ pkf2 emul(pkf2 a, pkf2 b) { pkf2 r; asm("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
}  // namespace pk

// ---- SU(3) link: 18 reals row-major (row*6 + col*2 + re/im); R = 18 stored fully, R = 12 rows 0,1 stored,
// row 2 = conj(row0 x row1) * sign (sign carries a folded anti-periodic boundary, reference
// tests/test_util.cpp:283-296 / lib/read_gauge.h) ----
// R = 8 (reference Reconstruct<8>, include/gauge_field_order.h:516-585; lib/read_gauge.h RECONSTRUCT_8 macros): stored are
//   [arg U00, arg U20, U01.re, U01.im, U02.re, U02.im, U10.re, U10.im]   (16-bit: the two phases in units of pi)
// and the matrix M = u0 V, V in SU(3), u0 = +-1 the folded anti-periodic boundary sign, comes back from the unit length of row 0 and
// column 0 and the SU(2) rotation of the lower-right block.  The same for the pre-daggered backward links of this library's layout
// (U^dagger is as much an SU(3) matrix as U).  37 % of the link bytes of R = 18 for ~110 flops, two sincos, two sqrt and one reciprocal.
__device__ __forceinline__ void qa_sincos(float x, float *s, float *c) { *s = __sinf(x); *c = __cosf(x); }   // v_sin_f32 / v_cos_f32 (|x| <= pi: ~1e-6)
__device__ __forceinline__ void qa_sincos(double x, double *s, double *c) { sincos(x, s, c); }
template <typename real> __device__ __forceinline__ void su3_reconstruct8(real *U, const real *in, real u0, real phaseUnit) {
  const real a2r = in[2], a2i = in[3], a3r = in[4], a3i = in[5], b1r = in[6], b1i = in[7];
  const real row_sum = a2r * a2r + a2i * a2i + a3r * a3r + a3i * a3i;
  const real d0 = (real)1 - row_sum;
  const real m0 = sqrt(d0 >= (real)0 ? d0 : (real)0);
  real s0, c0, s1, c1;
  qa_sincos(in[0] * phaseUnit, &s0, &c0);
  qa_sincos(in[1] * phaseUnit, &s1, &c1);
  const real a1r = m0 * c0, a1i = m0 * s0;
  const real d1 = (real)1 - (m0 * m0 + b1r * b1r + b1i * b1i);
  const real m1 = sqrt(d1 >= (real)0 ? d1 : (real)0);
  const real c1r = m1 * c1, c1i = m1 * s1;
  const real rinv = u0 / row_sum;                 // 1 / (u0 row_sum), u0 = +-1
  // A = conj(a1) b1 ; A2 = conj(a1) c1, both times u0
  const real Ar = u0 * (a1r * b1r + a1i * b1i), Ai = u0 * (a1r * b1i - a1i * b1r);
  const real Br = u0 * (a1r * c1r + a1i * c1i), Bi = u0 * (a1r * c1i - a1i * c1r);
  // conj(c1) conj(a3) etc.: conj(x) conj(y) = conj(x y)
  const real c1a3r = c1r * a3r - c1i * a3i, c1a3i = -(c1r * a3i + c1i * a3r);
  const real c1a2r = c1r * a2r - c1i * a2i, c1a2i = -(c1r * a2i + c1i * a2r);
  const real b1a3r = b1r * a3r - b1i * a3i, b1a3i = -(b1r * a3i + b1i * a3r);
  const real b1a2r = b1r * a2r - b1i * a2i, b1a2i = -(b1r * a2i + b1i * a2r);
  U[0] = a1r; U[1] = a1i; U[2] = a2r; U[3] = a2i; U[4] = a3r; U[5] = a3i;
  U[6] = b1r; U[7] = b1i;
  U[8] = -(c1a3r + (Ar * a2r - Ai * a2i)) * rinv;  U[9] = -(c1a3i + (Ar * a2i + Ai * a2r)) * rinv;    // U11
  U[10] = (c1a2r - (Ar * a3r - Ai * a3i)) * rinv;  U[11] = (c1a2i - (Ar * a3i + Ai * a3r)) * rinv;    // U12
  U[12] = c1r; U[13] = c1i;
  U[14] = (b1a3r - (Br * a2r - Bi * a2i)) * rinv;  U[15] = (b1a3i - (Br * a2i + Bi * a2r)) * rinv;    // U21
  U[16] = -(b1a2r + (Br * a3r - Bi * a3i)) * rinv; U[17] = -(b1a2i + (Br * a3i + Bi * a3r)) * rinv;   // U22
}
// fp32 (and the 16-bit kernels, which compute in fp32): the same in packed complex arithmetic — 28 v_pk instructions + 24 scalar ones per link
// instead of 124 (the generic form above compiled to 342 multiplies, 369 multiply-adds and IEEE square roots / divisions with their fix-ups for the 8 links
// of a site: more than the rest of the stencil).  v_sqrt_f32 / v_rcp_f32 / v_sin_f32 / v_cos_f32 are 1 ulp / 1e-6: inside the 2e-5 of the fp32 goldens.
__device__ __forceinline__ void su3_reconstruct8(float *U, const float *in, float u0, float phaseUnit) {
  const pkf2 a2 = {in[2], in[3]}, a3 = {in[4], in[5]}, b1 = {in[6], in[7]};
  const float rs = a2.x * a2.x + a2.y * a2.y + a3.x * a3.x + a3.y * a3.y;
  const float d0 = fmaxf(1.f - rs, 0.f);
  const float m0 = __builtin_amdgcn_sqrtf(d0);
  const float k = phaseUnit * 0.15915494309189535f;   // v_sin_f32 / v_cos_f32 take revolutions
  const float t0 = in[0] * k, t1 = in[1] * k;
  const pkf2 a1 = {m0 * __builtin_amdgcn_cosf(t0), m0 * __builtin_amdgcn_sinf(t0)};
  const float d1 = fmaxf(1.f - d0 - b1.x * b1.x - b1.y * b1.y, 0.f);
  const float m1 = __builtin_amdgcn_sqrtf(d1);
  const pkf2 c1 = {m1 * __builtin_amdgcn_cosf(t1), m1 * __builtin_amdgcn_sinf(t1)};
  const float ri = __builtin_amdgcn_rcpf(rs);
  const pkf2 S = {u0, -u0}, RI = {ri, ri}, NRI = {-ri, -ri};
  const pkf2 A = pk::cmulc(a1, b1), B = pk::cmulc(a1, c1);   // conj(a1) b1, conj(a1) c1
  // u0 conj(x y) = S o (x y);  U11 = -(u0 conj(c1 a3) + A a2) / rs, U12 = (u0 conj(c1 a2) - A a3) / rs, U21 = (u0 conj(b1 a3) - B a2) / rs, U22 = -(u0 conj(b1 a2) + B a3) / rs
  const pkf2 u11 = pk::emul(NRI, pk::efma(S, pk::cmul(c1, a3), pk::cmul(A, a2)));
  const pkf2 u12 = pk::emul(RI, pk::efms(S, pk::cmul(c1, a2), pk::cmul(A, a3)));
  const pkf2 u21 = pk::emul(RI, pk::efms(S, pk::cmul(b1, a3), pk::cmul(B, a2)));
  const pkf2 u22 = pk::emul(NRI, pk::efma(S, pk::cmul(b1, a2), pk::cmul(B, a3)));
  pk::st(U, 0, a1); pk::st(U, 2, a2); pk::st(U, 4, a3);
  pk::st(U, 6, b1); pk::st(U, 8, u11); pk::st(U, 10, u12);
  pk::st(U, 12, c1); pk::st(U, 14, u21); pk::st(U, 16, u22);
}
// the inverse: the 8 stored reals of M = u0 V (reference Reconstruct<8>::Pack)
template <typename real> __device__ __forceinline__ void su3_pack8(real *out, const real *U, real phaseUnitInv) {
  out[0] = atan2(U[1], U[0]) * phaseUnitInv;
  out[1] = atan2(U[13], U[12]) * phaseUnitInv;
#pragma unroll
  for (int i = 2; i < 8; i++) out[i] = U[i];
}
template <typename T> struct PhaseUnit { static constexpr double value = 1.0; };
template <> struct PhaseUnit<short> { static constexpr double value = 3.14159265358979323846; };   // 16-bit storage holds phase / pi in [-1, 1]

template <typename T, int R> struct Link {
  using real = typename Store<T>::real;
  using Raw = RawBlock<T, R>;
  // request / finish pair for the fenced stencil pipeline (RawBlock): finish converts and, for R = 12, rebuilds the third row
  template <int AUX = 0> static __device__ __forceinline__ void request(Raw &raw, const void *blk, int stride, int x) { raw.template load<AUX>(blk, stride, x, nullptr, 0); }
  static __device__ __forceinline__ void finish(real *U, const Raw &raw, real sign) {
    if constexpr (R == 8) {
      real in[8];
      raw.unpack(in);
      su3_reconstruct8(U, in, sign, (real)PhaseUnit<T>::value);
    } else {
      raw.unpack(U);
      if (R == 12) third_row(U, sign);
    }
  }
  static __device__ __forceinline__ void third_row(real *U, real sign) {
    // c = conj(a x b)
    if constexpr (sizeof(real) == 4) {
      // fp32: five packed instructions per element (a_j b_k, - a_k b_j, conjugate and sign in one multiply) instead of eleven scalar ones
      const pkf2 S = {sign, -sign};
      const pkf2 a0 = pk::ld(U, 0), a1 = pk::ld(U, 2), a2 = pk::ld(U, 4), b0 = pk::ld(U, 6), b1 = pk::ld(U, 8), b2 = pk::ld(U, 10);
      pk::st(U, 12, pk::emul(S, pk::cmsub(pk::cmul(a1, b2), a2, b1)));
      pk::st(U, 14, pk::emul(S, pk::cmsub(pk::cmul(a2, b0), a0, b2)));
      pk::st(U, 16, pk::emul(S, pk::cmsub(pk::cmul(a0, b1), a1, b0)));
      return;
    }
#define QA_CROSS(i, j, k)                                                                               \
  U[12 + 2 * i] = sign * ((U[2 * j] * U[6 + 2 * k] - U[2 * j + 1] * U[6 + 2 * k + 1]) -                \
                          (U[2 * k] * U[6 + 2 * j] - U[2 * k + 1] * U[6 + 2 * j + 1]));                \
  U[12 + 2 * i + 1] = -sign * ((U[2 * j] * U[6 + 2 * k + 1] + U[2 * j + 1] * U[6 + 2 * k]) -           \
                               (U[2 * k] * U[6 + 2 * j + 1] + U[2 * k + 1] * U[6 + 2 * j]));
    QA_CROSS(0, 1, 2)
    QA_CROSS(1, 2, 0)
    QA_CROSS(2, 0, 1)
#undef QA_CROSS
  }
  template <int AUX = 0> static __device__ __forceinline__ void load(real *U, const void *blk, int stride, int x, real sign) {
    if constexpr (R == 8) {
      real in[8];
      Planar<T, 8>::template load<AUX>(in, blk, stride, x, nullptr, 0);
      su3_reconstruct8(U, in, sign, (real)PhaseUnit<T>::value);
      return;
    }
    Planar<T, R>::template load<AUX>(U, blk, stride, x, nullptr, 0);
    if (R == 12) third_row(U, sign);
  }
};


// out(3 complex) = U(3x3) * in(3 complex); fp32: 18 packed instructions instead of 36 scalar ones
__device__ __forceinline__ void su3_mv(float *o, const float *U, const float *v) {
  const pkf2 v0 = pk::ld(v, 0), v1 = pk::ld(v, 2), v2 = pk::ld(v, 4);
#pragma unroll
  for (int r = 0; r < 3; r++) {
    pkf2 a = pk::cmul(pk::ld(U, r * 6), v0);
    a = pk::cmac(a, pk::ld(U, r * 6 + 2), v1);
    a = pk::cmac(a, pk::ld(U, r * 6 + 4), v2);
    pk::st(o, 2 * r, a);
  }
}
template <typename real> __device__ __forceinline__ void su3_mv(real *o, const real *U, const real *v) {
#pragma unroll
  for (int r = 0; r < 3; r++) {
    real re = U[r * 6 + 0] * v[0] - U[r * 6 + 1] * v[1];
    real im = U[r * 6 + 0] * v[1] + U[r * 6 + 1] * v[0];
    re += U[r * 6 + 2] * v[2] - U[r * 6 + 3] * v[3];
    im += U[r * 6 + 2] * v[3] + U[r * 6 + 3] * v[2];
    re += U[r * 6 + 4] * v[4] - U[r * 6 + 5] * v[5];
    im += U[r * 6 + 4] * v[5] + U[r * 6 + 5] * v[4];
    o[2 * r] = re;
    o[2 * r + 1] = im;
  }
}

// Hermitian 6x6 chiral block in packed order (6 real diagonal + 15 complex strictly-lower entries stored
// column by column; reference tests/clover_reference.cpp:45-53) times 6 complex.
// fp32: 11 packed instructions per row instead of 22 scalar ones
__device__ __forceinline__ void clover_block_mv(float *o, const float *C, const float *v) {
  pkf2 x[6];
#pragma unroll
  for (int j = 0; j < 6; j++) x[j] = pk::ld(v, 2 * j);
#pragma unroll
  for (int i = 0; i < 6; i++) {
    const pkf2 D = pk::ld(C, i & ~1);
    pkf2 a = (i & 1) ? pk::rscale<true>(D, x[i]) : pk::rscale<false>(D, x[i]);
#pragma unroll
    for (int j = 0; j < 6; j++) {
      if (j < i) {  // lower triangle: L(i,j)
        const int k = 15 - (6 - j) * (5 - j) / 2 + i - j - 1;
        a = pk::cmac(a, pk::ld(C, 6 + 2 * k), x[j]);
      } else if (j > i) {  // upper triangle: conj(L(j,i))
        const int k = 15 - (6 - i) * (5 - i) / 2 + j - i - 1;
        a = pk::cmac_conj(a, pk::ld(C, 6 + 2 * k), x[j]);
      }
    }
    pk::st(o, 2 * i, a);
  }
}
template <typename real> __device__ __forceinline__ void clover_block_mv(real *o, const real *C, const real *v) {
#pragma unroll
  for (int i = 0; i < 6; i++) {
    real re = C[i] * v[2 * i], im = C[i] * v[2 * i + 1];
#pragma unroll
    for (int j = 0; j < 6; j++) {
      if (j < i) {  // lower triangle: L(i,j)
        const int k = 15 - (6 - j) * (5 - j) / 2 + i - j - 1;
        const real lr = C[6 + 2 * k], li = C[6 + 2 * k + 1];
        re += lr * v[2 * j] - li * v[2 * j + 1];
        im += lr * v[2 * j + 1] + li * v[2 * j];
      } else if (j > i) {  // upper triangle: conj(L(j,i))
        const int k = 15 - (6 - i) * (5 - i) / 2 + j - i - 1;
        const real lr = C[6 + 2 * k], li = -C[6 + 2 * k + 1];
        re += lr * v[2 * j] - li * v[2 * j + 1];
        im += lr * v[2 * j + 1] + li * v[2 * j];
      }
    }
    o[2 * i] = re;
    o[2 * i + 1] = im;
  }
}

}  // namespace quda
