// dslash.hip — fine-grid even-odd Wilson / twisted-mass / twisted-clover stencil for gfx950 (MI355X).
//
// What it computes (reference semantics): tests/wilson_dslash_reference.cpp:106-133 (hop), :233-263 (twist),
// tests/clover_reference.cpp:19-64, :203-232 (clover, twisted clover); device-side behaviour of
// lib/dslash_core/tm_dslash_gt200_core.h / tmc_dslash_*_core.h (epilogues) — re-designed, not translated:
//
//  * one lane = one checkerboard site; every global access is a 16-byte (fp64/fp32) or 8-byte (16-bit)
//    per-lane, unit-stride stream, i.e. 1 KiB / 512 B per wave instruction;
//  * links come from the bidirectional pre-daggered layout (fields.h): no neighbour index and no dagger
//    branch on the link side, 8 perfectly aligned streams per site;
//  * spin basis is chiral (DeGrand-Rossi): each of the 8 projections keeps two independent colour vectors
//    (h0,h1), the other two rows are +-1/+-i multiples; the twist is a per-chirality complex scale and
//    the clover term two Hermitian 6x6 blocks — all fused into the epilogue, nothing is re-read;
//  * spinor neighbour re-use is left to L2/MALL: the block index is re-mapped so that each XCD (private
//    4 MiB L2) sweeps a contiguous slab of time slices instead of the default round-robin interleave;
//  * index arithmetic uses multiply-high fast division (no software integer divide on gfx950).
//
// Roofline: HBM-bound; algorithmic bytes per site = 8 R P + 24 P (in) + 24 P (out) [+ 24 P xpay]
// [+ 144 P clover & inverse] (+ norms for 16-bit), SURVEY.md section 8d.
#include "dslash.h"

#include <cstring>
#include <map>
#include <string>
#include <type_traits>
#include <vector>

#include "blas.h"
#include "device_io.h"
#include "halo.h"
#include "block.h"
#include "tune.h"

namespace quda {

// ---- face packing: spin-project the boundary sites of the INPUT field into the send buffers of every partitioned
// dimension in one launch (reference packFaceWilsonKernel / packTwistedFaceWilsonKernel, lib/dslash_pack.cu:272, :610) ----
template <typename real> struct PackArg {
  const void *in;
  const float *inNorm;
  int sp_stride;
  int X[4];         // full local extents
  int parity_in;    // parity of the input field
  real sfwd, a;
  char *send[4][2]; // [dim][0: to the -dim neighbour, 1: to the +dim neighbour]
  int faceCB[4];
  int normOff[4];
  int start[9];     // prefix offsets of the 8 (dim, dir) thread ranges
  // peer-store transport: send[][] point into the NEIGHBOURS' ghost zones; the last block to finish raises the flags there
  unsigned llFlag[4];      // flag-in-data value of this exchange per dimension (use count of the (dimension, buffer) zone)
  int llFormat;            // wire format of the peer-store ghost zones: 0 flag-in-data {word, flag, word, flag} vectors, 1 self-validating 16-byte atoms {3 words, flag} (GhostLL)
  unsigned long long *timeline;
};

template <typename real> struct DslashArg {
  void *out;
  float *outNorm;
  const void *in;
  const float *inNorm;
  const void *x;
  const float *xNorm;
  const char *gauge;  // base of this parity's 8 direction blocks
  size_t link_bytes;
  const void *clA, *clAinv;
  const float *clAn, *clAinvN;
  int sp_stride, g_stride, cl_stride;
  int Vh, Xh, Y, Z, T;
  FastDiv dXh, dY, dZ;
  int parity, mode, xpay;
  real sfwd;              // +1 no dagger, -1 dagger (selects P-/+ per reference :122)
  real a, b, k;
  real tsign_fwd, tsign_bwd;  // recon-12: sign of the reconstructed row of t-links on the boundary slices
  int nblocks, xcd_q, xcd_r;
  int ts, bps;                // time-slab interleave: ts slices per slab, bps blocks per time slice (ts = 0: off)
  // plane-tiled block order (tiled != 0): a block is chunk yc of the P chunks of one (z, t) plane; the 8 XCDs split the (z, t)
  // lattice nxz x (8 / nxz) ways and each walks its region tile by tile (tz x tt planes per tile, t fastest inside a tile)
  int tiled, P, nxz, Zs, Ts, tz, tt;
  FastDiv dNxz, dPerTile, dNtz, dTzTt, dTt, dPTt;
  // grid-decomposed lattices (halo.h)
  int commMask;               // bit d set: dimension d is partitioned
  const int *blist;           // exterior kernel: checkerboard indices of the boundary sites
  int nboundary;
  const char *ghost[4][2];    // [dim][0: from the -dim neighbour, 1: from the +dim neighbour] spin-projected half spinors
  int faceCB[4];
  int ghostNormOff[4];        // byte offset of the fp32 scales inside a ghost block (16-bit storage)
  // peer-store transport (p2p.h): every ghost word carries the flag of its exchange (GhostLL); waitCount[dir] is the value the
  // off-node hop in direction dir expects, waitTicks bounds the polling, a time-out lands in *errWord
  unsigned waitCount[8];
  unsigned long long waitTicks;
  int siteDelay, packPrio;   // peer-store launch: site blocks start siteDelay x 64 x 8 cycles late; pack waves raise their issue priority
  int *errWord;              // error record (p2p.h kP2pErrInts)
  unsigned exSeq; int exBuf; // exchange number / buffer of this launch, for that record
  // peer-store transport: the first packBlocks blocks of the interior launch pack the faces (pack_body) while the rest of
  // the grid does the interior stencil — one launch, the faces leave at time zero and travel during the interior pass
  int packBlocks, packChunk;
  // folded packing (small grids): the first packFoldBlocks blocks of the grid pack packShare face sites each, inside their site threads
  int packShare, packFoldBlocks;
  int edgeFirst;   // plane-tiled order of a partitioned launch: boundary planes first in every XCD (dslash_kernel)
  PackArg<real> pack;
  unsigned long long *timeline;   // QUDA_AMD_TIMELINE=1: per-block wall_clock64 stamps (measurement aid), else nullptr
};

// ---- spin projection / reconstruction in the chiral basis; s = +1 selects projector[2 mu], -1 projector[2 mu + 1]
// of the reference table (tests/wilson_dslash_reference.cpp:21-70) ----
// fp32: one packed multiply-add per projected colour component (6 per hop instead of 12), pk:: of device_io.h
template <int MU> __device__ __forceinline__ void spin_project(float *h, const float *p, float s) {
  const pkf2 S = {s, s};
#pragma unroll
  for (int c = 0; c < 3; c++) {
    const pkf2 p0 = pk::ld(p, 2 * c), p1 = pk::ld(p, 6 + 2 * c), p2 = pk::ld(p, 12 + 2 * c), p3 = pk::ld(p, 18 + 2 * c);
    pkf2 h0, h1;
    if (MU == 0) { h0 = pk::aixmy(S, p3, p0); h1 = pk::aixmy(S, p2, p1); }        // h0 = p0 - s i p3, h1 = p1 - s i p2
    else if (MU == 1) { h0 = pk::axpy(S, p3, p0); h1 = pk::axmy(S, p2, p1); }     // h0 = p0 + s p3,   h1 = p1 - s p2
    else if (MU == 2) { h0 = pk::aixmy(S, p2, p0); h1 = pk::aixpy(S, p3, p1); }   // h0 = p0 - s i p2, h1 = p1 + s i p3
    else { h0 = pk::axmy(S, p2, p0); h1 = pk::axmy(S, p3, p1); }                  // h0 = p0 - s p2,   h1 = p1 - s p3
    pk::st(h, 2 * c, h0); pk::st(h, 6 + 2 * c, h1);
  }
}
template <int MU, typename real> __device__ __forceinline__ void spin_project(real *h, const real *p, real s) {
#pragma unroll
  for (int c = 0; c < 3; c++) {
    const real p0r = p[0 + 2 * c], p0i = p[1 + 2 * c], p1r = p[6 + 2 * c], p1i = p[7 + 2 * c];
    const real p2r = p[12 + 2 * c], p2i = p[13 + 2 * c], p3r = p[18 + 2 * c], p3i = p[19 + 2 * c];
    if (MU == 0) {  // h0 = p0 - s i p3, h1 = p1 - s i p2
      h[2 * c] = p0r + s * p3i; h[2 * c + 1] = p0i - s * p3r;
      h[6 + 2 * c] = p1r + s * p2i; h[7 + 2 * c] = p1i - s * p2r;
    } else if (MU == 1) {  // h0 = p0 + s p3, h1 = p1 - s p2
      h[2 * c] = p0r + s * p3r; h[2 * c + 1] = p0i + s * p3i;
      h[6 + 2 * c] = p1r - s * p2r; h[7 + 2 * c] = p1i - s * p2i;
    } else if (MU == 2) {  // h0 = p0 - s i p2, h1 = p1 + s i p3
      h[2 * c] = p0r + s * p2i; h[2 * c + 1] = p0i - s * p2r;
      h[6 + 2 * c] = p1r - s * p3i; h[7 + 2 * c] = p1i + s * p3r;
    } else {  // h0 = p0 - s p2, h1 = p1 - s p3
      h[2 * c] = p0r - s * p2r; h[2 * c + 1] = p0i - s * p2i;
      h[6 + 2 * c] = p1r - s * p3r; h[7 + 2 * c] = p1i - s * p3i;
    }
  }
}

// fp32: acc += w (reconstruction of g) in 12 packed instructions (w = 1: plain reconstruction)
template <int MU> __device__ __forceinline__ void spin_reconstruct_pk(float *acc, const float *g, float s, float w) {
  const pkf2 W = {w, w}, SW = {s * w, s * w};
#pragma unroll
  for (int c = 0; c < 3; c++) {
    const pkf2 g0 = pk::ld(g, 2 * c), g1 = pk::ld(g, 6 + 2 * c);
    pk::st(acc, 2 * c, pk::axpy(W, g0, pk::ld(acc, 2 * c)));
    pk::st(acc, 6 + 2 * c, pk::axpy(W, g1, pk::ld(acc, 6 + 2 * c)));
    pkf2 a2 = pk::ld(acc, 12 + 2 * c), a3 = pk::ld(acc, 18 + 2 * c);
    if (MU == 0) { a2 = pk::aixpy(SW, g1, a2); a3 = pk::aixpy(SW, g0, a3); }        // r2 = s i g1, r3 = s i g0
    else if (MU == 1) { a2 = pk::axmy(SW, g1, a2); a3 = pk::axpy(SW, g0, a3); }     // r2 = -s g1,  r3 = s g0
    else if (MU == 2) { a2 = pk::aixpy(SW, g0, a2); a3 = pk::aixmy(SW, g1, a3); }   // r2 = s i g0, r3 = -s i g1
    else { a2 = pk::axmy(SW, g0, a2); a3 = pk::axmy(SW, g1, a3); }                  // r2 = -s g0,  r3 = -s g1
    pk::st(acc, 12 + 2 * c, a2); pk::st(acc, 18 + 2 * c, a3);
  }
}
template <int MU> __device__ __forceinline__ void spin_reconstruct(float *acc, const float *g, float s) { spin_reconstruct_pk<MU>(acc, g, s, 1.0f); }
template <int MU, typename real> __device__ __forceinline__ void spin_reconstruct(real *acc, const real *g, real s) {
#pragma unroll
  for (int c = 0; c < 3; c++) {
    const real g0r = g[2 * c], g0i = g[2 * c + 1], g1r = g[6 + 2 * c], g1i = g[7 + 2 * c];
    acc[0 + 2 * c] += g0r; acc[1 + 2 * c] += g0i;
    acc[6 + 2 * c] += g1r; acc[7 + 2 * c] += g1i;
    if (MU == 0) {  // r2 = s i g1, r3 = s i g0
      acc[12 + 2 * c] -= s * g1i; acc[13 + 2 * c] += s * g1r;
      acc[18 + 2 * c] -= s * g0i; acc[19 + 2 * c] += s * g0r;
    } else if (MU == 1) {  // r2 = -s g1, r3 = s g0
      acc[12 + 2 * c] -= s * g1r; acc[13 + 2 * c] -= s * g1i;
      acc[18 + 2 * c] += s * g0r; acc[19 + 2 * c] += s * g0i;
    } else if (MU == 2) {  // r2 = s i g0, r3 = -s i g1
      acc[12 + 2 * c] -= s * g0i; acc[13 + 2 * c] += s * g0r;
      acc[18 + 2 * c] += s * g1i; acc[19 + 2 * c] -= s * g1r;
    } else {  // r2 = -s g0, r3 = -s g1
      acc[12 + 2 * c] -= s * g0r; acc[13 + 2 * c] -= s * g0i;
      acc[18 + 2 * c] -= s * g1r; acc[19 + 2 * c] -= s * g1i;
    }
  }
}

// acc += w (reconstruction of g): the same 24 instructions as above (12 adds become multiply-adds) — how the 16-bit kernels apply the
// scale of a neighbour's integers (site scale x link scale) for free, instead of to every one of the 24 + 18 converted operands
template <int MU> __device__ __forceinline__ void spin_reconstruct_scaled(float *acc, const float *g, float s, float w) { spin_reconstruct_pk<MU>(acc, g, s, w); }
template <int MU, typename real> __device__ __forceinline__ void spin_reconstruct_scaled(real *acc, const real *g, real s, real w) {
  const real sw = s * w;
#pragma unroll
  for (int c = 0; c < 3; c++) {
    const real g0r = g[2 * c], g0i = g[2 * c + 1], g1r = g[6 + 2 * c], g1i = g[7 + 2 * c];
    acc[0 + 2 * c] += w * g0r; acc[1 + 2 * c] += w * g0i;
    acc[6 + 2 * c] += w * g1r; acc[7 + 2 * c] += w * g1i;
    if (MU == 0) {
      acc[12 + 2 * c] -= sw * g1i; acc[13 + 2 * c] += sw * g1r;
      acc[18 + 2 * c] -= sw * g0i; acc[19 + 2 * c] += sw * g0r;
    } else if (MU == 1) {
      acc[12 + 2 * c] -= sw * g1r; acc[13 + 2 * c] -= sw * g1i;
      acc[18 + 2 * c] += sw * g0r; acc[19 + 2 * c] += sw * g0i;
    } else if (MU == 2) {
      acc[12 + 2 * c] -= sw * g0i; acc[13 + 2 * c] += sw * g0r;
      acc[18 + 2 * c] += sw * g1i; acc[19 + 2 * c] -= sw * g1r;
    } else {
      acc[12 + 2 * c] -= sw * g0r; acc[13 + 2 * c] -= sw * g0i;
      acc[18 + 2 * c] -= sw * g1r; acc[19 + 2 * c] -= sw * g1i;
    }
  }
}

// (1 + i a g5) in place; g5 = diag(+,+,-,-)
template <typename real> __device__ __forceinline__ void twist_inplace(real *p, real a) {
#pragma unroll
  for (int k = 0; k < 6; k++) {
    const real r = p[2 * k], i = p[2 * k + 1];
    p[2 * k] = r - a * i;
    p[2 * k + 1] = i + a * r;
  }
#pragma unroll
  for (int k = 6; k < 12; k++) {
    const real r = p[2 * k], i = p[2 * k + 1];
    p[2 * k] = r + a * i;
    p[2 * k + 1] = i - a * r;
  }
}

// ---- flag-in-data ghost transport (peer stores) ----
// A face site travels as NV 16-byte vectors {w0, flag, w1, flag}: two 4-byte payload words, each next to the exchange's flag
// (the cumulative use count of the (dimension, buffer) zone, never 0), plane-major [vector][faceCB].  Every 8-byte half is
// self-validating, so the receiver needs neither a counter nor any ordering between stores: it polls the very words it is
// about to use, and the sender just stores and leaves — no acknowledgement wait, no barrier, no atomic (those put the
// slowest pack block, 18 us under a saturated memory system, on the critical path of a 10-20 us kernel; the same idea as the
// LL protocol of the collective libraries).  Costs 2x the face bytes, which are a few hundred KB.
#ifndef QA_LL_STORE_AUX
#define QA_LL_STORE_AUX 17   // sc0 sc1: system scope, write-through
#endif
#ifndef QA_P2P_FOLD
#define QA_P2P_FOLD 0   // 1 compiles the folded face packing (inside the site threads, stencil_site) into the peer-store kernels; it then runs with
                        // QUDA_AMD_P2P_FOLD=1 / tune key "p2p_fold".  Measured slower than pack blocks in round 2 (27.3 against 24.0 us), so it is
                        // compiled OUT by default: the kernels lose a ballot, a branch and the registers of a second basic-block structure
#endif
template <typename T> struct GhostLL;
template <> struct GhostLL<double> {
  static constexpr int NV = 12, NA = 8;
  static __device__ __forceinline__ void encode(unsigned *w, const double *h) {
#pragma unroll
    for (int k = 0; k < 12; k++) { const unsigned long long b = __builtin_bit_cast(unsigned long long, h[k]); w[2 * k] = (unsigned)b; w[2 * k + 1] = (unsigned)(b >> 32); }
  }
  static __device__ __forceinline__ void decode(double *h, const unsigned *w) {
#pragma unroll
    for (int k = 0; k < 12; k++) h[k] = __builtin_bit_cast(double, (unsigned long long)w[2 * k] | ((unsigned long long)w[2 * k + 1] << 32));
  }
};
template <> struct GhostLL<float> {
  static constexpr int NV = 6, NA = 4;
  static __device__ __forceinline__ void encode(unsigned *w, const float *h) {
#pragma unroll
    for (int k = 0; k < 12; k++) w[k] = __builtin_bit_cast(unsigned, h[k]);
  }
  static __device__ __forceinline__ void decode(float *h, const unsigned *w) {
#pragma unroll
    for (int k = 0; k < 12; k++) h[k] = __builtin_bit_cast(float, w[k]);
  }
};
template <> struct GhostLL<short> {   // 12 int16 + the fp32 scale of the site (same quantisation as Planar<short>::store)
  static constexpr int NV = 4, NA = 3;
  static __device__ __forceinline__ void encode(unsigned *w, const float *h) {
    float m = 0.f;
#pragma unroll
    for (int k = 0; k < 12; k++) m = fmaxf(m, fabsf(h[k]));
    const float sc = m > 0.f ? kShortMax / m : 0.f;
#pragma unroll
    for (int k = 0; k < 6; k++)
      w[k] = (unsigned)(unsigned short)Planar<short, 12>::q16(h[2 * k] * sc) | ((unsigned)(unsigned short)Planar<short, 12>::q16(h[2 * k + 1] * sc) << 16);
    w[6] = __builtin_bit_cast(unsigned, m);
    w[7] = 0u;
  }
  static __device__ __forceinline__ void decode(float *h, const unsigned *w) {
    const float sc = __builtin_bit_cast(float, w[6]) * kShortInv;
#pragma unroll
    for (int k = 0; k < 6; k++) { h[2 * k] = (float)(short)(w[k] & 0xffffu) * sc; h[2 * k + 1] = (float)(short)(w[k] >> 16) * sc; }
  }
};
template <typename T> __device__ __forceinline__ __amdgpu_buffer_rsrc_t ghost_ll_rsrc(const void *base, int faceCB) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)((unsigned)faceCB * (unsigned)(GhostLL<T>::NV * 16)), 0x00020000);
}
// sender: system-scope write-through stores into the neighbour's zone
template <typename T, typename real> __device__ __forceinline__ void ghost_ll_store(const real *h, void *zone, int faceCB, int f, unsigned flag) {
  constexpr int NV = GhostLL<T>::NV;
  unsigned w[2 * NV];
  GhostLL<T>::encode(w, h);
  const __amdgpu_buffer_rsrc_t rs = ghost_ll_rsrc<T>(zone, faceCB);
#pragma unroll
  for (int v = 0; v < NV; v++) {
    u32x4_t q; q.x = w[2 * v]; q.y = flag; q.z = w[2 * v + 1]; q.w = flag;
    __builtin_amdgcn_raw_buffer_store_b128(q, rs, f * 16, v * faceCB * 16, QA_LL_STORE_AUX);
  }
}
// receiver: poll the site's own vectors (system-scope loads) until every half carries this exchange's flag; bounded by `ticks`
// ---- the compact wire format: self-validating 16-byte atoms (QUDA_AMD_HALO_FORMAT=atom, tune key "halo_format" = 1) ----
// The flag-in-data vectors above spend half of every byte on the wire on flags (786 KB per fp64 face of the 8-GPU split of 32^3 x 64,
// two such faces sharing one xGMI link where a dimension has only two ranks).  Here a face site travels as NA atoms of 16 bytes =
// three 32-bit payload words + the exchange's 32-bit flag: fp64 24 words = 8 atoms = ONE 128-byte line per site, fp32 4 atoms = 64 bytes,
// 16-bit storage 6 packed words + the scale = 3 atoms = 48 bytes — 128 / 64 / 48 bytes per face site against 192 / 96 / 64.
// Plane-major [atom][faceCB].  Every atom is written by ONE 16-byte store of ONE lane and validates itself: the format assumes nothing
// about the order in which different stores, or different lanes of one store, become visible across xGMI — only that a naturally aligned
// 16-byte store of a lane is not torn, the granularity the collective libraries' 128-byte protocol builds on as well.  (Round 3 had
// 32-byte sectors written by two adjacent lanes with the flag in the second half: one assumption more, and an LDS transpose in the pack
// blocks to arrange it; replaced.)  The sender is pack_emit as for flag-in-data — no LDS, and the receiver needs 8 instead of 12
// sixteen-byte loads per fp64 site.
template <typename T> __device__ __forceinline__ __amdgpu_buffer_rsrc_t ghost_atom_rsrc(const void *base, int faceCB) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)((unsigned)faceCB * (unsigned)(GhostLL<T>::NA * 16)), 0x00020000);
}
template <typename T, typename real> __device__ __forceinline__ void ghost_atom_store(const real *h, void *zone, int faceCB, int f, unsigned flag) {
  constexpr int NV = GhostLL<T>::NV, NA = GhostLL<T>::NA;
  unsigned w[3 * NA > 2 * NV ? 3 * NA : 2 * NV];
  GhostLL<T>::encode(w, h);
#pragma unroll
  for (int k = 2 * NV; k < 3 * NA; k++) w[k] = 0u;
  const __amdgpu_buffer_rsrc_t rs = ghost_atom_rsrc<T>(zone, faceCB);
#pragma unroll
  for (int v = 0; v < NA; v++) {
    u32x4_t q; q.x = w[3 * v]; q.y = w[3 * v + 1]; q.z = w[3 * v + 2]; q.w = flag;
    __builtin_amdgcn_raw_buffer_store_b128(q, rs, f * 16, v * faceCB * 16, QA_LL_STORE_AUX);
  }
}
// one look at the atoms of face site f; false: some atom does not carry this exchange's flag yet (sy = the flag seen, sv = its atom)
template <typename T> __device__ __forceinline__ bool ghost_atom_read(unsigned *w, const __amdgpu_buffer_rsrc_t &rs, int faceCB, int f, unsigned flag, unsigned &sy, unsigned &sw, int &sv) {
  constexpr int NV = GhostLL<T>::NV, NA = GhostLL<T>::NA;
  bool ok = true;
#pragma unroll
  for (int v = 0; v < NA; v++) {
    const u32x4_t q = __builtin_amdgcn_raw_buffer_load_b128(rs, f * 16, v * faceCB * 16, 17);
    if (3 * v < 2 * NV) w[3 * v] = q.x;
    if (3 * v + 1 < 2 * NV) w[3 * v + 1] = q.y;
    if (3 * v + 2 < 2 * NV) w[3 * v + 2] = q.z;
    const bool good = q.w == flag;
    if (!good && ok) { sy = q.w; sw = q.w; sv = v; }
    ok = ok && good;
  }
  return ok;
}
template <typename T, typename real>
__device__ __forceinline__ void ghost_ll_load(real *h, const void *zone, int faceCB, int f, unsigned flag, unsigned long long ticks, int *errWord, int code, unsigned exSeq, int exBuf,
                                              int fmt) {
  constexpr int NV = GhostLL<T>::NV;
  unsigned w[2 * NV];
  if (fmt) {
    const __amdgpu_buffer_rsrc_t rs = ghost_atom_rsrc<T>(zone, faceCB);
    unsigned long long t0 = 0;
    for (;;) {
      unsigned sy = flag, sw = flag; int sv = -1;
      if (ghost_atom_read<T>(w, rs, faceCB, f, flag, sy, sw, sv)) break;
      if (!t0) t0 = wall_clock64();
      else if (__hip_atomic_load(errWord, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
      else if (wall_clock64() - t0 > ticks) {
        if (atomicCAS(errWord, 0, code) == 0) { errWord[1] = f; errWord[2] = (int)flag; errWord[3] = (int)sy; errWord[4] = (int)sw; errWord[5] = (int)exSeq; errWord[6] = exBuf; errWord[7] = sv; }
        break;
      }
      __builtin_amdgcn_s_sleep(2);
    }
    GhostLL<T>::decode(h, w);
    return;
  }
  const __amdgpu_buffer_rsrc_t rs = ghost_ll_rsrc<T>(zone, faceCB);
  unsigned long long t0 = 0;
  for (;;) {
    bool ok = true;
#pragma unroll
    for (int v = 0; v < NV; v++) {
      const u32x4_t q = __builtin_amdgcn_raw_buffer_load_b128(rs, f * 16, v * faceCB * 16, 17);
      w[2 * v] = q.x; w[2 * v + 1] = q.z;
      ok = ok && q.y == flag && q.w == flag;
    }
    if (ok) break;
    if (!t0) t0 = wall_clock64();
    else if (__hip_atomic_load(errWord, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;   // another wait has given up already
    else if (wall_clock64() - t0 > ticks) {
      // give up: the host reports it (p2pCheck).  The first wait to get here claims the record and says what it was waiting for and
      // what it last saw there (one more look at the vectors: nothing of this is kept live in the polling loop)
      if (atomicCAS(errWord, 0, code) == 0) {
        unsigned sy = flag, sw = flag; int sv = -1;
        for (int v = NV - 1; v >= 0; v--) {
          const u32x4_t q = __builtin_amdgcn_raw_buffer_load_b128(rs, f * 16, v * faceCB * 16, 17);
          if (q.y != flag || q.w != flag) { sy = q.y; sw = q.w; sv = v; }
        }
        errWord[1] = f; errWord[2] = (int)flag; errWord[3] = (int)sy; errWord[4] = (int)sw; errWord[5] = (int)exSeq; errWord[6] = exBuf; errWord[7] = sv;
      }
      break;
    }
    __builtin_amdgcn_s_sleep(2);
  }
  GhostLL<T>::decode(h, w);
}
// One hop = load (neighbour spinor + this site's pre-daggered link) then project / multiply / reconstruct.
// The two phases are separate so the kernel can software-pipeline them: loads of direction d+1 are issued
// before the arithmetic of direction d (register double-buffering), fenced with sched_barrier so hipcc neither
// hoists all 8 directions' loads to the top (fp64: 512 registers + scratch spills, 1 wave/SIMD) nor serialises them.
// GAUX: cache policy of the link stream (0 default, 2 = nt: read-once data that should not displace the spinor
// working set from L2 / Infinity Cache).  QA_GAUX / QA_GAUX16 / QA_PAUX: build-time overrides for A/B timing (tools/build_variant.sh).
#ifndef QA_GAUX
#define QA_GAUX 2
#endif
#ifndef QA_GAUX16
#define QA_GAUX16 0
#endif
#ifndef QA_PAUX
#define QA_PAUX 0
#endif
// GH: 0 = every neighbour is local; 1 = off-node neighbours are read from the ghost zone (exterior pass); 2 = off-node hops are
// skipped here and done by ghost_hop() once the faces have arrived (single-launch peer-store path)
// hop_load only REQUESTS (raw register images, device_io.h RawBlock): the 16-bit -> fp32 conversion and the third row of a
// 12-real link are arithmetic on the loaded data and belong to hop_compute, behind the fence — inside the request phase they made
// the wave wait for the data of hop d + 1 before it started on hop d (16-bit and recon-12 kernels ran without any overlap).
template <typename T, int R, int DIR, int GAUX, int GH = 0, typename real>
__device__ __forceinline__ void hop_load(RawBlock<T, 24> &psi, typename Link<T, R>::Raw &U, const DslashArg<real> &arg, int idx, int nbr, bool off_node = false, int face = 0) {
  constexpr int MU = DIR / 2;
  // GH == 2: the load is issued for every lane (the periodic wrap makes the index valid) and the contribution of an off-node
  // lane is zeroed in hop_compute: a per-lane branch here splits the 8-hop pipeline into many basic blocks and costs the
  // fp64 kernel its second wave per SIMD (256 + 8 registers against 231 for the branch-free loop)
  if (GH == 1 && off_node) {
    // the neighbour lives on another rank: its (pre-twisted,) spin-projected half spinor was packed there with the same
    // projector sign; it goes into the first 12 slots of the buffer.  Issued here, with the other loads of the hop, so it
    // does not cost a dependent round trip in the compute phase.  sc0 sc1 (system-scope) loads: the zone may have been
    // written by another GPU while this kernel was running.
    const char *gb = arg.ghost[MU][(DIR & 1) ? 0 : 1];
    psi.template load12<17>(gb, arg.faceCB[MU], face, reinterpret_cast<const float *>(gb + arg.ghostNormOff[MU]), face);
  } else {
    psi.template load<QA_PAUX>(arg.in, arg.sp_stride, nbr, arg.inNorm, nbr);
  }
  Link<T, R>::template request<GAUX>(U, arg.gauge + (size_t)DIR * arg.link_bytes, arg.g_stride, idx);
}
// project / multiply / reconstruct on unpacked operands
template <int DIR, bool PRETWIST, int GH, bool SCALED = false, typename real>
__device__ __forceinline__ void hop_arith(real *acc, real *psi, const real *U, const DslashArg<real> &arg, bool off_node = false, real w = 1) {
  constexpr int MU = DIR / 2;
  real h[12], g[12];
  const real s = (DIR & 1) ? -arg.sfwd : arg.sfwd;
  if (GH == 1 && off_node) {
#pragma unroll
    for (int k = 0; k < 12; k++) h[k] = psi[k];
  } else {
    if (PRETWIST) twist_inplace(psi, arg.a);  // QUDA_DEG_TWIST_INV_DSLASH: A^-1 applied to the neighbour before the hop
    spin_project<MU>(h, psi, s);
    if (GH == 2) {   // off-node hop: added later from the ghost zone (ghost_hop); select, not branch
#pragma unroll
      for (int k = 0; k < 12; k++) h[k] = off_node ? (real)0 : h[k];
    }
  }
  su3_mv(g, U, h);
  su3_mv(g + 6, U, h + 6);
  if (SCALED) spin_reconstruct_scaled<MU>(acc, g, s, w);
  else spin_reconstruct<MU>(acc, g, s);
}
template <typename T, int R, int DIR, bool PRETWIST, int GH, typename real>
__device__ __forceinline__ void hop_compute(real *acc, const RawBlock<T, 24> &psiRaw, const typename Link<T, R>::Raw &URaw, const DslashArg<real> &arg, real sign, bool off_node = false,
                                            int face = 0) {
  real psi[24], U[18];
  if constexpr (sizeof(T) == 2) {
    // 16-bit: the hop is linear in the neighbour and in the link, so the integers go through it as they are and the product of
    // the two scales is applied inside the reconstruction (conversion: one v_cvt per operand instead of v_cvt + v_mul)
    psiRaw.unpack_unscaled(psi);
    if constexpr (R == 8) {
      // 8-real links: the reconstruction is not linear in the stored values, so the link comes back in its own units and only the
      // neighbour's integers and scale go through the hop
      Link<T, R>::finish(U, URaw, sign);
      hop_arith<DIR, PRETWIST, GH, true>(acc, psi, U, arg, off_node, psiRaw.scale());
    } else {
      URaw.unpack_unscaled(U);
      if (R == 12) Link<T, R>::third_row(U, sign * kShortInv);   // row 3 = conj(row 1 x row 2): back to the units of the stored rows
      hop_arith<DIR, PRETWIST, GH, true>(acc, psi, U, arg, off_node, psiRaw.scale() * kShortInv);
    }
  } else {
    psiRaw.unpack(psi);
    Link<T, R>::finish(U, URaw, sign);
    hop_arith<DIR, PRETWIST, GH>(acc, psi, U, arg, off_node);
  }
}

// one off-node hop from the ghost zone (half spinor packed by the neighbour + this site's link), for the lanes that need it
template <typename T, int R, int DIR, int GAUX, typename real>
__device__ __forceinline__ void ghost_hop(real *acc, const DslashArg<real> &arg, int idx, bool off_node, int face, real sign) {
  constexpr int MU = DIR / 2;
  if (!off_node) return;
  real h[12], g[12], U[18];
  Link<T, R>::template load<GAUX>(U, arg.gauge + (size_t)DIR * arg.link_bytes, arg.g_stride, idx, sign);
  ghost_ll_load<T>(h, arg.ghost[MU][(DIR & 1) ? 0 : 1], arg.faceCB[MU], face, arg.waitCount[DIR], arg.waitTicks, arg.errWord, 1 + DIR, arg.exSeq, arg.exBuf, arg.pack.llFormat);
  const real s = (DIR & 1) ? -arg.sfwd : arg.sfwd;
  su3_mv(g, U, h);
  su3_mv(g + 6, U, h + 6);
  spin_reconstruct<MU>(acc, g, s);
}

// o = (chiral block held as a raw image) v.  16-bit: the block's integers go through the product unscaled and the 12 results take
// the block's scale (12 multiplications instead of 36)
template <typename T, typename real> __device__ __forceinline__ void clover_mv_raw(real *o, const RawBlock<T, 36> &raw, real *C, const real *v) {
  if constexpr (sizeof(T) == 2) {
    raw.unpack_unscaled(C);
    clover_block_mv(o, C, v);
    const real w = raw.scale();
#pragma unroll
    for (int k = 0; k < 12; k++) o[k] *= w;
  } else {
    raw.unpack(C);
    clover_block_mv(o, C, v);
  }
}

// fused epilogue of the stencil: twist / inverse twist / xpay / clover-twist (+ inverse), then the store of the site
template <typename T, int VARIANT, int GAUX, int SAUX, typename real>
__device__ __forceinline__ void dslash_epilogue(real *acc, const DslashArg<real> &arg, int idx) {
  constexpr bool CLOVER = VARIANT == 2;
  real xs[24];
  // (clover-twist inverse: x is only added at the very end; requested there, it does not sit in 24 registers through the four block products)
  RawBlock<T, 24> xsRaw;
  if (arg.xpay && !(CLOVER && arg.mode == DSLASH_CLOVER_TWIST_INV)) {
    xsRaw.load(arg.x, arg.sp_stride, idx, arg.xNorm, idx);
    if (!CLOVER) xsRaw.unpack(xs);
  }

  if (arg.mode == DSLASH_PLAIN) {
    if (arg.xpay) {
#pragma unroll
      for (int k = 0; k < 24; k++) acc[k] = xs[k] + arg.k * acc[k];
    }
  } else if (arg.mode == DSLASH_TWIST_INV || arg.mode == DSLASH_TWIST_INV_DSLASH) {
    if (arg.mode == DSLASH_TWIST_INV) twist_inplace(acc, arg.a);
    if (arg.xpay) {
#pragma unroll
      for (int k = 0; k < 24; k++) acc[k] = xs[k] + arg.b * acc[k];
    } else {
#pragma unroll
      for (int k = 0; k < 24; k++) acc[k] *= arg.b;
    }
  } else if (arg.mode == DSLASH_TWIST_XPAY) {
    twist_inplace(xs, arg.a);
#pragma unroll
    for (int k = 0; k < 24; k++) acc[k] = arg.k * acc[k] + xs[k];
  } else if (CLOVER) {
    real C[36], tmp[24];
    if (arg.mode == DSLASH_CLOVER_TWIST_INV) {
      // tmp = (A + i a g5) acc ; res = Ainv tmp, chirality by chirality (the two 6 x 6 blocks do not mix).  Four blocks of 36 reals
      // are read; left to itself the compiler requests all four up front and converts them at once (144 live registers on top of
      // acc: 176 VGPRs for the 16-bit kernel, 198 for fp32 — two waves per SIMD where the plain twisted-mass kernels run four).
      // Two raw images (RawBlock: 19 registers each in 16-bit), the next block requested before the product with the current
      // one, fenced like the hop pipeline.
      // (fp64: two images are 144 registers; its blocks are requested and used one after the other)
      if constexpr (sizeof(T) == 8) {
        const real sa[2] = {arg.a, -arg.a};
#pragma unroll
        for (int chi = 0; chi < 2; chi++) {
          real *t = tmp + 12 * chi, *v = acc + 12 * chi;
          Planar<T, 36>::template load<GAUX>(C, (const char *)arg.clA + (size_t)chi * 36 * sizeof(T) * arg.cl_stride, arg.cl_stride, idx, arg.clAn, chi * arg.cl_stride + idx);
          clover_block_mv(t, C, v);
#pragma unroll
          for (int k = 0; k < 6; k++) { t[2 * k] -= sa[chi] * v[2 * k + 1]; t[2 * k + 1] += sa[chi] * v[2 * k]; }
        }
#pragma unroll
        for (int chi = 0; chi < 2; chi++) {
          Planar<T, 36>::template load<GAUX>(C, (const char *)arg.clAinv + (size_t)chi * 36 * sizeof(T) * arg.cl_stride, arg.cl_stride, idx, arg.clAinvN, chi * arg.cl_stride + idx);
          clover_block_mv(acc + 12 * chi, C, tmp + 12 * chi);
        }
      } else {
        // one raw image at a time, each block requested after the product with the previous one (fenced): four exposed latencies,
        // but the kernel keeps the register count of the plain twisted-mass stencil and other waves cover them
        RawBlock<T, 36> Ca;
        const real sa[2] = {arg.a, -arg.a};
#pragma unroll
        for (int chi = 0; chi < 2; chi++) {
          real *t = tmp + 12 * chi, *v = acc + 12 * chi;
          Ca.template load<GAUX>((const char *)arg.clA + (size_t)chi * 36 * sizeof(T) * arg.cl_stride, arg.cl_stride, idx, arg.clAn, chi * arg.cl_stride + idx);
          __builtin_amdgcn_sched_barrier(0);
          clover_mv_raw<T>(t, Ca, C, v);
#pragma unroll
          for (int k = 0; k < 6; k++) { t[2 * k] -= sa[chi] * v[2 * k + 1]; t[2 * k + 1] += sa[chi] * v[2 * k]; }
          __builtin_amdgcn_sched_barrier(0);
          Ca.template load<GAUX>((const char *)arg.clAinv + (size_t)chi * 36 * sizeof(T) * arg.cl_stride, arg.cl_stride, idx, arg.clAinvN, chi * arg.cl_stride + idx);
          __builtin_amdgcn_sched_barrier(0);
          clover_mv_raw<T>(v, Ca, C, t);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (arg.xpay) {
        Planar<T, 24>::load(xs, arg.x, arg.sp_stride, idx, arg.xNorm, idx);
#pragma unroll
        for (int k = 0; k < 24; k++) acc[k] = xs[k] + arg.k * acc[k];
      }
    } else {  // DSLASH_CLOVER_TWIST_XPAY: out = k acc + (A + i a g5) x, chirality by chirality
      RawBlock<T, 36> Ca;
      const real sa[2] = {arg.a, -arg.a};
#pragma unroll
      for (int chi = 0; chi < 2; chi++) {
        real xv[24];
        xsRaw.unpack(xv);   // only the 12 values of this chirality survive: x stays a raw image (13 registers in 16-bit) until here
        real *t = tmp + 12 * chi, *v = xv + 12 * chi, *o = acc + 12 * chi;
        Ca.load((const char *)arg.clA + (size_t)chi * 36 * sizeof(T) * arg.cl_stride, arg.cl_stride, idx, arg.clAn, chi * arg.cl_stride + idx);
        __builtin_amdgcn_sched_barrier(0);
        clover_mv_raw<T>(t, Ca, C, v);
#pragma unroll
        for (int k = 0; k < 6; k++) { t[2 * k] -= sa[chi] * v[2 * k + 1]; t[2 * k + 1] += sa[chi] * v[2 * k]; }
#pragma unroll
        for (int k = 0; k < 12; k++) o[k] = arg.k * o[k] + t[k];
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  Planar<T, 24>::template store<SAUX>(acc, arg.out, arg.sp_stride, idx, arg.outNorm, idx);
}

// ---- face packing (reference packFaceWilsonKernel / packTwistedFaceWilsonKernel, lib/dslash_pack.cu:272, :610) ----
// P2P: the send pointers are peer-mapped ghost zones — flag-in-data vectors, system-scope write-through stores (GhostLL)
// Block `bid` packs the face sites [bid * chunk, (bid + 1) * chunk) of the concatenated (dim, dir) ranges, chunk <= blockDim.
// The work of one face site in two steps, so that a caller can put other loads between the load and the stores:
// pack_locate: face item `tid` of the concatenated (dim, dir) ranges -> (slot = 2 dim + to_fwd, face index f, checkerboard index)
template <typename real> __device__ __forceinline__ int pack_locate(const PackArg<real> &arg, int tid, int &slot, int &f) {
  slot = 0;
#pragma unroll
  for (int k = 1; k < 8; k++) slot += tid >= arg.start[k];
  const int d = slot >> 1, to_fwd = slot & 1;
  f = tid - arg.start[slot];
  // other three coordinates from the face index (lexicographic, halved), then the site's full index
  int c[4], L[3], o[3], n = 0;
  for (int k = 0; k < 4; k++) if (k != d) { L[n] = arg.X[k]; o[n] = k; n++; }
  int l = 2 * f;
  const int c0 = l % L[0]; l /= L[0];
  const int c1 = l % L[1]; const int c2 = l / L[1];
  c[d] = to_fwd ? arg.X[d] - 1 : 0;
  c[o[0]] = c0; c[o[1]] = c1; c[o[2]] = c2;
  c[o[0]] += (arg.parity_in + c[0] + c[1] + c[2] + c[3]) & 1;  // pick the site of the pair that has the input parity
  return (((c[3] * arg.X[2] + c[2]) * arg.X[1] + c[1]) * arg.X[0] + c[0]) >> 1;
}
// pack_emit: (pre-twist,) project for the receiver's hop and store — flag-in-data vectors into the neighbour's zone (P2P) or planes
// of the send buffer
template <typename T, bool PRETWIST, bool P2P, typename real> __device__ __forceinline__ void pack_emit(const PackArg<real> &arg, real *psi, int slot, int f) {
  const int d = slot >> 1, to_fwd = slot & 1;
  real h[12];
  if (PRETWIST) twist_inplace(psi, arg.a);
  // the receiver uses this face for its hop in direction -d (if we send forward) / +d (if we send backward)
  const real s = to_fwd ? -arg.sfwd : arg.sfwd;
  switch (d) {
    case 0: spin_project<0>(h, psi, s); break;
    case 1: spin_project<1>(h, psi, s); break;
    case 2: spin_project<2>(h, psi, s); break;
    default: spin_project<3>(h, psi, s); break;
  }
  char *sb = arg.send[d][to_fwd];
  if (P2P) {   // straight into the neighbour's zone, flag in the data
    if (arg.llFormat) ghost_atom_store<T>(h, sb, arg.faceCB[d], f, arg.llFlag[d]);
    else ghost_ll_store<T>(h, sb, arg.faceCB[d], f, arg.llFlag[d]);
  }
  else Planar<T, 12>::store(h, sb, arg.faceCB[d], f, reinterpret_cast<float *>(sb + arg.normOff[d]), f);
}
template <typename T, bool PRETWIST, bool P2P, typename real> __device__ __forceinline__ void pack_body(const PackArg<real> &arg, int bid, int chunk) {
  const int tid = (int)threadIdx.x < chunk ? bid * chunk + (int)threadIdx.x : arg.start[8];
  if (P2P && arg.timeline && threadIdx.x == 0) {
    arg.timeline[bid] = wall_clock64();
    // where the block runs: HW_ID (cu_id 11:8, sh_id 12, se_id 15:13) and XCC_ID, for the placement statistics of the timeline
    arg.timeline[3072 + bid] = 0x100000000ull | ((unsigned long long)(__builtin_amdgcn_s_getreg(20 | (31 << 11)) & 0xf) << 16) | (__builtin_amdgcn_s_getreg(4 | (31 << 11)) & 0xff00u);
  }
  if (tid < arg.start[8]) {
    int slot, f;
    const int idx = pack_locate(arg, tid, slot, f);
    real psi[24];
    Planar<T, 24>::load(psi, arg.in, arg.sp_stride, idx, arg.inNorm, idx);
    if (P2P && arg.timeline) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (threadIdx.x == 0) arg.timeline[13312 + bid] = wall_clock64();
    }
    pack_emit<T, PRETWIST, P2P>(arg, psi, slot, f);
  }
  if (P2P && arg.timeline && threadIdx.x == 0) arg.timeline[1024 + bid] = wall_clock64();
}

// VARIANT: 0 = Wilson / twist epilogues, 1 = twist applied to the neighbours first (TWIST_INV_DSLASH), 2 = clover epilogues.
// Compile-time so the 8-hop pipeline below is one straight-line basic block (a wave-uniform runtime branch per hop
// made hipcc split it into ~80 blocks and shuttle the double buffers through AGPRs).
// All 8 hops + epilogue of one site.  KT: 0 = every neighbour is local (periodic wrap inside this rank); 1 = interior pass of a
// grid-decomposed lattice (sites that touch a partitioned boundary are skipped); 2 = exterior pass (boundary site, off-node
// neighbours from the ghost zone); 3 = single-launch peer-store path: local hops first, then the off-node hops once the
// neighbours' faces have arrived.
template <typename T, int R, int VARIANT, int GAUX, int KT, int SAUX, typename real>
__device__ __forceinline__ void stencil_site(const DslashArg<real> &arg, const int idx) {
  // checkerboard index -> coordinates (tests/test_util.cpp:419-443)
  const uint32_t za = arg.dXh.div((uint32_t)idx);
  const int xh = idx - (int)za * arg.Xh;
  const uint32_t zb = arg.dY.div(za);
  const int y = (int)za - (int)zb * arg.Y;
  const int t = (int)arg.dZ.div(zb);
  const int z = (int)zb - t * arg.Z;
  const int xodd = (y + z + t + arg.parity) & 1;
  const int xf = 2 * xh + xodd;  // full x coordinate
  // which hops leave this rank (KT != 0 only)
  const bool gx = (arg.commMask & 1) != 0, gy = (arg.commMask & 2) != 0, gz = (arg.commMask & 4) != 0, gt = (arg.commMask & 8) != 0;
  const bool o_xp = gx && xf == 2 * arg.Xh - 1, o_xm = gx && xf == 0, o_yp = gy && y == arg.Y - 1, o_ym = gy && y == 0;
  const bool o_zp = gz && z == arg.Z - 1, o_zm = gz && z == 0, o_tp = gt && t == arg.T - 1, o_tm = gt && t == 0;
  if (KT == 1 && (o_xp || o_xm || o_yp || o_ym || o_zp || o_zm || o_tp || o_tm)) return;
  // face (ghost-zone) indices: lexicographic over the three other coordinates, halved (reference tests/dslash_util.h:291-394)
  const int X1 = 2 * arg.Xh;
  const int f_x = ((t * arg.Z + z) * arg.Y + y) >> 1, f_y = ((t * arg.Z + z) * X1 + xf) >> 1;
  const int f_z = ((t * arg.Y + y) * X1 + xf) >> 1, f_t = ((z * arg.Y + y) * X1 + xf) >> 1;

  const int Xh = arg.Xh, sy = Xh, sz = Xh * arg.Y, st = Xh * arg.Y * arg.Z;
  const int n_xp = xodd ? (xh == Xh - 1 ? idx - (Xh - 1) : idx + 1) : idx;
  const int n_xm = xodd ? idx : (xh == 0 ? idx + (Xh - 1) : idx - 1);
  const int n_yp = y == arg.Y - 1 ? idx - (arg.Y - 1) * sy : idx + sy;
  const int n_ym = y == 0 ? idx + (arg.Y - 1) * sy : idx - sy;
  const int n_zp = z == arg.Z - 1 ? idx - (arg.Z - 1) * sz : idx + sz;
  const int n_zm = z == 0 ? idx + (arg.Z - 1) * sz : idx - sz;
  const int n_tp = t == arg.T - 1 ? idx - (arg.T - 1) * st : idx + st;
  const int n_tm = t == 0 ? idx + (arg.T - 1) * st : idx - st;

  real acc[24];
#pragma unroll
  for (int k = 0; k < 24; k++) acc[k] = 0;

  constexpr bool PT = VARIANT == 1;
  const real one = 1;
  const real sg_tp = t == arg.T - 1 ? arg.tsign_fwd : one, sg_tm = t == 0 ? arg.tsign_bwd : one;
  // direction order: dir = 2 mu + (0 forward, 1 backward), as the reference (tests/dslash_util.h:131-140)
  RawBlock<T, 24> pA, pB, pC;
  typename Link<T, R>::Raw uA, uB, uC;
  // QA_FENCE: nothing may be scheduled across (machine scheduler).  QA_PIN: an empty volatile asm that "modifies" the
  // accumulators, so LLVM's IR-level sinking cannot push a hop's arithmetic past the following loads either (without it
  // all 8 hops' FMAs sink below the last fence and every loaded register stays live: 390-512 registers, scratch spills).
#define QA_FENCE() __builtin_amdgcn_sched_barrier(0)
#define QA_PIN()                                                   \
  _Pragma("unroll") for (int k_ = 0; k_ < 24; k_++) asm volatile("" : "+v"(acc[k_]))
  // Hop order x+ x- y+ y- z+ z- t+ t- (the reference's direction numbering).  Measured and dropped: t- z- y- x- x+ y+ z+ t+
  // (aligned with an increasing sweep, so that the reads of one input site by different blocks fall closer together in time)
  // changed nothing at 32^4 or 48^3 x 96 in any precision (gpurun_out/sweep48b.log of round 2).
  constexpr int GH = KT == 2 ? 1 : (KT == 3 ? 2 : 0);
  const int nb[8] = {n_xp, n_xm, n_yp, n_ym, n_zp, n_zm, n_tp, n_tm};
  const real sg[8] = {one, one, one, one, one, one, sg_tp, sg_tm};
  const bool of[8] = {o_xp, o_xm, o_yp, o_ym, o_zp, o_zm, o_tp, o_tm};
  const int fc[8] = {f_x, f_x, f_y, f_y, f_z, f_z, f_t, f_t};
#define QA_LD(B, D) hop_load<T, R, D, GAUX, GH>(p##B, u##B, arg, idx, nb[D], of[D], fc[D])
#define QA_CP(B, D) QA_FENCE(); hop_compute<T, R, D, PT, GH>(acc, p##B, u##B, arg, sg[D], of[D], fc[D]); QA_PIN(); QA_FENCE()
#define QA_PIPE(d0, d1, d2, d3, d4, d5, d6, d7)                                                      \
  QA_LD(A, d0); QA_LD(B, d1); QA_CP(A, d0); QA_LD(A, d2); QA_CP(B, d1); QA_LD(B, d3); QA_CP(A, d2);  \
  QA_LD(A, d4); QA_CP(B, d3); QA_LD(B, d5); QA_CP(A, d4); QA_LD(A, d6); QA_CP(B, d5); QA_LD(B, d7);  \
  QA_CP(A, d6); QA_FENCE(); hop_compute<T, R, d7, PT, GH>(acc, pB, uB, arg, sg[d7], of[d7], fc[d7])
  // 16-bit storage: a hop in flight is 22 registers (raw image) and ~1.4 KB per wave, half of what the fp32 kernel keeps in flight at
  // the same depth — three hops deep instead of two
#define QA_PIPE3()                                                                                                     \
  QA_LD(A, 0); QA_LD(B, 1); QA_LD(C, 2); QA_CP(A, 0); QA_LD(A, 3); QA_CP(B, 1); QA_LD(B, 4); QA_CP(C, 2); QA_LD(C, 5); \
  QA_CP(A, 3); QA_LD(A, 6); QA_CP(B, 4); QA_LD(B, 7); QA_CP(C, 5); QA_CP(A, 6); QA_FENCE();                            \
  hop_compute<T, R, 7, PT, GH>(acc, pB, uB, arg, sg[7], of[7], fc[7])
#if QA_P2P_FOLD
  if (KT == 3) {
    // Folded packing (launchDslash decides): thread `threadIdx.x` < packShare of one of the first packFoldBlocks blocks also packs one
    // face site.  Its load is issued in front of the first two hops' loads and its stores behind them, so the stores — whose
    // acknowledgement a later s_waitcnt would have to sit out, vector memory operations of a wave retire in order — are older than
    // everything the pipeline waits for from hop 2 on, and by then long acknowledged.
    const int item = (int)blockIdx.x * arg.packShare + (int)threadIdx.x;
    const bool packs = (int)blockIdx.x < arg.packFoldBlocks && (int)threadIdx.x < arg.packShare && item < arg.pack.start[8];
    if (__builtin_amdgcn_ballot_w64(packs) != 0) {
      real pk[24];
      int slot, f;
      const int pidx = pack_locate(arg.pack, packs ? item : 0, slot, f);
      Planar<T, 24>::load(pk, arg.pack.in, arg.pack.sp_stride, pidx, arg.pack.inNorm, pidx);
      QA_LD(A, 0);
      QA_FENCE();
      if (!packs) f = 0x07ffffff;   // beyond the zone's record count: the buffer stores of this lane are dropped
      pack_emit<T, PT, true>(arg.pack, pk, slot, f);
      QA_FENCE();
    } else {
      QA_LD(A, 0);
    }
    QA_LD(B, 1); QA_CP(A, 0); QA_LD(A, 2); QA_CP(B, 1); QA_LD(B, 3); QA_CP(A, 2);
    QA_LD(A, 4); QA_CP(B, 3); QA_LD(B, 5); QA_CP(A, 4); QA_LD(A, 6); QA_CP(B, 5); QA_LD(B, 7);
    QA_CP(A, 6); QA_FENCE(); hop_compute<T, R, 7, PT, GH>(acc, pB, uB, arg, sg[7], of[7], fc[7]);
  } else if (sizeof(T) == 2) {
    QA_PIPE3();
  } else {
    QA_PIPE(0, 1, 2, 3, 4, 5, 6, 7);
  }
#else
  if (sizeof(T) == 2) { QA_PIPE3(); } else { QA_PIPE(0, 1, 2, 3, 4, 5, 6, 7); }
#endif
#undef QA_PIPE
#undef QA_PIPE3
#undef QA_LD
#undef QA_CP
#undef QA_FENCE
#undef QA_PIN

  if (KT == 3) {
    // the hops that leave this rank: wait (only the waves that own boundary sites) until the neighbours' faces are in, then add them
    const bool anyoff = o_xp || o_xm || o_yp || o_ym || o_zp || o_zm || o_tp || o_tm;
    if (__builtin_amdgcn_ballot_w64(anyoff) != 0) {
      if (arg.timeline && (threadIdx.x & 63) == 0) arg.timeline[4096 + (blockIdx.x * 4 + (threadIdx.x >> 6))] = wall_clock64();
      // (two directions in flight per wave — requests of the next while working on this one — measured slower: 26.6 against 25.2 us
      // fp64 on the 8-GPU sub-lattice, the exec-masked merges of the forward and backward lanes cost more than the overlap gains)
      ghost_hop<T, R, 0, GAUX>(acc, arg, idx, o_xp, f_x, one);
      ghost_hop<T, R, 1, GAUX>(acc, arg, idx, o_xm, f_x, one);
      ghost_hop<T, R, 2, GAUX>(acc, arg, idx, o_yp, f_y, one);
      ghost_hop<T, R, 3, GAUX>(acc, arg, idx, o_ym, f_y, one);
      ghost_hop<T, R, 4, GAUX>(acc, arg, idx, o_zp, f_z, one);
      ghost_hop<T, R, 5, GAUX>(acc, arg, idx, o_zm, f_z, one);
      ghost_hop<T, R, 6, GAUX>(acc, arg, idx, o_tp, f_t, sg_tp);
      ghost_hop<T, R, 7, GAUX>(acc, arg, idx, o_tm, f_t, sg_tm);
      if (arg.timeline && (threadIdx.x & 63) == 0) arg.timeline[8192 + (blockIdx.x * 4 + (threadIdx.x >> 6))] = wall_clock64();
    }
  }
  dslash_epilogue<T, VARIANT, GAUX, SAUX>(acc, arg, idx);
}

template <typename T, int R, int VARIANT, int GAUX, int KT, int SAUX = 0>
__global__ void __launch_bounds__(256) dslash_kernel(const DslashArg<typename Store<T>::real> arg) {
  if (KT == 2) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= arg.nboundary) return;
    stencil_site<T, R, VARIANT, GAUX, 2, SAUX>(arg, arg.blist[tid]);
    return;
  }
  int b = blockIdx.x;
  if (KT == 3) {
    // single-launch peer-store path: [pack blocks | every site]; boundary sites do their local hops first, then poll the
    // ghost words they need and add the off-node hops (stencil_site, KT == 3)
    if (b < arg.packBlocks) {
      if (arg.packPrio) __builtin_amdgcn_s_setprio(3);
      pack_body<T, VARIANT == 1, true>(arg.pack, b, arg.packChunk);
      return;
    }
    b -= arg.packBlocks;
    // the pack blocks' loads go first: a site wave keeps ~8 us worth of requests queued in its CU, and a pack wave that starts
    // together with it needs 12 (up to 18) us for its two round trips instead of 4
    for (int i = 0; i < arg.siteDelay; i++) __builtin_amdgcn_s_sleep(8);
  }
  // XCD-aware block remap: blocks b, b+8, b+16, ... share an XCD (round-robin dispatch); give each XCD a
  // contiguous range of logical blocks (a slab of time slices) so t/z neighbours hit its own L2.
  const int xcd = b & 7, within = b >> 3;
  int lb;
  if (arg.tiled) {
    const uint32_t xt = arg.dNxz.div((uint32_t)xcd), xz = (uint32_t)xcd - xt * (uint32_t)arg.nxz;
    const uint32_t tile = arg.dPerTile.div((uint32_t)within), r = (uint32_t)within - tile * arg.dPerTile.d;
    const uint32_t tile_t = arg.dNtz.div(tile), tile_z = tile - tile_t * arg.dNtz.d;
    uint32_t yc, zi, ti;
    if (arg.tiled == 3) {
      // y groups: the XCD walks its whole (z-slab, t) range once per GROUP of plane chunks — group slowest, then t, z, chunk in the
      // group.  The set an XCD has to hold for the +-t re-use (three time slices of its slab) shrinks by the number of groups; fp64
      // at 48^3 x 96: 3 x 6 planes x 221 KB = 4 MB, the whole L2, without groups.  (tile = group here: dPerTile = Ts Zs Pg, dTzTt = Zs Pg,
      // dPTt = Pg, tz = Zs, tt = 1)
      const uint32_t tl3 = arg.dTzTt.div(r), r2 = r - tl3 * arg.dTzTt.d;
      zi = arg.dPTt.div(r2);
      yc = tile * arg.dPTt.d + (r2 - zi * arg.dPTt.d);
      ti = tl3;
    } else if (arg.tiled == 2) {   // inside a tile: z slowest, then the chunk of the plane, t fastest (lexicographic for tt = 1)
      zi = arg.dPTt.div(r); const uint32_t r2 = r - zi * arg.dPTt.d;
      yc = arg.dTt.div(r2); ti = r2 - yc * arg.dTt.d;
    } else {                // chunk slowest, then z, t fastest
      yc = arg.dTzTt.div(r); const uint32_t r2 = r - yc * arg.dTzTt.d;
      zi = arg.dTt.div(r2); ti = r2 - zi * arg.dTt.d;
    }
    int zl = (int)tile_z * arg.tz + (int)zi, tl = (int)tile_t * arg.tt + (int)ti;
    if (arg.tiled == 3) { zl = (int)zi; tl = (int)ti; }
    if (arg.edgeFirst) {
      // partitioned launch: an XCD starts with the blocks that own boundary sites (they have the most to do: the off-node hops come
      // on top, after a wait), and the blocks dispatched last — the ones that share a CU with a pack block — are interior ones.
      // z: the upper half of the XCDs walks its slab downwards; t: a slab that spans the whole extent starts at T - 1 and wraps to
      // 0 (still a contiguous sweep on the periodic lattice), otherwise the upper half walks downwards as well.
      if (2 * (int)xz >= arg.nxz) zl = arg.Zs - 1 - zl;
      if (arg.nxz == 8) tl = tl == 0 ? arg.Ts - 1 : tl - 1;
      else if (xt) tl = arg.Ts - 1 - tl;
    }
    const int z = (int)xz * arg.Zs + zl, t = (int)xt * arg.Ts + tl;
    lb = (t * arg.Z + z) * arg.P + (int)yc;
  } else {
    lb = (xcd < arg.xcd_r ? xcd * (arg.xcd_q + 1) : arg.xcd_r * (arg.xcd_q + 1) + (xcd - arg.xcd_r) * arg.xcd_q) + within;
    if (arg.xcd_q < 0) lb = b;   // remap off: consecutive blocks dealt round-robin over the XCDs (even spread of the boundary work)
    if (arg.ts > 0) {
      // inside an XCD's slab of ts time slices walk t fastest: consecutive blocks are the same (y,z) chunk on ts successive
      // slices, so the +-t neighbours of a chunk are touched within a few blocks of each other (L2-resident) instead of a
      // whole slice apart
      const int per = arg.bps * arg.ts, s = lb / per, w = lb - s * per, c = w / arg.ts, tt = w - c * arg.ts;
      lb = (s * arg.ts + tt) * arg.bps + c;
    }
  }
  const int idx = lb * blockDim.x + threadIdx.x;
  if (idx >= arg.Vh) return;
  if (KT == 3 && arg.timeline && threadIdx.x == 0) {
    arg.timeline[2048 + blockIdx.x] = wall_clock64();
    arg.timeline[3072 + blockIdx.x] = 0x100000000ull | ((unsigned long long)(__builtin_amdgcn_s_getreg(20 | (31 << 11)) & 0xf) << 16) | (__builtin_amdgcn_s_getreg(4 | (31 << 11)) & 0xff00u);
  }
  stencil_site<T, R, VARIANT, GAUX, KT, SAUX>(arg, idx);
  if (KT == 3 && arg.timeline && threadIdx.x == 0) arg.timeline[12288 + blockIdx.x] = wall_clock64();
}

// ================================================================================================
// Multi-right-hand-side full operator on block fields (block.h layout: [site][spin-colour j][rhs i] float2), fp32, recon 18:
//     out(x) = (1 + i a g5) in(x) - kappa sum_{8 hops} U P in(x + mu)          on the sites of ONE parity per launch
// for NRHS right-hand sides per link load — the operator behind the lockstep null-vector solves of the multigrid setup
// (multigrid.cpp; reference MG::generateNullVectors, lib/multigrid.cpp:693-779, where every one of the Nvec BiCGstab solves
// re-reads the links: 576 of the 768 B per site of the fp32 stencil).  One thread per (site, right-hand side); the links of the
// work-group's sites are staged once in LDS (full-line loads from the planar link blocks, broadcast reads), the spinor panels are
// read as 8-byte pairs, NRHS x 8 B contiguous per (site, component).  NRHS = 8 keeps the panel of a site at 768 B so that the
// neighbour re-use still fits the caches in the XCD-slab order of the single-vector kernel: 72 + ~110 + 96 + 96 B per site and
// right-hand side instead of 768 + 96.
// ================================================================================================
struct BlockOrder {   // the plane-tiled XCD mapping of dslash_kernel for work-groups of `bs` consecutive checkerboard sites
  int tiled, P, nxz, Zs, Ts, tz, tt, Z;
  FastDiv dNxz, dPerTile, dNtz, dPTt, dTt;
  __device__ __forceinline__ int map(int b) const {
    if (!tiled) return b;
    const int xcd = b & 7, within = b >> 3;
    const uint32_t xt = dNxz.div((uint32_t)xcd), xz = (uint32_t)xcd - xt * (uint32_t)nxz;
    const uint32_t tile = dPerTile.div((uint32_t)within), r = (uint32_t)within - tile * dPerTile.d;
    const uint32_t tile_t = dNtz.div(tile), tile_z = tile - tile_t * dNtz.d;
    const uint32_t zi = dPTt.div(r), r2 = r - zi * dPTt.d;
    const uint32_t yc = dTt.div(r2), ti = r2 - yc * dTt.d;
    const int z = (int)xz * Zs + (int)tile_z * tz + (int)zi, t = (int)xt * Ts + (int)tile_t * tt + (int)ti;
    return (t * Z + z) * P + (int)yc;
  }
};
static BlockOrder makeBlockOrder(const LatticeGeom &g, int bs) {
  BlockOrder o;
  memset(&o, 0, sizeof(o));
  const int plane = g.Xh * g.X[1];
  if (plane % bs || (g.Vh / bs) % 8) return o;
  int nxz = 0;
  for (int c : {8, 4, 2, 1}) if (g.X[2] % c == 0 && g.X[3] % (8 / c) == 0) { nxz = c; break; }
  if (!nxz) return o;
  o.tiled = 1; o.P = plane / bs; o.nxz = nxz; o.Zs = g.X[2] / nxz; o.Ts = g.X[3] / (8 / nxz); o.tz = o.Zs; o.tt = 1; o.Z = g.X[2];
  o.dNxz = FastDiv((uint32_t)nxz); o.dPerTile = FastDiv((uint32_t)(o.P * o.tz * o.tt)); o.dNtz = FastDiv((uint32_t)(o.Zs / o.tz));
  o.dPTt = FastDiv((uint32_t)(o.P * o.tt)); o.dTt = FastDiv((uint32_t)o.tt);
  return o;
}

struct FineBlockArg {
  float2 *out;             // panels of the output parity
  const float2 *in_same;   // panels of the same parity (twist term)
  const float2 *in_other;  // panels of the other parity (hops)
  const char *gauge;       // this parity's 8 direction blocks
  size_t link_bytes;
  int g_stride;
  int Vh, Xh, Y, Z, T, parity;
  FastDiv dXh, dY, dZ;
  // out = s0 (1 + i a0 g5) in_same + k1 (1 + i a1 g5) [sum of the 8 hops of in_other]; s0 = 0: in_same is not read
  float s0, a0, k1, a1;
  // twisted clover (kernel template CL): a dense site matrix [site][chirality][6 x 6 complex] of the OUTPUT parity replaces the factor
  // (1 + i a1 g5) of the hop sum (CL = 1: (A + i a g5)^-1 of the even-odd preconditioned operator) or (1 + i a0 g5) of in_same
  // (CL = 2: A + i a g5 of the full operator)
  const float *tmat;
  // grid-decomposed lattice: a hop across a partitioned face reads the neighbour rank's panel from the ghost zone of in_other (block.h
  // BlockGhost, parity field): ghostBase[d][k] = panel index, RELATIVE to in_other, of zone [d][k]
  int commMask;
  int ghostBase[4][2];
  BlockOrder order;
  // compact work-group tiles (NRHS = 8, 32 sites per work-group): 4 x 4 x 2 x 2 lattice sites = 32 sites of the output parity instead of 32
  // consecutive checkerboard sites.  The 256 neighbour panels a work-group reads are then only 128 distinct ones (every input site inside
  // the tile is the neighbour of up to 8 of its output sites), so half of the requests can be served by the CU's L1 instead of the L2:
  // the kernel is bound by the latency of its L2 requests (~100 KB in flight per CU), not by HBM.  tile = 0: the linear mapping.
  // MEASURED (round 3, 48^4, tools/fine_block_timing.py): 2278 us per parity launch with the tiles against 2027 us with the linear mapping
  // (twisted clover 2568 / 2281) — the scattered link staging and the lost XCD z-slab order cost more than the L1 hits gain; kept as an
  // opt-in (QUDA_AMD_BLOCK_FINE_TILE=1) for the record, the linear mapping stays the default.
  int tile, tilesX, tilesY, tilesZ;
  // inner products in the epilogue (NRHS = 8; dslash.h FineBlockDots): per right-hand side, summed over the work-group's sites in site order,
  // one row of partials per work-group.  dotMode 1: (a, out) [2 sums]; 2: (out, same) [2], |out|^2 [1], (a, same) [2], (a, out) [2]
  const float2 *dotA;
  double *dotPart;
  int dotMode;
  // 12-real links (recon-12): the third row is rebuilt while the links are staged; tsign_*: sign of the rebuilt row of the t links that carry the
  // folded antiperiodic boundary (forward links of the last time slice, backward links of the first), as in DslashArg
  int recon;
  float tsign_fwd, tsign_bwd;
};

// XY = 1 (NRHS = 8, CL = 0): the work-group is an 8 (x, checkerboard) x 4 (y) tile of one (z, t) plane and the panels its x and y hops need — 56 instead of
// 4 x 32 — are fetched ONCE into LDS (dynamic, 60 slots of 896 B); the z and t hops keep the register pipeline.  The kernel is bound by its L2 -> L1 requests
// (profiles/r03_fine_block_kernel_pmc.log): this takes a quarter of them out, at the price of two instead of four work-groups per CU (72.7 KB of LDS, 138 registers).
// MEASURED (48^4, tools/fine_block_timing.py): 2459 us per parity launch against 1874 us for the linear mapping — the lost occupancy costs more than the
// requests saved.  Correct under every partition mask (tools/fine_block_xy_check.py); kept as QUDA_AMD_BLOCK_FINE_XYTILE=1 for the record, off by default.
template <int NRHS, int CL = 0, int XY = 0> __global__ void __launch_bounds__(256) fine_block_kernel(const FineBlockArg arg) {
  constexpr int SPB = NRHS == 24 ? 8 : 256 / NRHS;   // sites per work-group (192 threads for 24 right-hand sides, else 256)
  constexpr int USTR = 8 * 18 + 4;                    // floats per site of staged links: 148 = 20 mod 32, the up to 8 sites of a wave start on 8 different banks
  constexpr int TSTR = 144 + 4;                       // the dense clover-twist matrices of a site: 2 x 36 complex, same bank spread
  __shared__ float ulds[SPB * USTR];
  __shared__ __attribute__((aligned(16))) float tlds[CL ? SPB * TSTR : 4];
  const int s = threadIdx.x / NRHS, i = threadIdx.x - s * NRHS;
  extern __shared__ __attribute__((aligned(16))) float plds[];   // XY: staged x / y neighbour panels
  constexpr int PSTR = 224;   // floats per staged panel: 192 + 32 — consecutive slots start 128 B apart modulo the 256-byte bank row (two sites per 16-lane phase of a ds_read_b128)
  // site ss of this work-group -> checkerboard index
  int idx0 = 0, tbase = 0;
  int tX0 = 0, tY0 = 0, tZ = 0, tT = 0;
  if (XY) {
    const int m = arg.order.map(blockIdx.x), P = arg.tilesX * arg.tilesY;   // (t Z + z) P + tile of the plane, in the XCD-slab order of the linear mapping
    const int yc = m % P, zt = m / P;
    tZ = zt % arg.Z; tT = zt / arg.Z;
    tX0 = (yc % arg.tilesX) * 8; tY0 = (yc / arg.tilesX) * 4;
    tbase = ((tT * arg.Z + tZ) * arg.Y + tY0) * arg.Xh + tX0;
  } else if (NRHS == 8 && arg.tile) {
    // work-groups dealt to the XCDs in contiguous eighths of the tile list (t slowest), tile -> its corner
    const int nwg = (int)gridDim.x, b = (int)blockIdx.x;
    int tl = (nwg & 7) == 0 ? (b & 7) * (nwg >> 3) + (b >> 3) : b;
    const int tx = tl % arg.tilesX; tl /= arg.tilesX;
    const int ty = tl % arg.tilesY; tl /= arg.tilesY;
    const int tz = tl % arg.tilesZ; const int tt = tl / arg.tilesZ;
    tbase = ((2 * tt * arg.Z + 2 * tz) * arg.Y + 4 * ty) * arg.Xh + 2 * tx;
  } else {
    idx0 = arg.order.map(blockIdx.x) * SPB;
  }
  auto site_of = [&](int ss) -> int {
    if (XY) return tbase + (ss >> 3) * arg.Xh + (ss & 7);   // ss = y_l 8 + xh_l
    if (NRHS == 8 && arg.tile) {   // ss = ((t_l 2 + z_l) 4 + y_l) 2 + xh_l
      const int xl = ss & 1, yl = (ss >> 1) & 3, zl = (ss >> 3) & 1, tl = ss >> 4;
      return tbase + ((tl * arg.Z + zl) * arg.Y + yl) * arg.Xh + xl;
    }
    return idx0 + ss;
  };
  const int idx = site_of(s);
  if (CL) {   // 36 float4 per site, contiguous in memory
    for (int e = threadIdx.x; e < 36 * SPB; e += blockDim.x) {
      const int ss = e / 36, q = e - ss * 36;
      typedef float f32x4_t __attribute__((ext_vector_type(4)));
      const int sidx = site_of(ss);
      if (sidx < arg.Vh) *reinterpret_cast<f32x4_t *>(&tlds[ss * TSTR + 4 * q]) = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t *>(arg.tmat) + (size_t)sidx * 36 + q);
    }
  }
  // ---- stage the links of the SPB sites: [site][dir][18]; planes 0..3 are float4, plane 4 the trailing float2 ----
  if (arg.recon == 12) {
    // one thread per (site, direction): two stored rows in three 16-byte planes, the third rebuilt here — once per work-group, for NRHS right-hand sides
    for (int e = threadIdx.x; e < 8 * SPB; e += blockDim.x) {
      const int ss = e % SPB, d = e / SPB;
      const int sidx = site_of(ss);
      if (sidx < arg.Vh) {
        float sign = 1.f;
        if (d >= 6) {
          const int tt = sidx / (arg.Xh * arg.Y * arg.Z);
          sign = d == 6 ? (tt == arg.T - 1 ? arg.tsign_fwd : 1.f) : (tt == 0 ? arg.tsign_bwd : 1.f);
        }
        float U[18];
        Link<float, 12>::load(U, arg.gauge + (size_t)d * arg.link_bytes, arg.g_stride, sidx, sign);
        float *dst = &ulds[ss * USTR + d * 18];
#pragma unroll
        for (int k = 0; k < 18; k++) dst[k] = U[k];
      }
    }
  } else
  for (int e = threadIdx.x; e < 8 * 5 * SPB; e += blockDim.x) {
    const int ss = e % SPB, pl = (e / SPB) % 5, d = e / (5 * SPB);
    const char *blk = arg.gauge + (size_t)d * arg.link_bytes;
    float *dst = &ulds[ss * USTR + d * 18 + pl * 4];
    const int sidx = site_of(ss);
    if (sidx < arg.Vh) {
      if (pl < 4) {
        const float4 v = reinterpret_cast<const float4 *>(blk)[(size_t)pl * arg.g_stride + sidx];
        dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
      } else {
        const float2 v = reinterpret_cast<const float2 *>(blk + (size_t)4 * arg.g_stride * 16)[sidx];
        dst[0] = v.x; dst[1] = v.y;
      }
    }
  }
  if (XY) {
    // slots (row r = 0..5: y0 - 1 .. y0 + 4, column c = 0..9: xh0 - 1 .. xh0 + 8), 48 sixteen-byte words each; the corner slots belong to no hop (they are
    // filled with a valid panel all the same: unconditional loads, issued back to back, then the LDS stores).  A slot outside the lattice wraps around or,
    // on a partitioned dimension, is the neighbour rank's panel in the ghost zone — column 0 only serves the -x hop, column 9 +x, row 0 -y, row 5 +y.
    float4 stg[12];
#pragma unroll
    for (int q = 0; q < 12; q++) {
      int e = (int)threadIdx.x + 256 * q;
      e = e < 2880 ? e : 2879;
      const int slot = e / 48, w = e - slot * 48, r = slot / 10, c = slot - r * 10;
      int yy = tY0 + r - 1, xx = tX0 + c - 1;
      long pn = -1;
      if (xx < 0) { if (arg.commMask & 1) pn = arg.ghostBase[0][0] + ((((tT * arg.Z + tZ) * arg.Y + (yy < 0 ? 0 : (yy >= arg.Y ? arg.Y - 1 : yy)))) >> 1); else xx = arg.Xh - 1; }
      else if (xx >= arg.Xh) { if (arg.commMask & 1) pn = arg.ghostBase[0][1] + ((((tT * arg.Z + tZ) * arg.Y + (yy < 0 ? 0 : (yy >= arg.Y ? arg.Y - 1 : yy)))) >> 1); else xx = 0; }
      if (pn < 0) {
        if (yy < 0) { if (arg.commMask & 2) pn = arg.ghostBase[1][0] + (tT * arg.Z + tZ) * arg.Xh + xx; else yy = arg.Y - 1; }
        else if (yy >= arg.Y) { if (arg.commMask & 2) pn = arg.ghostBase[1][1] + (tT * arg.Z + tZ) * arg.Xh + xx; else yy = 0; }
      }
      if (pn < 0) pn = ((long)(tT * arg.Z + tZ) * arg.Y + yy) * arg.Xh + xx;
      stg[q] = reinterpret_cast<const float4 *>(arg.in_other + pn * 12 * NRHS)[w];
    }
#pragma unroll
    for (int q = 0; q < 12; q++) {
      const int e = (int)threadIdx.x + 256 * q;
      if (e < 2880) { const int slot = e / 48, w = e - slot * 48; *reinterpret_cast<float4 *>(&plds[slot * PSTR + 4 * w]) = stg[q]; }
    }
  }
  __syncthreads();
  if (idx >= arg.Vh) return;
  // coordinates and neighbours (as stencil_site)
  const uint32_t za = arg.dXh.div((uint32_t)idx);
  const int xh = idx - (int)za * arg.Xh;
  const uint32_t zb = arg.dY.div(za);
  const int y = (int)za - (int)zb * arg.Y;
  const int t = (int)arg.dZ.div(zb);
  const int z = (int)zb - t * arg.Z;
  const int xodd = (y + z + t + arg.parity) & 1;
  const int Xh = arg.Xh, sy = Xh, sz = Xh * arg.Y, st = Xh * arg.Y * arg.Z;
  int nb[8];
  nb[0] = xodd ? (xh == Xh - 1 ? idx - (Xh - 1) : idx + 1) : idx;
  nb[1] = xodd ? idx : (xh == 0 ? idx + (Xh - 1) : idx - 1);
  nb[2] = y == arg.Y - 1 ? idx - (arg.Y - 1) * sy : idx + sy;
  nb[3] = y == 0 ? idx + (arg.Y - 1) * sy : idx - sy;
  nb[4] = z == arg.Z - 1 ? idx - (arg.Z - 1) * sz : idx + sz;
  nb[5] = z == 0 ? idx + (arg.Z - 1) * sz : idx - sz;
  nb[6] = t == arg.T - 1 ? idx - (arg.T - 1) * st : idx + st;
  nb[7] = t == 0 ? idx + (arg.T - 1) * st : idx - st;
  if (arg.commMask) {   // face index: the other three coordinates, lexicographic, halved (block.hip face_site_to_panel packs in this order)
    if (arg.commMask & 1) {
      const int f = ((t * arg.Z + z) * arg.Y + y) >> 1;
      if (xodd && xh == Xh - 1) nb[0] = arg.ghostBase[0][1] + f;
      if (!xodd && xh == 0) nb[1] = arg.ghostBase[0][0] + f;
    }
    if (arg.commMask & 2) {
      const int f = (t * arg.Z + z) * Xh + xh;
      if (y == arg.Y - 1) nb[2] = arg.ghostBase[1][1] + f;
      if (y == 0) nb[3] = arg.ghostBase[1][0] + f;
    }
    if (arg.commMask & 4) {
      const int f = (t * arg.Y + y) * Xh + xh;
      if (z == arg.Z - 1) nb[4] = arg.ghostBase[2][1] + f;
      if (z == 0) nb[5] = arg.ghostBase[2][0] + f;
    }
    if (arg.commMask & 8) {
      const int f = (z * arg.Y + y) * Xh + xh;
      if (t == arg.T - 1) nb[6] = arg.ghostBase[3][1] + f;
      if (t == 0) nb[7] = arg.ghostBase[3][0] + f;
    }
  }

  float acc[24];
#pragma unroll
  for (int k = 0; k < 24; k++) acc[k] = 0.f;
  const float *U0 = &ulds[s * USTR];
  // Same register discipline as stencil_site: two panel buffers, the loads of hop d + 1 issued before the arithmetic of hop d,
  // fenced so the compiler neither hoists all 108 loads to the top (256 registers, one wave per SIMD: 12.7 ms per launch at
  // 48^3 x 96, four times the bandwidth bound) nor serialises them.
  // The panels travel as 16-byte words, and the block field is PAIR-MAJOR (block.h): word ((site 6 + k) NRHS + i) holds components 2k, 2k + 1 of
  // right-hand side i, so a lane's six loads are its own 12 components and the NRHS lanes of a site still read whole 128-byte lines (8-byte
  // loads run at 0.54-0.70 of the 16-byte rate on gfx950, and the kernel is bound by its L2 -> L1 requests).  History: with the rhs-fastest
  // order the lane pair (i, i ^ 1) shared its loads and transposed 2 x 2 by DPP — 620 of the kernel's ~1950 vector instructions.
  float4 pA[6], pB[6];
#ifndef QA_FB_PANEL_POLICY
#define QA_FB_PANEL_POLICY 0   // 0 plain loads, 1 non-temporal (L1 bypassed), 2 sc1 raw buffer loads: measured, see DESIGN 3
#endif
  auto load_panel = [&](float4 *raw, const float2 *base, int site) {
    const float4 *p = reinterpret_cast<const float4 *>(base) + (long)site * 6 * NRHS + i;
#pragma unroll
    for (int k = 0; k < 6; k++) {
      if (QA_FB_PANEL_POLICY == 1) {
        typedef float f32x4_t __attribute__((ext_vector_type(4)));
        const f32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t *>(p + k * NRHS));
        raw[k] = make_float4(v.x, v.y, v.z, v.w);
      } else {
        raw[k] = p[k * NRHS];
      }
    }
  };
  auto unpack_panel = [&](float *psi, const float4 *raw) {
#pragma unroll
    for (int k = 0; k < 6; k++) { psi[4 * k] = raw[k].x; psi[4 * k + 1] = raw[k].y; psi[4 * k + 2] = raw[k].z; psi[4 * k + 3] = raw[k].w; }
  };
  auto hop = [&](auto DIRC, const float4 *raw) {
    constexpr int DIR = decltype(DIRC)::value, MU = DIR / 2;
    float psi[24], h[12], g[12], U[18];
    unpack_panel(psi, raw);
#pragma unroll
    for (int k = 0; k < 18; k++) U[k] = U0[DIR * 18 + k];
    const float sgn = (DIR & 1) ? -1.f : 1.f;
    spin_project<MU>(h, psi, sgn);
    su3_mv(g, U, h);
    su3_mv(g + 6, U, h + 6);
    spin_reconstruct<MU>(acc, g, sgn);
  };
#define FB_FENCE() __builtin_amdgcn_sched_barrier(0)
#define FB_PIN() _Pragma("unroll") for (int k_ = 0; k_ < 24; k_++) asm volatile("" : "+v"(acc[k_]))
#define FB_LD(B, D) load_panel(p##B, arg.in_other, nb[D])
#define FB_CP(B, D) FB_FENCE(); hop(std::integral_constant<int, D>{}, p##B); FB_PIN(); FB_FENCE()
  if constexpr (XY) {
    // z / t panels requested first, the four x / y hops out of LDS while they travel
    float4 pC[6];
    const int r0 = (s >> 3) + 1, c0 = (s & 7) + 1;
    auto load_lds = [&](float4 *raw, int slot) {
      const float *b = &plds[slot * PSTR + 4 * i];
#pragma unroll
      for (int k = 0; k < 6; k++) raw[k] = *reinterpret_cast<const float4 *>(b + k * 4 * NRHS);   // word k NRHS + i of the panel
    };
    FB_LD(A, 4); FB_LD(B, 5);
    load_lds(pC, r0 * 10 + c0 + (xodd ? 1 : 0)); FB_CP(C, 0);
    load_lds(pC, r0 * 10 + c0 - (xodd ? 0 : 1)); FB_CP(C, 1);
    load_lds(pC, (r0 + 1) * 10 + c0); FB_CP(C, 2);
    load_lds(pC, (r0 - 1) * 10 + c0); FB_CP(C, 3);
    FB_CP(A, 4); FB_LD(A, 6); FB_CP(B, 5); FB_LD(B, 7);
    FB_CP(A, 6);
    if (arg.s0 != 0.f) load_panel(pA, arg.in_same, idx);
    FB_CP(B, 7);
  } else {
  FB_LD(A, 0); FB_LD(B, 1); FB_CP(A, 0); FB_LD(A, 2); FB_CP(B, 1); FB_LD(B, 3); FB_CP(A, 2);
  FB_LD(A, 4); FB_CP(B, 3); FB_LD(B, 5); FB_CP(A, 4); FB_LD(A, 6); FB_CP(B, 5); FB_LD(B, 7);
  FB_CP(A, 6);
  if (arg.s0 != 0.f) load_panel(pA, arg.in_same, idx);   // the site's own panel travels while the last hop is computed
  FB_CP(B, 7);
  }
#undef FB_FENCE
#undef FB_PIN
#undef FB_LD
#undef FB_CP
  float same[24];
  if (arg.s0 != 0.f) unpack_panel(same, pA);
  float outv[24];
  if (CL) {
    // dense 6 x 6 complex matrix per chirality on the hop sum (CL = 1) or on the site's own panel (CL = 2); rows read as three
    // 16-byte LDS words, the same address for the NRHS lanes of a site
    float *src = CL == 1 ? acc : same;
    float res[24];
    const float *T0 = &tlds[s * TSTR];
#pragma unroll
    for (int chi = 0; chi < 2; chi++)
#pragma unroll
      for (int r = 0; r < 6; r++) {
        float re = 0.f, im = 0.f;
#pragma unroll
        for (int q = 0; q < 3; q++) {
          const float4 m = *reinterpret_cast<const float4 *>(&T0[(chi * 6 + r) * 12 + 4 * q]);
          const float *v = &src[12 * chi + 4 * q];
          re += m.x * v[0] - m.y * v[1] + m.z * v[2] - m.w * v[3];
          im += m.x * v[1] + m.y * v[0] + m.z * v[3] + m.w * v[2];
        }
        res[12 * chi + 2 * r] = re; res[12 * chi + 2 * r + 1] = im;
      }
#pragma unroll
    for (int j = 0; j < 12; j++) {
      float re, im;
      if (CL == 1) {
        re = arg.k1 * res[2 * j]; im = arg.k1 * res[2 * j + 1];
        if (arg.s0 != 0.f) {
          const float a0 = (j < 6 ? 1.f : -1.f) * arg.a0;
          re += arg.s0 * (same[2 * j] - a0 * same[2 * j + 1]); im += arg.s0 * (same[2 * j + 1] + a0 * same[2 * j]);
        }
      } else {
        const float a1 = (j < 6 ? 1.f : -1.f) * arg.a1;
        re = arg.k1 * (acc[2 * j] - a1 * acc[2 * j + 1]) + arg.s0 * res[2 * j];
        im = arg.k1 * (acc[2 * j + 1] + a1 * acc[2 * j]) + arg.s0 * res[2 * j + 1];
      }
      outv[2 * j] = re; outv[2 * j + 1] = im;
    }
  } else {
#pragma unroll
    for (int j = 0; j < 12; j++) {
      const float sg = j < 6 ? 1.f : -1.f;   // g5 = diag(+,+,-,-): spins 0,1 are components 0..5
      const float a1 = sg * arg.a1;
      float re = arg.k1 * (acc[2 * j] - a1 * acc[2 * j + 1]), im = arg.k1 * (acc[2 * j + 1] + a1 * acc[2 * j]);
      if (arg.s0 != 0.f) {
        const float a0 = sg * arg.a0;
        re += arg.s0 * (same[2 * j] - a0 * same[2 * j + 1]); im += arg.s0 * (same[2 * j + 1] + a0 * same[2 * j]);
      }
      outv[2 * j] = re; outv[2 * j + 1] = im;
    }
  }
  if (NRHS <= 8 && arg.dotMode) {   // uniform
    // the two dot products of a BiCGstab half step where their operands already are: `out` and the site's own input panel in registers, one more
    // panel (a = r0) loaded — instead of separate passes over the fields (blockblas::cDot / bicgstabDots: 2 resp. 3 field reads); mode 3: the
    // two sums of a minimal-residual step, (out, same) and |out|^2, nothing loaded at all
    float av[24];
    if (arg.dotMode != 3) {
      float4 ra[6];
      load_panel(ra, arg.dotA, idx);
      unpack_panel(av, ra);
    }
    const int ns = arg.dotMode == 1 ? 2 : (arg.dotMode == 3 ? 3 : 7);
    float sm[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (arg.dotMode == 3) {
#pragma unroll
      for (int j = 0; j < 12; j++) {
        const float orr = outv[2 * j], oi = outv[2 * j + 1], sr = same[2 * j], si = same[2 * j + 1];
        sm[0] += orr * sr + oi * si; sm[1] += orr * si - oi * sr;    // conj(out) same
        sm[2] += orr * orr + oi * oi;
      }
    } else if (arg.dotMode == 1) {
#pragma unroll
      for (int j = 0; j < 12; j++) {
        sm[0] += av[2 * j] * outv[2 * j] + av[2 * j + 1] * outv[2 * j + 1];       // conj(a) out
        sm[1] += av[2 * j] * outv[2 * j + 1] - av[2 * j + 1] * outv[2 * j];
      }
    } else {
#pragma unroll
      for (int j = 0; j < 12; j++) {
        const float orr = outv[2 * j], oi = outv[2 * j + 1], sr = same[2 * j], si = same[2 * j + 1], ar = av[2 * j], ai = av[2 * j + 1];
        sm[0] += orr * sr + oi * si; sm[1] += orr * si - oi * sr;    // conj(out) same
        sm[2] += orr * orr + oi * oi;
        sm[3] += ar * sr + ai * si; sm[4] += ar * si - ai * sr;      // conj(a) same
        sm[5] += ar * orr + ai * oi; sm[6] += ar * oi - ai * orr;    // conj(a) out
      }
    }
    __syncthreads();   // every wave is through its hops: the staged links are dead, their LDS holds the partial sums now
    float *red = ulds;
    for (int k = 0; k < ns; k++) red[k * 256 + threadIdx.x] = sm[k];
    __syncthreads();
    if ((int)threadIdx.x < ns * NRHS) {
      const int k = threadIdx.x / NRHS, ii = threadIdx.x - k * NRHS;
      int nact = SPB;
      if (!arg.tile) { const int left = arg.Vh - idx0; nact = left < SPB ? left : SPB; }
      double t = 0.0;
      for (int ss = 0; ss < nact; ss++) t += (double)red[k * 256 + ss * NRHS + ii];
      arg.dotPart[((size_t)blockIdx.x * ns + k) * NRHS + ii] = t;
    }
  }
  float4 *o = reinterpret_cast<float4 *>(arg.out) + (size_t)idx * 6 * NRHS + i;
#pragma unroll
  for (int k = 0; k < 6; k++) o[k * NRHS] = make_float4(outv[4 * k], outv[4 * k + 1], outv[4 * k + 2], outv[4 * k + 3]);
}

// Dense clover-twist matrices for fine_block_kernel: per site of one parity and chirality the 6 x 6 complex matrix
//     A + i a s      (s = +1 upper, -1 lower chirality; inverse = false)      or its inverse      (inverse = true)
// from the packed Hermitian clover blocks.  The inverse is a Gauss-Jordan elimination in fp64 on the matrix itself, not the
// stored (A^2 + mu2)^-1 field (reference lib/clover_invert.cu:56-85), so it is exact for the a it is given — the multigrid setup may
// run with a rescaled mu (delta_muPR, reference lib/interface_quda.cpp:2196-2211).  One thread per (site, chirality).
__global__ void __launch_bounds__(128) clover_twist_dense_kernel(float *out, const void *clA, int cl_stride, int Vh, double a, int inverse) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= 2 * Vh) return;
  const int idx = e >> 1, chi = e & 1;
  float C[36];
  Planar<float, 36>::load(C, (const char *)clA + (size_t)chi * 36 * sizeof(float) * cl_stride, cl_stride, idx, nullptr, 0);
  double mr[6][6], mi[6][6];
  for (int i = 0; i < 6; i++)
    for (int j = 0; j < 6; j++) {
      if (i == j) { mr[i][j] = C[i]; mi[i][j] = chi ? -a : a; }
      else if (j < i) { const int k = 15 - (6 - j) * (5 - j) / 2 + i - j - 1; mr[i][j] = C[6 + 2 * k]; mi[i][j] = C[6 + 2 * k + 1]; }
      else { const int k = 15 - (6 - i) * (5 - i) / 2 + j - i - 1; mr[i][j] = C[6 + 2 * k]; mi[i][j] = -C[6 + 2 * k + 1]; }
    }
  float *o = out + ((size_t)idx * 2 + chi) * 72;
  if (!inverse) {
    for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) { o[(i * 6 + j) * 2] = (float)mr[i][j]; o[(i * 6 + j) * 2 + 1] = (float)mi[i][j]; }
    return;
  }
  // the Hermitian part A is positive definite on any sensible field and i a s only adds to the diagonal: no pivoting needed, but
  // the largest remaining diagonal element is taken anyway
  double vr[6][6], vi[6][6];
  for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) { vr[i][j] = i == j ? 1.0 : 0.0; vi[i][j] = 0.0; }
  for (int c = 0; c < 6; c++) {
    int p = c;
    double best = mr[c][c] * mr[c][c] + mi[c][c] * mi[c][c];
    for (int r = c + 1; r < 6; r++) { const double m2 = mr[r][c] * mr[r][c] + mi[r][c] * mi[r][c]; if (m2 > best) { best = m2; p = r; } }
    if (p != c)
      for (int j = 0; j < 6; j++) {
        double t = mr[c][j]; mr[c][j] = mr[p][j]; mr[p][j] = t; t = mi[c][j]; mi[c][j] = mi[p][j]; mi[p][j] = t;
        t = vr[c][j]; vr[c][j] = vr[p][j]; vr[p][j] = t; t = vi[c][j]; vi[c][j] = vi[p][j]; vi[p][j] = t;
      }
    const double dr = mr[c][c] / best, di = -mi[c][c] / best;   // 1 / pivot
    for (int j = 0; j < 6; j++) {
      double xr = mr[c][j] * dr - mi[c][j] * di, xi = mr[c][j] * di + mi[c][j] * dr; mr[c][j] = xr; mi[c][j] = xi;
      xr = vr[c][j] * dr - vi[c][j] * di; xi = vr[c][j] * di + vi[c][j] * dr; vr[c][j] = xr; vi[c][j] = xi;
    }
    for (int r = 0; r < 6; r++) {
      if (r == c) continue;
      const double fr = mr[r][c], fi = mi[r][c];
      for (int j = 0; j < 6; j++) {
        mr[r][j] -= fr * mr[c][j] - fi * mi[c][j]; mi[r][j] -= fr * mi[c][j] + fi * mr[c][j];
        vr[r][j] -= fr * vr[c][j] - fi * vi[c][j]; vi[r][j] -= fr * vi[c][j] + fi * vr[c][j];
      }
    }
  }
  for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) { o[(i * 6 + j) * 2] = (float)vr[i][j]; o[(i * 6 + j) * 2 + 1] = (float)vi[i][j]; }
}

void cloverTwistDense(float *out, const CloverField &C, int parity, double a, bool inverse) {
  if (C.precision != QUDA_SINGLE_PRECISION) errorQuda("dense clover-twist matrices are built from an fp32 clover field (got precision %d)", C.precision);
  const int Vh = C.geom.Vh;
  hipLaunchKernelGGL(clover_twist_dense_kernel, dim3((2 * Vh + 127) / 128), dim3(128), 0, computeStream(), out, C.A(parity), C.stride, Vh, a, inverse ? 1 : 0);
  HIP_CHECK(hipGetLastError());
}

bool fineBlockSupported(const GaugeField &U, int nrhs) {
  if (U.precision != QUDA_SINGLE_PRECISION || (U.reconstruct != QUDA_RECONSTRUCT_NO && U.reconstruct != QUDA_RECONSTRUCT_12)) return false;
  if (nrhs != 4 && nrhs != 8 && nrhs != 16 && nrhs != 24 && nrhs != 32) return false;
  static int off = -1;
  if (off < 0) { const char *e = getenv("QUDA_AMD_BLOCK_FINE"); off = (e && !atoi(e)) ? 1 : 0; }
  return !off;
}

// one parity of the generalised multi-right-hand-side stencil:
//   out(x) = s0 (1 + i a0 g5) in_same(x) + k1 (1 + i a1 g5) sum_{8 hops} U P in_other(x + mu)       x of parity `parity`
// all three fields are single-parity block panels [Vh][12][nrhs]; in_same may be nullptr when s0 = 0
// ---- epilogue inner products of fine_block_kernel: work-group partials -> 128 chunk sums -> pinned host memory, in a fixed order ----
static double *g_fbPart = nullptr, *g_fbPart2 = nullptr, *g_fbRes = nullptr, *g_fbResDev = nullptr;
static size_t g_fbPartBytes = 0;
static int g_fbBlocks = 0, g_fbVals = 0;
constexpr int kFbChunks = 128;
__global__ void __launch_bounds__(512) fine_block_dots_reduce(double *part2, const double *part, int nblocks, int nval) {
  // work-group g sums its contiguous range of partial rows; thread (row, val): flat, contiguous reads
  __shared__ double red[512];
  const int rows = blockDim.x / nval, row = threadIdx.x / nval, val = threadIdx.x - row * nval;
  const int per = (nblocks + gridDim.x - 1) / gridDim.x, b0 = blockIdx.x * per, b1 = min(nblocks, b0 + per);
  double t = 0.0;
  if (row < rows) for (int b = b0 + row; b < b1; b += rows) t += part[(size_t)b * nval + val];
  red[threadIdx.x] = row < rows ? t : 0.0;
  __syncthreads();
  if ((int)threadIdx.x < nval) {
    double u = 0.0;
    for (int r = 0; r < rows; r++) u += red[r * nval + threadIdx.x];
    part2[(size_t)blockIdx.x * nval + threadIdx.x] = u;
  }
}
__global__ void fine_block_dots_finish(double *res, const double *part2, int nchunks, int nval) {
  const int v = blockIdx.x, lane = threadIdx.x;
  double t = 0.0;
  for (int c = lane; c < nchunks; c += 64) t += part2[(size_t)c * nval + v];
  for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off, 64);
  if (lane == 0) res[v] = t;
}
void freeFineBlockDots() {
  if (g_fbPart) { poolDeviceFree(g_fbPart, 0); g_fbPart = nullptr; g_fbPartBytes = 0; }
  if (g_fbPart2) { (void)hipFree(g_fbPart2); g_fbPart2 = nullptr; }
  if (g_fbRes) { (void)hipHostFree(g_fbRes); g_fbRes = nullptr; g_fbResDev = nullptr; }
}
bool fineBlockDotsSupported(int nrhs) {
  static int off = -1;
  if (off < 0) { const char *e = getenv("QUDA_AMD_BLOCK_FINE_DOTS"); off = (e && !atoi(e)) ? 1 : 0; }
  return !off && (nrhs == 8 || nrhs == 4);
}
static int fbDotSums(int mode) { return mode == 1 ? 2 : (mode == 3 ? 3 : 7); }
// after a launch with dots: sums[k * nrhs + i], k as in FineBlockArg::dotMode; a global sum on a grid-decomposed lattice
static void fbDotsReduce(double *dst, int nrhs, int mode) {
  const int ns = fbDotSums(mode), nval = ns * nrhs;
  if (!g_fbBlocks || g_fbVals != nval) errorQuda("no multi-right-hand-side stencil launch with inner products (mode %d) to finish", mode);
  const int threads = 512 / nval * nval;
  hipLaunchKernelGGL(fine_block_dots_reduce, dim3(kFbChunks), dim3(threads), 0, computeStream(), g_fbPart2, (const double *)g_fbPart, g_fbBlocks, nval);
  hipLaunchKernelGGL(fine_block_dots_finish, dim3(nval), dim3(64), 0, computeStream(), dst, (const double *)g_fbPart2, kFbChunks, nval);
  HIP_CHECK(hipGetLastError());
  g_fbBlocks = 0;
}
void fineBlockDotsFinish(double *sums, int nrhs, int mode) {
  const int nval = fbDotSums(mode) * nrhs;
  fbDotsReduce(g_fbResDev, nrhs, mode);
  HIP_CHECK(hipStreamSynchronize(computeStream()));
  for (int q = 0; q < nval; q++) sums[q] = g_fbRes[q];
  if (commReductionsNeeded()) comm_allreduce(sums, nval);
}
// the same into DEVICE memory, rank-local, no host round trip: the next kernel on the stream reads the sums (blockblas::mrUpdateDev)
void fineBlockDotsFinishDev(double *d_sums, int nrhs, int mode) { fbDotsReduce(d_sums, nrhs, mode); }

void applyFineBlockParity(float2 *out, const float2 *in_same, const float2 *in_other, int nrhs, const GaugeField &U, int parity, double s0, double a0, double k1,
                          double a1, const float *tmat, int tmode, float2 *ghost, const FineBlockDots *dots) {
  if (!fineBlockSupported(U, nrhs)) errorQuda("multi-right-hand-side fine operator: fp32 links (18 or 12 reals), 4/8/16/24/32 right-hand sides");
  if (s0 != 0.0 && !in_same) errorQuda("same-parity input missing");
  if (tmat && (tmode != 1 && tmode != 2)) errorQuda("site-matrix mode %d (1: on the hop sum, 2: on the same-parity input)", tmode);
  if (tmat && tmode == 2 && s0 == 0.0) errorQuda("site matrix on the same-parity input, but that input is switched off");
  const LatticeGeom &g = U.geom;
  const int spb = nrhs == 24 ? 8 : 256 / nrhs, threads = spb * nrhs;
  FineBlockArg arg;
  arg.link_bytes = U.link_bytes; arg.g_stride = U.stride;
  arg.Vh = g.Vh; arg.Xh = g.Xh; arg.Y = g.X[1]; arg.Z = g.X[2]; arg.T = g.X[3];
  arg.dXh = g.dXh; arg.dY = g.dY; arg.dZ = g.dZ;
  arg.s0 = (float)s0; arg.a0 = (float)a0; arg.k1 = (float)k1; arg.a1 = (float)a1;
  arg.order = makeBlockOrder(g, spb);
  int nb = (g.Vh + spb - 1) / spb;
  arg.tile = 0; arg.tilesX = arg.tilesY = arg.tilesZ = 1;
  {
    static int tileEnv = -1;
    if (tileEnv < 0) { const char *e = getenv("QUDA_AMD_BLOCK_FINE_TILE"); tileEnv = e ? atoi(e) : 0; }   // measured SLOWER (48^4: 2278 against 2027 us per launch): off
    if (tileEnv && nrhs == 8 && g.X[0] % 4 == 0 && g.X[1] % 4 == 0 && g.X[2] % 2 == 0 && g.X[3] % 2 == 0) {
      arg.tile = 1; arg.tilesX = g.X[0] / 4; arg.tilesY = g.X[1] / 4; arg.tilesZ = g.X[2] / 2;
      nb = arg.tilesX * arg.tilesY * arg.tilesZ * (g.X[3] / 2);   // = Vh / 32
    }
  }
  arg.parity = parity;
  arg.out = out; arg.in_same = in_same ? in_same : in_other; arg.in_other = in_other;
  arg.gauge = (const char *)U.parityBase(parity);
  arg.tmat = tmat;
  arg.recon = (int)U.reconstruct;
  {
    const bool first_t = commGrid().coords[3] == 0, last_t = commGrid().coords[3] == commGrid().dims[3] - 1;
    arg.tsign_fwd = (U.reconstruct == QUDA_RECONSTRUCT_12 && U.t_boundary == QUDA_ANTI_PERIODIC_T && last_t) ? -1.f : 1.f;
    arg.tsign_bwd = (U.reconstruct == QUDA_RECONSTRUCT_12 && U.t_boundary == QUDA_ANTI_PERIODIC_T && first_t) ? -1.f : 1.f;
  }
  arg.dotA = nullptr; arg.dotPart = nullptr; arg.dotMode = 0;
  if (dots) {
    if (!fineBlockDotsSupported(nrhs)) errorQuda("inner products in the stencil epilogue: 4 or 8 right-hand sides only");
    if (dots->mode < 1 || dots->mode > 3) errorQuda("inner-product mode %d", dots->mode);
    if (dots->mode >= 2 && s0 == 0.0) errorQuda("inner products with the same-parity input, but that input is switched off");
    const int ns = fbDotSums(dots->mode);
    const size_t need = (size_t)nb * ns * nrhs * sizeof(double);
    if (need > g_fbPartBytes) {
      if (g_fbPart) poolDeviceFree(g_fbPart, 0);
      g_fbPart = (double *)poolDeviceMalloc(need);
      g_fbPartBytes = need;
    }
    if (!g_fbPart2) {
      HIP_CHECK(qaMalloc((void **)&g_fbPart2, (size_t)kFbChunks * 7 * 8 * sizeof(double)));
      HIP_CHECK(hipHostMalloc((void **)&g_fbRes, 7 * 8 * sizeof(double), hipHostMallocMapped));
      HIP_CHECK(hipHostGetDevicePointer((void **)&g_fbResDev, g_fbRes, 0));
    }
    arg.dotA = dots->a; arg.dotPart = g_fbPart; arg.dotMode = dots->mode;
    g_fbBlocks = nb; g_fbVals = ns * nrhs;
  }
  // grid-decomposed lattice: the faces of in_other (parity 1 - parity) go to the neighbours' ghost zones first — one pack launch and one
  // grouped exchange on the compute stream (setup-time traffic; the solve-time stencil has its own overlapped transports, halo.h)
  arg.commMask = 0;
  for (int d = 0; d < 4; d++) arg.ghostBase[d][0] = arg.ghostBase[d][1] = 0;
  {
    const BlockGhost gh = blockGhost(g.X, true);
    if (gh.mask) {
      if (!ghost) errorQuda("multi-right-hand-side stencil on a grid-decomposed lattice: the input needs a ghost zone (block.h BlockGhost)");
      blockExchangeGhostRaw(in_other, ghost, 12, nrhs, gh, 1 - parity);
      const long panel = 12l * nrhs, rel = ghost - in_other;
      if (rel % panel) errorQuda("ghost zone not aligned with the panels of its field");
      arg.commMask = gh.mask;
      for (int d = 0; d < 4; d++) for (int k = 0; k < 2; k++) arg.ghostBase[d][k] = (int)(rel / panel) + gh.offset[d][k];
    }
  }
  {
    // the x / y tile variant (QUDA_AMD_BLOCK_FINE_XYTILE): 8 right-hand sides, no site matrices, planes that 8 x 4 tiles cover
    static int xyEnv = -1;
    if (xyEnv < 0) { const char *e = getenv("QUDA_AMD_BLOCK_FINE_XYTILE"); xyEnv = e ? atoi(e) : 0; }
    if (xyEnv && nrhs == 8 && !tmat && !arg.tile && g.Xh % 8 == 0 && g.X[1] % 4 == 0) {
      arg.tilesX = g.Xh / 8; arg.tilesY = g.X[1] / 4;
      constexpr size_t ldsXY = (size_t)60 * 224 * sizeof(float);
      static bool attr = false;
      if (!attr) { HIP_CHECK(hipFuncSetAttribute((const void *)fine_block_kernel<8, 0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsXY)); attr = true; }
      hipLaunchKernelGGL((fine_block_kernel<8, 0, 1>), dim3(nb), dim3(threads), ldsXY, computeStream(), arg);
      HIP_CHECK(hipGetLastError());
      return;
    }
  }
  if (g_acctOn) {   // links once per site (+ the dense site matrices), per right-hand side: 8 neighbour panels at ideal re-use = 1 read, own panel, output
    char tag[64]; snprintf(tag, sizeof(tag), "level 0, %d rhs%s%s", nrhs, s0 != 0.0 ? " xpay" : "", dots ? " + sums" : "");
    acct("fine_block_kernel", (double)g.Vh * (8.0 * (U.reconstruct == QUDA_RECONSTRUCT_12 ? 48 : 72) + (tmat ? 576.0 : 0.0) + nrhs * 96.0 * (s0 != 0.0 ? 3 : 2)), tag);
  }
#define FB_LAUNCH(N) \
  if (!tmat) hipLaunchKernelGGL((fine_block_kernel<N, 0>), dim3(nb), dim3(threads), 0, computeStream(), arg); \
  else if (tmode == 1) hipLaunchKernelGGL((fine_block_kernel<N, 1>), dim3(nb), dim3(threads), 0, computeStream(), arg); \
  else hipLaunchKernelGGL((fine_block_kernel<N, 2>), dim3(nb), dim3(threads), 0, computeStream(), arg)
  switch (nrhs) {
    case 4: FB_LAUNCH(4); break;
    case 8: FB_LAUNCH(8); break;
    case 16: FB_LAUNCH(16); break;
    case 24: FB_LAUNCH(24); break;
    default: FB_LAUNCH(32); break;
  }
#undef FB_LAUNCH
  HIP_CHECK(hipGetLastError());
}

// out = (1 + i a g5) in - kappa D in on full block fields of 12 components (both parities, two launches)
// grid-decomposed lattice: `in` carries 2 x blockGhost(X, true).nGhost panels behind its V local ones — the ghost zone of the even
// half, then that of the odd half
void applyFineBlockM(float2 *out, float2 *in, int nrhs, const GaugeField &U, double kappa, double a, const float *const tmat[2]) {
  const size_t par = (size_t)U.geom.Vh * 12 * nrhs;
  const BlockGhost gh = blockGhost(U.geom.X, true);
  for (int p = 0; p < 2; p++)
    applyFineBlockParity(out + p * par, in + p * par, in + (1 - p) * par, nrhs, U, p, 1.0, a, -kappa, 0.0, tmat ? tmat[p] : nullptr, 2,
                         gh.mask ? in + 2 * par + (size_t)(1 - p) * gh.nGhost * 12 * nrhs : nullptr);
}

// ---- single-direction hop (setup-time helper for the multigrid coarse-operator construction) ----
template <typename T, int R>
__global__ void __launch_bounds__(256) hop_dir_kernel(const DslashArg<typename Store<T>::real> arg, int dir, typename Store<T>::real coef,
                                                      const void *ghost, int ghostFaceCB, int plain) {
  using real = typename Store<T>::real;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= arg.Vh) return;
  const uint32_t za = arg.dXh.div((uint32_t)idx);
  const int xh = idx - (int)za * arg.Xh;
  const uint32_t zb = arg.dY.div(za);
  const int y = (int)za - (int)zb * arg.Y;
  const int t = (int)arg.dZ.div(zb);
  const int z = (int)zb - t * arg.Z;
  const int xodd = (y + z + t + arg.parity) & 1;
  const int Xh = arg.Xh, sy = Xh, sz = Xh * arg.Y, st = Xh * arg.Y * arg.Z;
  int nbr;
  real sign = 1;
  switch (dir) {
    case 0: nbr = xodd ? (xh == Xh - 1 ? idx - (Xh - 1) : idx + 1) : idx; break;
    case 1: nbr = xodd ? idx : (xh == 0 ? idx + (Xh - 1) : idx - 1); break;
    case 2: nbr = y == arg.Y - 1 ? idx - (arg.Y - 1) * sy : idx + sy; break;
    case 3: nbr = y == 0 ? idx + (arg.Y - 1) * sy : idx - sy; break;
    case 4: nbr = z == arg.Z - 1 ? idx - (arg.Z - 1) * sz : idx + sz; break;
    case 5: nbr = z == 0 ? idx + (arg.Z - 1) * sz : idx - sz; break;
    case 6: nbr = t == arg.T - 1 ? idx - (arg.T - 1) * st : idx + st; if (t == arg.T - 1) sign = arg.tsign_fwd; break;
    default: nbr = t == 0 ? idx + (arg.T - 1) * st : idx - st; if (t == 0) sign = arg.tsign_bwd; break;
  }
  real acc[24], psi[24], U[18];
#pragma unroll
  for (int k = 0; k < 24; k++) acc[k] = 0;
  // grid-decomposed direction: the neighbour of a face site lives on the adjacent rank; its full spinor was exchanged into
  // `ghost` (face index = lexicographic index of the other three coordinates, halved — same convention as pack_kernel)
  bool cross = false;
  int f = 0;
  if (ghost) {
    const int xf = 2 * xh + xodd, X0 = 2 * Xh;
    switch (dir >> 1) {
      case 0: cross = (dir & 1) ? xf == 0 : xf == X0 - 1; f = (y + arg.Y * (z + arg.Z * t)) >> 1; break;
      case 1: cross = (dir & 1) ? y == 0 : y == arg.Y - 1; f = (xf + X0 * (z + arg.Z * t)) >> 1; break;
      case 2: cross = (dir & 1) ? z == 0 : z == arg.Z - 1; f = (xf + X0 * (y + arg.Y * t)) >> 1; break;
      default: cross = (dir & 1) ? t == 0 : t == arg.T - 1; f = (xf + X0 * (y + arg.Y * z)) >> 1; break;
    }
  }
  if (cross) Planar<T, 24>::load(psi, ghost, ghostFaceCB, f, nullptr, f);
  else Planar<T, 24>::load(psi, arg.in, arg.sp_stride, nbr, arg.inNorm, nbr);
  Link<T, R>::load(U, arg.gauge + (size_t)dir * arg.link_bytes, arg.g_stride, idx, sign);
  if (plain) {
    // covariant shift without spin projection: acc = U psi on all four spins (the links of the backward directions are
    // stored daggered, so this is U_mu(x - mu)^dagger psi(x - mu) there)
#pragma unroll
    for (int sp = 0; sp < 4; sp++) su3_mv(acc + 6 * sp, U, psi + 6 * sp);
  } else
  switch (dir) {
    case 0: hop_arith<0, false, 0>(acc, psi, U, arg); break;
    case 1: hop_arith<1, false, 0>(acc, psi, U, arg); break;
    case 2: hop_arith<2, false, 0>(acc, psi, U, arg); break;
    case 3: hop_arith<3, false, 0>(acc, psi, U, arg); break;
    case 4: hop_arith<4, false, 0>(acc, psi, U, arg); break;
    case 5: hop_arith<5, false, 0>(acc, psi, U, arg); break;
    case 6: hop_arith<6, false, 0>(acc, psi, U, arg); break;
    default: hop_arith<7, false, 0>(acc, psi, U, arg); break;
  }
  if (arg.xpay) {   // out = k x + coef hop (x may be the output field itself: every thread reads its site before it writes it)
    real xs[24];
    Planar<T, 24>::load(xs, arg.x, arg.sp_stride, idx, arg.xNorm, idx);
#pragma unroll
    for (int k = 0; k < 24; k++) acc[k] = arg.k * xs[k] + coef * acc[k];
  } else {
#pragma unroll
    for (int k = 0; k < 24; k++) acc[k] *= coef;
  }
  Planar<T, 24>::store(acc, arg.out, arg.sp_stride, idx, arg.outNorm, idx);
}

// ---- site-local kernel: twist / clover / twisted clover and inverse (reference lib/dslash_quda.cu:348-600) ----
template <typename real> struct SiteArg {
  void *out; float *outNorm;
  const void *in; const float *inNorm;
  const void *clA, *clAinv;
  const float *clAn, *clAinvN;
  int sp_stride, cl_stride, Vh, op;
  real a, b;
};

template <typename T, bool CLOVER>
__global__ void __launch_bounds__(256) site_kernel(const SiteArg<typename Store<T>::real> arg) {
  using real = typename Store<T>::real;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= arg.Vh) return;
  real v[24];
  Planar<T, 24>::load(v, arg.in, arg.sp_stride, idx, arg.inNorm, idx);
  if (arg.op == SITE_TWIST) {
    twist_inplace(v, arg.a);
#pragma unroll
    for (int k = 0; k < 24; k++) v[k] *= arg.b;
  } else if (CLOVER) {
    real C[36], tmp[24];
#pragma unroll
    for (int chi = 0; chi < 2; chi++) {
      Planar<T, 36>::load(C, (const char *)arg.clA + (size_t)chi * 36 * sizeof(T) * arg.cl_stride, arg.cl_stride, idx, arg.clAn,
                          chi * arg.cl_stride + idx);
      clover_block_mv(tmp + 12 * chi, C, v + 12 * chi);
    }
    if (arg.op != SITE_CLOVER) {
#pragma unroll
      for (int k = 0; k < 6; k++) { tmp[2 * k] -= arg.a * v[2 * k + 1]; tmp[2 * k + 1] += arg.a * v[2 * k]; }
#pragma unroll
      for (int k = 6; k < 12; k++) { tmp[2 * k] += arg.a * v[2 * k + 1]; tmp[2 * k + 1] -= arg.a * v[2 * k]; }
    }
    if (arg.op == SITE_CLOVER_TWIST_INV) {
#pragma unroll
      for (int chi = 0; chi < 2; chi++) {
        Planar<T, 36>::load(C, (const char *)arg.clAinv + (size_t)chi * 36 * sizeof(T) * arg.cl_stride, arg.cl_stride, idx,
                            arg.clAinvN, chi * arg.cl_stride + idx);
        clover_block_mv(v + 12 * chi, C, tmp + 12 * chi);
      }
    } else {
#pragma unroll
      for (int k = 0; k < 24; k++) v[k] = tmp[k];
    }
  }
  Planar<T, 24>::store(v, arg.out, arg.sp_stride, idx, arg.outNorm, idx);
}

// ------------------------------------------------------------------------------------------------
// launch-geometry knobs of the stencil (defaults from the environment, once; qudaAmdSetDslashTune changes them at run time so one
// process can sweep them — tools/dslash_sweep.py)
DslashTune &dslashTune() {
  static DslashTune t;
  static bool init = false;
  if (!init) {
    init = true;
    auto env = [](const char *n, int d) { const char *e = getenv(n); return e ? atoi(e) : d; };
    t.block = env("QUDA_AMD_DSLASH_BLOCK", 0);
    t.remap = env("QUDA_AMD_XCD_REMAP", 1);
    t.order = env("QUDA_AMD_DSLASH_ORDER", 1);
    t.store_aux = env("QUDA_AMD_STORE_AUX", -1);
    t.link_aux = env("QUDA_AMD_LINK_AUX", -1);
    t.tiled = env("QUDA_AMD_DSLASH_TILED", -1);
    t.nxz = env("QUDA_AMD_DSLASH_NXZ", 0);
    t.tz = env("QUDA_AMD_DSLASH_TZ", 0);
    t.tt = env("QUDA_AMD_DSLASH_TT", 0);
    t.lds_pad = env("QUDA_AMD_DSLASH_LDS", 0);
    t.ygroups = env("QUDA_AMD_DSLASH_YGROUPS", -1);
    t.edge_first = env("QUDA_AMD_EDGE_FIRST", 1);
    { const char *e = getenv("QUDA_AMD_HALO_FORMAT"); t.halo_format = !e ? -1 : ((!strcmp(e, "atom") || !strcmp(e, "atom16") || !strcmp(e, "sector") || !strcmp(e, "1")) ? 1 : 0); }
  }
  return t;
}
// automatic (-1): the compact atoms between DEVICES — two thirds of flag-in-data's bytes on the wire (fp64 / fp32; three quarters for 16-bit): in a
// 1 x 2 x 2 x 2 grid the +mu and -mu neighbour are the same GPU, so both faces of a dimension share one xGMI link, 1.57 MB per link and application in
// fp64 against 1.05 MB, ~26 against ~17 us at 60 GB/s next to a 20 us interior kernel — and flag-in-data where no link is crossed (ranks sharing a device in
// a rehearsal, the self-neighbour emulation on one rank).  Both formats validate every single store by itself (8-byte halves resp. 16-byte atoms), neither
// relies on the order in which stores become visible.  The same on every rank by construction.
int haloWireFormat() {
  const int f = dslashTune().halo_format;
  if (f >= 0) return f ? 1 : 0;
  return (commGrid().size > 1 && !p2pDeviceShared()) ? 1 : 0;
}
void setDslashTune(const char *key, int value) {
  DslashTune &t = dslashTune();
  const std::string k(key);
  if (k == "block") t.block = value;
  else if (k == "remap") t.remap = value;
  else if (k == "order") t.order = value;
  else if (k == "store_aux") t.store_aux = value;
  else if (k == "link_aux") t.link_aux = value;
  else if (k == "tiled") t.tiled = value;
  else if (k == "nxz") t.nxz = value;
  else if (k == "tz") t.tz = value;
  else if (k == "tt") t.tt = value;
  else if (k == "lds_pad") t.lds_pad = value;
  else if (k == "ygroups") t.ygroups = value;
  else if (k == "p2p_fold") { t.p2p_fold = value; if (value > 0 && !QA_P2P_FOLD) warningQuda("folded face packing is compiled out (QA_P2P_FOLD = 0 in dslash.hip): pack blocks are used"); }
  else if (k == "site_delay") t.site_delay = value;
  else if (k == "pack_prio") t.pack_prio = value;
  else if (k == "edge_first") t.edge_first = value;
  else if (k == "halo_format") t.halo_format = value;
  else errorQuda("unknown stencil tuning key '%s'", key);
}

template <typename T, bool PRETWIST, bool P2P = false> __global__ void __launch_bounds__(256) pack_kernel(const PackArg<typename Store<T>::real> arg) {
  pack_body<T, PRETWIST, P2P>(arg, blockIdx.x, (int)blockDim.x);
}

// ---- ghost-zone storage and the boundary-site lists ----
static HaloBuffers g_halo[3];
static std::vector<BoundaryList> g_blists;

static void releaseHalo(HaloBuffers &h) {
  if (h.window) {
    // neighbours may still be storing into / polling this window: drain the device and meet them before unmapping
    HIP_CHECK(hipDeviceSynchronize());
    commBarrier();
    commUnmapPeers(h.map);
    commBarrier();
    p2pFree(h.window);
  }
  if (h.pool) HIP_CHECK(hipFree(h.pool));
  h = HaloBuffers();
}

HaloBuffers &haloBuffers(const LatticeGeom &g, QudaPrecision prec) {
  HaloBuffers &h = g_halo[prec == QUDA_DOUBLE_PRECISION ? 0 : (prec == QUDA_SINGLE_PRECISION ? 1 : 2)];
  bool same = h.precision == prec;
  for (int d = 0; d < 4; d++) same = same && h.faceCB[d] == g.faceCB[d];
  if (same && h.pool) return h;
  releaseHalo(h);
  h.precision = prec;
  size_t total = 0;
  for (int d = 0; d < 4; d++) {
    h.faceCB[d] = g.faceCB[d];
    const size_t payload = ((size_t)g.faceCB[d] * 12 * (int)prec + 255) / 256 * 256;
    h.norm_offset[d] = payload;
    h.face_bytes[d] = payload + (prec == QUDA_HALF_PRECISION ? ((size_t)g.faceCB[d] * sizeof(float) + 255) / 256 * 256 : 0);
    total += 4 * h.face_bytes[d];
  }
  HIP_CHECK(qaMalloc((void **)&h.pool, total));
  // hipMemset on the null stream may still be in flight when it returns and does not order against the non-blocking
  // compute/comm streams: a late memset would wipe a freshly packed send buffer, so zero on the compute stream and drain
  HIP_CHECK(hipMemsetAsync(h.pool, 0, total, computeStream()));
  HIP_CHECK(hipStreamSynchronize(computeStream()));
  h.pool_bytes = total;
  char *p = h.pool;
  for (int d = 0; d < 4; d++)
    for (int dir = 0; dir < 2; dir++) { h.send[d][dir] = p; p += h.face_bytes[d]; h.ghost[d][dir] = p; p += h.face_bytes[d]; }

  // ---- peer-store transport: same offsets on every rank (same local lattice), so a neighbour address = its base + my offset
  h.p2p = p2pHaloEnabled();
  if (h.p2p) {
    // window = double-buffered flag-in-data zones [dim][k][buf], each NV 16-byte vectors per face site (GhostLL); zero-filled,
    // and the flag of an exchange is never 0
    const int nv = prec == QUDA_DOUBLE_PRECISION ? GhostLL<double>::NV : (prec == QUDA_SINGLE_PRECISION ? GhostLL<float>::NV : GhostLL<short>::NV);
    size_t wtotal = 0;
    size_t zone[4];
    for (int d = 0; d < 4; d++) { zone[d] = ((size_t)g.faceCB[d] * nv * 16 + 255) / 256 * 256; wtotal += 4 * zone[d]; }
    h.window = (char *)p2pAlloc(wtotal + 256);
    if (!commMapPeers(h.window, h.map)) errorQuda("peer mapping of a halo window failed after the transport probe succeeded");
    HIP_CHECK(hipDeviceSynchronize());
    size_t off = 0;
    for (int d = 0; d < 4; d++)
      for (int k = 0; k < 2; k++) {
        // k = 0: zone filled by the -d neighbour (it sent forward), k = 1: by the +d neighbour (it sent backward)
        const int to_fwd = k == 0 ? 1 : 0;                 // the sender's direction that fills zone k
        const int slot = 2 * d + (to_fwd ? 1 : 0);         // MY neighbour in that direction receives my (d, to_fwd) face in ITS zone k
        for (int buf = 0; buf < 2; buf++) {
          h.ghostBuf[d][k][buf] = h.window + off;
          h.peerGhost[d][to_fwd][buf] = (char *)h.map.peer[slot] + off;
          off += zone[d];
        }
      }
    h.seq = 0;
    for (int d = 0; d < 4; d++) h.uses[d][0] = h.uses[d][1] = 0;
    commBarrier();
  }
  return h;
}
void freeFullFaceBuffers();
void freeCoarseGhosts();  // coarse.hip
void freeHaloBuffers() {
  for (HaloBuffers &h : g_halo) releaseHalo(h);
  freeBoundaryLists();
  freeFullFaceBuffers();
  freeCoarseGhosts();
}

const BoundaryList &boundaryList(const LatticeGeom &g, int mask) {
  for (const BoundaryList &b : g_blists)
    if (b.mask == mask && b.X[0] == g.X[0] && b.X[1] == g.X[1] && b.X[2] == g.X[2] && b.X[3] == g.X[3]) return b;
  BoundaryList b;
  b.mask = mask;
  for (int d = 0; d < 4; d++) b.X[d] = g.X[d];
  for (int parity = 0; parity < 2; parity++) {
    std::vector<int> list;
    for (int i = 0; i < g.Vh; i++) {
      const int za = i / g.Xh, xh = i - za * g.Xh, zb = za / g.X[1], y = za - zb * g.X[1], t = zb / g.X[2], z = zb - t * g.X[2];
      const int c[4] = {2 * xh + ((y + z + t + parity) & 1), y, z, t};
      bool bd = false;
      for (int d = 0; d < 4; d++) bd = bd || (((mask >> d) & 1) && (c[d] == 0 || c[d] == g.X[d] - 1));
      if (bd) list.push_back(i);
    }
    b.count[parity] = (int)list.size();
    if (!list.empty()) {
      HIP_CHECK(qaMalloc((void **)&b.d_idx[parity], list.size() * sizeof(int)));
      HIP_CHECK(hipMemcpy(b.d_idx[parity], list.data(), list.size() * sizeof(int), hipMemcpyHostToDevice));
    }
  }
  g_blists.push_back(b);
  return g_blists.back();
}
void freeBoundaryLists() {
  for (BoundaryList &b : g_blists)
    for (int p = 0; p < 2; p++) if (b.d_idx[p]) (void)hipFree(b.d_idx[p]);
  g_blists.clear();
}

static hipEvent_t g_evIn = nullptr, g_evHalo = nullptr;

template <typename T, int R, int VARIANT, int GAUX, typename Arg> static void launchExterior(const Arg &arg, hipStream_t s) {
  if (arg.nboundary <= 0) return;
  static int eb = 0;
  if (!eb) { const char *e = getenv("QUDA_AMD_EXT_BLOCK"); eb = e ? atoi(e) : 64; }
  hipLaunchKernelGGL((dslash_kernel<T, R, VARIANT, GAUX, 2>), dim3((arg.nboundary + eb - 1) / eb), dim3(eb), 0, s, arg);
  HIP_CHECK(hipGetLastError());
}

// ---- launch-parameter sweep (tune.h) ----
static const DslashTune *g_sweepTune = nullptr;   // candidate being timed: launchDslash takes its knobs from here and leaves the cache alone
static TuneKey dslashTuneKey(const LatticeGeom &g, int precBytes, int recon, int variant, const DslashParam &p, int pmask) {
  char vol[32], aux[256];
  snprintf(vol, sizeof(vol), "%dx%dx%dx%d", g.X[0], g.X[1], g.X[2], g.X[3]);
  snprintf(aux, sizeof(aux), "prec=%d,recon=%d,mode=%d,xpay=%d,dagger=%d,comm=%d%d%d%d", precBytes, recon, (int)p.mode, p.x ? 1 : 0, p.dagger ? 1 : 0, pmask & 1, (pmask >> 1) & 1, (pmask >> 2) & 1,
           (pmask >> 3) & 1);
  return TuneKey(vol, variant == 2 ? "dslash_kernel<clover>" : (variant == 1 ? "dslash_kernel<twist-first>" : "dslash_kernel"), aux);
}
template <typename T, int R, int VARIANT>
static void launchDslash(ColorSpinorField &out, const ColorSpinorField &in, const GaugeField &U, const DslashParam &p);
// Times the candidate knob settings of one key INTERLEAVED (A B C ... three rounds, minimum per candidate: consecutive runs of one candidate
// share the device's placement / clock state, tools/policy_interleaved.sh) with device events on the compute stream and stores the fastest.
// Every launch is the caller's own application (`out` is simply produced several times); on a grid-decomposed lattice every rank runs the
// same number of launches, so the exchanges pair up.
template <typename T, int R, int VARIANT>
static void sweepDslash(ColorSpinorField &out, const ColorSpinorField &in, const GaugeField &U, const DslashParam &p, const TuneKey &key, int pmask) {
  std::vector<DslashTune> cand;
  const DslashTune &user = dslashTune();
  const LatticeGeom &g = U.geom;
  for (int yg : {1, 2})
    for (int st : {0, 2})
      for (int lk : {0, 2})
        for (int ef : {1, 0}) {
          if (user.ygroups >= 0 && yg != (user.ygroups > 1 ? user.ygroups : 1)) continue;
          if (user.store_aux >= 0 && st != user.store_aux) continue;
          if ((sizeof(T) != 2 || pmask) && lk != 0) continue;                 // the link policy is a knob of the 16-bit unpartitioned kernel only
          if (sizeof(T) == 2 && user.link_aux >= 0 && lk != user.link_aux) continue;
          if (pmask && (st != 0 || yg != 1)) continue;                       // partitioned launch: boundary-first or interior-first order
          if (!pmask && ef != 1) continue;
          DslashTune t;
          t.ygroups = yg; t.store_aux = st; t.link_aux = sizeof(T) == 2 && !pmask ? lk : -1; t.edge_first = pmask ? ef : -1;
          cand.push_back(t);
        }
  if (cand.empty()) return;
  const int reps = 10, rounds = 3;
  std::vector<float> best(cand.size(), 1e30f);
  hipEvent_t e0, e1;
  HIP_CHECK(hipEventCreate(&e0)); HIP_CHECK(hipEventCreate(&e1));
  hipStream_t cs = computeStream();
  for (int round = 0; round < rounds; round++)
    for (size_t c = 0; c < cand.size(); c++) {
      g_sweepTune = &cand[c];
      launchDslash<T, R, VARIANT>(out, in, U, p);   // warm-up of this candidate
      HIP_CHECK(hipEventRecord(e0, cs));
      for (int i = 0; i < reps; i++) launchDslash<T, R, VARIANT>(out, in, U, p);
      HIP_CHECK(hipEventRecord(e1, cs));
      HIP_CHECK(hipEventSynchronize(e1));
      float ms = 0;
      HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
      best[c] = std::min(best[c], ms / reps);
    }
  g_sweepTune = nullptr;
  HIP_CHECK(hipEventDestroy(e0)); HIP_CHECK(hipEventDestroy(e1));
  size_t w = 0;
  for (size_t c = 1; c < cand.size(); c++) if (best[c] < best[w]) w = c;
  TuneParam tp;
  const int plane = g.Xh * g.X[1];
  int bs = 256;
  for (int c : {256, 192, 128, 64}) if (plane % c == 0) { bs = c; break; }
  tp.block[0] = bs; tp.grid[0] = (g.Vh + bs - 1) / bs;
  tp.aux[0] = cand[w].ygroups; tp.aux[1] = cand[w].link_aux < 0 ? 0 : cand[w].link_aux; tp.aux[2] = cand[w].store_aux; tp.aux[3] = cand[w].edge_first < 0 ? 1 : cand[w].edge_first;
  tp.time = 1e-3f * best[w];
  char text[160];
  int n = snprintf(text, sizeof(text), "# %.2f us;", 1e3 * best[w]);
  for (size_t c = 0; c < cand.size() && n < (int)sizeof(text) - 24; c++) n += snprintf(text + n, sizeof(text) - n, " yg%d st%d lk%d ef%d=%.2f", cand[c].ygroups, cand[c].store_aux, cand[c].link_aux, cand[c].edge_first, 1e3 * best[c]);
  tp.comment = text;
  tuneStore(key, tp);
  tuneCountSweep();
  saveTuneCache();
  if (getVerbosity() >= QUDA_SUMMARIZE) printfQuda("Tuned %s %s %s: y groups %d, link policy %d, store policy %d, boundary-first %d (%s)\n", key.volume, key.name, key.aux, tp.aux[0], tp.aux[1], tp.aux[2], tp.aux[3], text);
}

template <typename T, int R, int VARIANT>
static void launchDslash(ColorSpinorField &out, const ColorSpinorField &in, const GaugeField &U, const DslashParam &p) {
  using real = typename Store<T>::real;
  DslashArg<real> arg;
  const LatticeGeom &g = U.geom;
  arg.out = out.V(); arg.outNorm = (float *)out.Norm();
  arg.in = in.V(); arg.inNorm = (const float *)in.Norm();
  arg.x = p.x ? p.x->V() : nullptr; arg.xNorm = p.x ? (const float *)p.x->Norm() : nullptr;
  arg.gauge = (const char *)U.parityBase(p.parity);
  arg.link_bytes = U.link_bytes;
  arg.clA = arg.clAinv = nullptr; arg.clAn = arg.clAinvN = nullptr;
  arg.cl_stride = 0;
  if (VARIANT == 2) {
    arg.clA = p.clover->A(p.parity); arg.clAinv = p.clover->Ainv(p.parity);
    arg.clAn = p.clover->Anorm(p.parity); arg.clAinvN = p.clover->AinvNorm(p.parity);
    arg.cl_stride = p.clover->stride;
  }
  arg.sp_stride = in.Stride(); arg.g_stride = U.stride;
  arg.Vh = g.Vh; arg.Xh = g.Xh; arg.Y = g.X[1]; arg.Z = g.X[2]; arg.T = g.X[3];
  arg.dXh = g.dXh; arg.dY = g.dY; arg.dZ = g.dZ;
  arg.parity = p.parity; arg.mode = p.mode; arg.xpay = p.x ? 1 : 0;
  arg.sfwd = p.dagger ? -1 : 1;
  arg.a = (real)p.a; arg.b = (real)p.b; arg.k = (real)p.k;
  // recon-12: the reconstructed row of boundary t-links carries the (folded) boundary sign; recon-8: u0 of the reconstruction is that sign
  const bool first_t = commGrid().coords[3] == 0, last_t = commGrid().coords[3] == commGrid().dims[3] - 1;
  arg.tsign_fwd = (R != 18 && U.t_boundary == QUDA_ANTI_PERIODIC_T && last_t) ? -1 : 1;
  arg.tsign_bwd = (R != 18 && U.t_boundary == QUDA_ANTI_PERIODIC_T && first_t) ? -1 : 1;
  DslashTune tune = dslashTune();
  {
    // launch-parameter cache (tune.h; reference lib/tune.cpp): where the caller left a knob automatic, the value a sweep found for this
    // (lattice, kernel, precision, reconstruct, epilogue, partition mask) is used; with tuning enabled the first launch of a key runs the sweep
    int pmask = 0;
    for (int d = 0; d < 4; d++) if (commGrid().partitioned(d)) pmask |= 1 << d;
    if (g_sweepTune) {
      if (g_sweepTune->ygroups >= 0) tune.ygroups = g_sweepTune->ygroups;
      if (g_sweepTune->link_aux >= 0) tune.link_aux = g_sweepTune->link_aux;
      if (g_sweepTune->store_aux >= 0) tune.store_aux = g_sweepTune->store_aux;
      if (g_sweepTune->edge_first >= 0) tune.edge_first = g_sweepTune->edge_first;
    } else if (tuningEnabled() || tuneCacheSize() > 0) {
      const TuneKey key = dslashTuneKey(g, (int)sizeof(T), R, VARIANT, p, pmask);
      const TuneParam *tp = tuneLookup(key);
      if (!tp && tuningEnabled() && !(pmask && haloBuffers(g, in.Precision()).verified == 0)) {   // (a partitioned launch is first verified, then tuned)
        sweepDslash<T, R, VARIANT>(out, in, U, p, key, pmask);
        tp = tuneLookup(key);
      }
      if (tp) {
        if (tune.ygroups < 0) tune.ygroups = tp->aux[0];
        if (tune.link_aux < 0) tune.link_aux = tp->aux[1];
        if (tune.store_aux < 0) tune.store_aux = tp->aux[2];
        if (pmask) tune.edge_first = tp->aux[3];
      }
    }
  }
  // block size: the largest of 256 / 192 / 128 / 64 threads that cuts an (x, y) plane into whole blocks (needed by the
  // plane-tiled order), 256 otherwise; QUDA_AMD_DSLASH_BLOCK overrides
  const int plane = g.Xh * g.X[1];
  int bs = tune.block;
  if (bs < 64 || bs > 256 || bs % 64) {
    bs = 256;
    for (int c : {256, 192, 128, 64}) if (plane % c == 0) { bs = c; break; }
  }
  const int nb = (g.Vh + bs - 1) / bs;
  setLastKernel(VARIANT == 2 ? "dslash_kernel (clover)" : "dslash_kernel", g.X, (int)sizeof(T), R, bs);
  arg.nblocks = nb; arg.xcd_q = nb / 8; arg.xcd_r = nb % 8;
  if (!tune.remap) arg.xcd_q = -1;
  {
    const int slice = g.Vh / g.X[3];
    arg.ts = 0; arg.bps = 0;
    if (tune.order > 0 && slice % bs == 0 && g.X[3] % 8 == 0 && nb % 8 == 0) {
      arg.bps = slice / bs;
      arg.ts = tune.order == 1 ? g.X[3] / 8 : tune.order;   // order > 1: explicit slab thickness (must divide T/8)
      if ((g.X[3] / 8) % arg.ts != 0) arg.ts = g.X[3] / 8;
    }
    if (arg.xcd_q < 0) arg.ts = 0;
  }
  // plane-tiled order (see DslashArg): needs whole blocks per plane and an even split of the (z, t) lattice over the 8 XCDs
  // Default since round 2: the 8 XCDs split z (or z x t where z has too few planes), every XCD walks its slab one time slice
  // after the other (tt = 1), all of them in the same time slices at the same time: 48^3 x 96 fp32 0.59 -> 0.61 of the
  // roofline, 32^4 fp32 0.70 -> 0.73 (with nt stores 0.64 / 0.73); tiles in t, t-slabs per XCD and deeper z tiles are all slower.
  arg.tiled = 0;
  if (tune.tiled != 0 && tune.remap && plane % bs == 0 && g.Vh % bs == 0) {
    int nxz = 0;
    for (int c : {8, 4, 2, 1}) if ((tune.nxz <= 0 || tune.nxz == c) && g.X[2] % c == 0 && g.X[3] % (8 / c) == 0) { nxz = c; break; }
    if (nxz) {
      const int Zs = g.X[2] / nxz, Ts = g.X[3] / (8 / nxz);
      int tz = tune.tz > 0 ? tune.tz : Zs, tt = tune.tt > 0 ? tune.tt : 1;
      if (tz > Zs || Zs % tz) tz = Zs;
      if (tt > Ts || Ts % tt) tt = 1;
      arg.tiled = tune.tiled == 1 ? 1 : 2; arg.dPTt = FastDiv((uint32_t)((plane / bs) * tt)); arg.P = plane / bs; arg.nxz = nxz; arg.Zs = Zs; arg.Ts = Ts; arg.tz = tz; arg.tt = tt;
      arg.dNxz = FastDiv((uint32_t)nxz); arg.dPerTile = FastDiv((uint32_t)(arg.P * tz * tt)); arg.dNtz = FastDiv((uint32_t)(Zs / tz));
      arg.dTzTt = FastDiv((uint32_t)(tz * tt)); arg.dTt = FastDiv((uint32_t)tt);
      // y groups (tune.ygroups > 1; automatic: fp64 fields whose three-slice working set per XCD exceeds the 4 MB L2)
      int yg = tune.ygroups;
      if (yg == 0) yg = 1;
      if (yg < 0) {
        yg = 1;
        const size_t slice3 = (size_t)3 * Zs * plane * 24 * sizeof(T);
        if (sizeof(T) == 8) while (yg < arg.P && arg.P % (2 * yg) == 0 && slice3 / yg > ((size_t)5 << 19)) yg *= 2;
      }
      if (yg > 1 && arg.P % yg == 0 && tz == Zs && tt == 1) {
        const int Pg = arg.P / yg;
        arg.tiled = 3;
        arg.dPerTile = FastDiv((uint32_t)(Ts * Zs * Pg)); arg.dTzTt = FastDiv((uint32_t)(Zs * Pg)); arg.dPTt = FastDiv((uint32_t)Pg); arg.dNtz = FastDiv(1u);
      }
    }
  }
  arg.commMask = 0; arg.blist = nullptr; arg.nboundary = 0;
  arg.exSeq = 0; arg.exBuf = 0;
  arg.waitTicks = 0; arg.errWord = nullptr; arg.packBlocks = 0; arg.siteDelay = 0; arg.packPrio = 0; arg.packShare = 0; arg.packFoldBlocks = 0; arg.edgeFirst = 0;
  for (int k = 0; k < 8; k++) arg.waitCount[k] = 0;
  for (int d = 0; d < 4; d++) { arg.ghost[d][0] = arg.ghost[d][1] = nullptr; arg.faceCB[d] = g.faceCB[d]; arg.ghostNormOff[d] = 0; }
  // link/clover stream cache policy: nt for the 16-byte-per-lane formats (measured on 32^4: fp64 4.67 -> 5.2 TB/s, fp32
  // 4.68 -> 5.16 TB/s algorithmic; the read-once links no longer evict the re-used spinors), default for the 8-byte 16-bit format
  // (nt there costs 18 %).
  constexpr int GAUX = sizeof(T) == 2 ? QA_GAUX16 : QA_GAUX;
  int mask = 0;
  for (int d = 0; d < 4; d++) if (commGrid().partitioned(d)) mask |= 1 << d;
  hipStream_t cs = computeStream();
  if (mask == 0) {
    const size_t lds = tune.lds_pad > 0 ? (size_t)tune.lds_pad : 0;   // measurement aid: dynamic LDS only to cap the blocks per CU
    // output stores: nt from 2^18 checkerboard sites up (measured +1...+5 % at 32^4 and 48^3 x 96 in every precision and action,
    // +10 % for 16-bit twisted clover; on the 65k-site sub-lattice of an 8-GPU split it costs fp32 5 %), QUDA_AMD_STORE_AUX / "store_aux" = 0 | 2 overrides
    const bool ntStore = tune.store_aux >= 0 ? tune.store_aux == 2 : g.Vh >= (1 << 18);
    if constexpr (sizeof(T) == 2 && GAUX == 0) {
      // 16-bit links: default policy while one application's working set fits the Infinity Cache (32^4: 205 MB of 256 MiB — the links
      // are re-read from it by the next application, nt costs 13 % there), nt beyond it (48^3 x 96, 2.1 GB: 405 -> 381 us, measured twice)
      const size_t working = (size_t)g.Vh * (size_t)dslashBytesPerSite(in.Precision(), R, p.mode, p.x != nullptr);
      const bool ntLinks = tune.link_aux >= 0 ? tune.link_aux == 2 : working > ((size_t)256 << 20);
      if (ntLinks) {
        if (ntStore) hipLaunchKernelGGL((dslash_kernel<T, R, VARIANT, 2, 0, 2>), dim3(nb), dim3(bs), lds, cs, arg);
        else hipLaunchKernelGGL((dslash_kernel<T, R, VARIANT, 2, 0, 0>), dim3(nb), dim3(bs), lds, cs, arg);
        HIP_CHECK(hipGetLastError());
        return;
      }
    }
    if (ntStore) hipLaunchKernelGGL((dslash_kernel<T, R, VARIANT, GAUX, 0, 2>), dim3(nb), dim3(bs), lds, cs, arg);
    else hipLaunchKernelGGL((dslash_kernel<T, R, VARIANT, GAUX, 0, 0>), dim3(nb), dim3(bs), lds, cs, arg);
    HIP_CHECK(hipGetLastError());
    return;
  }

  // ---- grid-decomposed lattice (reference policies DslashCuda2 / DslashFusedExterior, lib/dslash_policy.cuh) ----
  HaloBuffers &hb = haloBuffers(g, in.Precision());
  const BoundaryList &bl = boundaryList(g, mask);
  PackArg<real> pa;
  pa.in = in.V(); pa.inNorm = (const float *)in.Norm(); pa.sp_stride = in.Stride();
  for (int d = 0; d < 4; d++) pa.X[d] = g.X[d];
  pa.parity_in = 1 - p.parity; pa.sfwd = arg.sfwd; pa.a = arg.a;
  pa.timeline = nullptr; arg.timeline = nullptr;
  for (int k = 0; k < 8; k++) arg.waitCount[k] = 0;
  for (int d = 0; d < 4; d++) pa.llFlag[d] = 0;
  pa.llFormat = 0;
  arg.waitTicks = 0; arg.errWord = nullptr;
  arg.commMask = mask;
  arg.blist = bl.d_idx[p.parity]; arg.nboundary = bl.count[p.parity];

  // First use of the peer-store transport for this precision: the start-up probe has exercised the access patterns, this checks
  // the production kernel itself — the same application once through the staged transport and once through peer stores must
  // agree on every rank, otherwise (wrong data, or a wait that ran out) all ranks drop back to the staged transport for good.
  // Applications with an xpay field before that point go the staged way (the check runs the kernel twice on `out`).
  if (hb.p2p && hb.verified == 0 && !p.x) {
    hb.verified = 3;
    hb.p2p = false;
    launchDslash<T, R, VARIANT>(out, in, U, p);
    HIP_CHECK(hipStreamSynchronize(cs));
    ColorSpinorField ref(out);
    hb.p2p = true;
    launchDslash<T, R, VARIANT>(out, in, U, p);
    HIP_CHECK(hipStreamSynchronize(cs));
    double fail = p2pTakeError() ? 1.0 : 0.0;
    const bool wasGlobal = blas::globalReduction();
    blas::setGlobalReduction(false);   // rank-local sums; the verdict is shared through the host collective below
    const double n2 = blas::norm2(out), d2 = blas::xmyNorm(out, ref);
    blas::setGlobalReduction(wasGlobal);
    const double tol = sizeof(T) == 8 ? 1e-10 : (sizeof(T) == 4 ? 1e-4 : 3e-2);
    if (!(d2 <= tol * tol * n2)) fail = 1.0;
    if (getenv("QUDA_AMD_P2P_VERIFY_FAIL")) fail = 1.0;   // test hook: exercise the fall-back
    comm_allreduce(&fail, 1);
    if (fail != 0.0) {
      if (commGrid().rank == 0) warningQuda("peer-store halo disagrees with the staged transport on its first use: staying with RCCL send/recv");
      p2pDisable();
      for (HaloBuffers &h : g_halo) { h.p2p = false; h.verified = 0; }
      launchDslash<T, R, VARIANT>(out, in, U, p);
      return;
    }
    hb.verified = 1;
    return;
  }
  if (hb.p2p && (hb.verified == 1 || hb.verified == 3)) {
    // peer-store transport: ONE launch on ONE stream, [pack blocks | every site].  The pack blocks store the faces straight into
    // the neighbours' ghost zones with the exchange's flag inside every word pair; a boundary site does its local hops first
    // and then polls exactly the ghost words it needs (ghost_hop).  Zones are double-buffered by the parity of the exchange
    // counter: a neighbour can be at most one exchange ahead (to finish exchange k+1 it needs OUR face k+1, which is packed
    // after our kernel k), and the flag — the use count of the (dimension, buffer) zone — tells this exchange's words from
    // the ones two exchanges old.
    const unsigned seq = ++hb.seq;
    const int buf = seq & 1;
    arg.exSeq = seq; arg.exBuf = buf;
    p2pStats()[0]++;
    int nt = 0;
    for (int d = 0; d < 4; d++) {
      pa.faceCB[d] = g.faceCB[d]; pa.normOff[d] = 0;
      for (int dir = 0; dir < 2; dir++) {
        pa.send[d][dir] = hb.peerGhost[d][dir][buf];
        pa.start[2 * d + dir] = nt;
        if ((mask >> d) & 1) nt += g.faceCB[d];
      }
      if ((mask >> d) & 1) {
        arg.ghost[d][0] = hb.ghostBuf[d][0][buf]; arg.ghost[d][1] = hb.ghostBuf[d][1][buf];
        unsigned flag = ++hb.uses[d][buf];
        if (flag == 0) flag = ++hb.uses[d][buf];   // 0 is the value of a never-written word
        pa.llFlag[d] = flag;
        arg.waitCount[2 * d] = arg.waitCount[2 * d + 1] = flag;
      }
    }
    pa.start[8] = nt;
    arg.waitTicks = p2pTimeoutTicks(); arg.errWord = p2pErrorWord();
    arg.siteDelay = tune.site_delay; arg.packPrio = tune.pack_prio; arg.edgeFirst = p2pDeviceShared() ? 0 : tune.edge_first;   // ranks sharing a device: interior blocks first (p2p.h)
    // Full pack blocks (256 face sites each) in front of the grid.  They share CUs with site blocks, and the placement statistics of
    // the timeline show what that costs on the 8-GPU sub-lattice: a site block next to a pack block ends 4.3 us later than one
    // that has its CU to itself.  Spreading the packing thinly (one 96-thread pack block on EVERY CU) is far worse — 46 us
    // instead of 29, every site block ends late — so the system-scope write-through stores of a pack wave appear to hold up the
    // memory pipeline of the whole CU, and fewer CUs doing all of it is the better trade.
    // rounded up to a multiple of 8: blockIdx.x and the site-block number then agree mod 8, i.e. on the XCD (surplus pack blocks leave at once)
    arg.packBlocks = ((nt + bs - 1) / bs + 7) / 8 * 8; arg.packChunk = bs;
    // Folded packing where the face sites fit the threads of the first blocks of the grid (one wave of blocks: they are running
    // before any block that could wait for them, so the faces always get out): no pack blocks at all, see stencil_site.
    // Measured on the 8-GPU sub-lattice 32 x 16 x 16 x 16 (y, z, t partitioned, self-neighbour emulation): the pack blocks need
    // 11 us (up to 18) next to the stencil traffic against 3 us on an idle device (tools/ubench_ll_store.hip) and the site block
    // that shares their CU ends 4-6 us late.  Off by default and compiled out (QA_P2P_FOLD above); with -DQA_P2P_FOLD=1, QUDA_AMD_P2P_FOLD=1 selects it.
    {
      static int foldEnv = -1;
      if (foldEnv < 0) { const char *e = getenv("QUDA_AMD_P2P_FOLD"); foldEnv = e ? atoi(e) : 0; }
      pa.llFormat = haloWireFormat();
      const int fold = tune.p2p_fold >= 0 ? tune.p2p_fold : foldEnv;
      const int nbp = nb < 256 ? nb : 256;
      const int share = (nt + nbp - 1) / nbp;
      arg.packShare = 0; arg.packFoldBlocks = 0;
      if (QA_P2P_FOLD && fold && share <= bs) { arg.packShare = share; arg.packFoldBlocks = nbp; arg.packBlocks = 0; }
    }
    static unsigned long long *tl = nullptr;
    static int tlmode = -1;
    if (tlmode < 0) { const char *e = getenv("QUDA_AMD_TIMELINE"); tlmode = e ? atoi(e) : 0; if (tlmode) HIP_CHECK(hipHostMalloc((void **)&tl, 16384 * sizeof(unsigned long long), hipHostMallocMapped)); }
    if (tlmode) {
      static int calls = 0;
      if (++calls == 60 && arg.packBlocks + nb <= 1024) {
        HIP_CHECK(hipStreamSynchronize(cs));
        memset(tl, 0, 16384 * sizeof(unsigned long long));
        pa.timeline = tl; arg.timeline = tl; arg.pack = pa;
        hipLaunchKernelGGL((dslash_kernel<T, R, VARIANT, GAUX, 3>), dim3(arg.packBlocks + nb), dim3(bs), 0, cs, arg);
        HIP_CHECK(hipStreamSynchronize(cs));
        unsigned long long t0 = ~0ull;
        for (int i = 0; i < 16384; i++) if ((i < 3072 || i >= 4096) && tl[i] && tl[i] < t0) t0 = tl[i];   // 3072..4095 hold placement words
        auto stat = [&](int off, int n, const char *name) {
          double mn = 1e30, mx = 0, sum = 0; int c = 0;
          for (int i = 0; i < n; i++) if (tl[off + i]) { const double v = (tl[off + i] - t0) * 0.01; mn = v < mn ? v : mn; mx = v > mx ? v : mx; sum += v; c++; }
          if (c) printfQuda("timeline %-22s n=%4d  min %6.2f  mean %6.2f  max %6.2f us\n", name, c, mn, sum / c, mx);
        };
        stat(0, 1024, "pack block start"); stat(13312, 1024, "pack loads returned"); stat(1024, 1024, "pack block end"); stat(2048, 1024, "stencil block start");
        stat(4096, 4096, "boundary wave ghost beg"); stat(8192, 4096, "boundary wave ghost end"); stat(12288, 1024, "stencil block end");
        {   // placement: how many site / pack blocks share a CU, and when the site blocks of such CUs finish
          std::map<unsigned, std::pair<int, int>> cu;   // key -> (site blocks, pack blocks)
          const int npk = arg.packBlocks, ntot = arg.packBlocks + nb;
          for (int i = 0; i < ntot && i < 1024; i++) if (tl[3072 + i]) { auto &c = cu[(unsigned)tl[3072 + i]]; if (i < npk) c.second++; else c.first++; }
          double sum[4][2] = {}, mx[4][2] = {}; int cnt[4][2] = {};
          for (int i = npk; i < ntot && i < 1024; i++) if (tl[3072 + i] && tl[12288 + i]) {
            const auto &c = cu[(unsigned)tl[3072 + i]];
            const int a = c.first > 3 ? 3 : c.first, b = c.second > 0 ? 1 : 0;
            const double e = (tl[12288 + i] - t0) * 0.01;
            sum[a][b] += e; mx[a][b] = e > mx[a][b] ? e : mx[a][b]; cnt[a][b]++;
          }
          printfQuda("timeline placement: %zu CUs in use\n", cu.size());
          for (int a = 1; a < 4; a++) for (int b = 0; b < 2; b++)
            if (cnt[a][b]) printfQuda("timeline   site blocks on a CU with %d site block(s)%s: n=%3d  mean end %6.2f  max end %6.2f us\n", a, b ? " + pack block(s)" : "", cnt[a][b], sum[a][b] / cnt[a][b], mx[a][b]);
        }
        for (int x = 0; x < 8; x++) {   // per XCD (blocks are dealt round-robin): start / end of its stencil blocks
          double sb = 0, se = 0, mx = 0; int c = 0;
          for (int i = x; i < 1024; i += 8) if (tl[12288 + i]) { sb += (tl[2048 + i] - t0) * 0.01; const double e = (tl[12288 + i] - t0) * 0.01; se += e; mx = e > mx ? e : mx; c++; }
          if (c) printfQuda("timeline xcd %d: %3d stencil blocks, mean start %6.2f, mean end %6.2f, max end %6.2f us\n", x, c, sb / c, se / c, mx);
        }
        return;
      }
    }
    pa.timeline = nullptr; arg.timeline = nullptr;
    arg.pack = pa;
    hipLaunchKernelGGL((dslash_kernel<T, R, VARIANT, GAUX, 3>), dim3(arg.packBlocks + nb), dim3(bs), 0, cs, arg);
    HIP_CHECK(hipGetLastError());
    return;
  }

  // staged transport: pack + ONE grouped RCCL send/recv on the comms stream, interior stencil on the compute stream meanwhile,
  // then the exterior pass over the boundary-site list
  if (!g_evIn) { HIP_CHECK(hipEventCreateWithFlags(&g_evIn, hipEventDisableTiming)); HIP_CHECK(hipEventCreateWithFlags(&g_evHalo, hipEventDisableTiming)); }
  p2pStats()[1]++;
  hipStream_t ms = commStream();
  HIP_CHECK(hipEventRecord(g_evIn, cs));            // `in` is complete and the previous exterior pass has released the ghost zone
  HIP_CHECK(hipStreamWaitEvent(ms, g_evIn, 0));
  int nt = 0;
  std::vector<HaloMsg> msgs;
  for (int d = 0; d < 4; d++) {
    pa.faceCB[d] = g.faceCB[d]; pa.normOff[d] = (int)hb.norm_offset[d];
    for (int dir = 0; dir < 2; dir++) {
      pa.send[d][dir] = hb.send[d][dir];
      pa.start[2 * d + dir] = nt;
      if ((mask >> d) & 1) nt += g.faceCB[d];
    }
    if ((mask >> d) & 1) {
      // sent forward -> arrives in the +d neighbour's "from behind" zone; the matching receive fills ours from our -d neighbour
      msgs.push_back({d, +1, hb.send[d][1], hb.ghost[d][0], hb.face_bytes[d]});
      msgs.push_back({d, -1, hb.send[d][0], hb.ghost[d][1], hb.face_bytes[d]});
      arg.ghost[d][0] = hb.ghost[d][0]; arg.ghost[d][1] = hb.ghost[d][1];
      arg.ghostNormOff[d] = (int)hb.norm_offset[d];
    }
  }
  pa.start[8] = nt;
  hipLaunchKernelGGL((pack_kernel<T, VARIANT == 1>), dim3((nt + 255) / 256), dim3(256), 0, ms, pa);
  HIP_CHECK(hipGetLastError());
  commExchange(msgs, ms);
  HIP_CHECK(hipEventRecord(g_evHalo, ms));

  hipLaunchKernelGGL((dslash_kernel<T, R, VARIANT, GAUX, 1>), dim3(nb), dim3(bs), 0, cs, arg);   // interior
  HIP_CHECK(hipGetLastError());
  HIP_CHECK(hipStreamWaitEvent(cs, g_evHalo, 0));
  launchExterior<T, R, VARIANT, GAUX>(arg, cs);
}

template <typename T> static void dispatchRecon(ColorSpinorField &out, const ColorSpinorField &in, const GaugeField &U, const DslashParam &p) {
  const bool clover = p.mode == DSLASH_CLOVER_TWIST_INV || p.mode == DSLASH_CLOVER_TWIST_XPAY;
  if (clover && !p.clover) errorQuda("clover field required for dslash mode %d", p.mode);
  const int variant = clover ? 2 : (p.mode == DSLASH_TWIST_INV_DSLASH ? 1 : 0);
  if (U.reconstruct == QUDA_RECONSTRUCT_NO) {
    if (variant == 2) launchDslash<T, 18, 2>(out, in, U, p);
    else if (variant == 1) launchDslash<T, 18, 1>(out, in, U, p);
    else launchDslash<T, 18, 0>(out, in, U, p);
  } else if (U.reconstruct == QUDA_RECONSTRUCT_12) {
    if (variant == 2) launchDslash<T, 12, 2>(out, in, U, p);
    else if (variant == 1) launchDslash<T, 12, 1>(out, in, U, p);
    else launchDslash<T, 12, 0>(out, in, U, p);
  } else if (U.reconstruct == QUDA_RECONSTRUCT_8) {   // reference: the RECONSTRUCT_8 variants of every Wilson-type stencil, lib/dslash_quda.cuh:16-52
    if (variant == 2) launchDslash<T, 8, 2>(out, in, U, p);
    else if (variant == 1) launchDslash<T, 8, 1>(out, in, U, p);
    else launchDslash<T, 8, 0>(out, in, U, p);
  } else {
    errorQuda("reconstruct %d not supported (18, 12 and 8)", U.reconstruct);
  }
}

void applyDslash(ColorSpinorField &out, const ColorSpinorField &in, const GaugeField &U, const DslashParam &p) {
  if (out.Location() != QUDA_CUDA_FIELD_LOCATION || in.Location() != QUDA_CUDA_FIELD_LOCATION) errorQuda("device fields required");
  if (in.SiteSubset() != QUDA_PARITY_SITE_SUBSET || out.SiteSubset() != QUDA_PARITY_SITE_SUBSET) errorQuda("parity fields required");
  if (in.Precision() != out.Precision() || in.Precision() != U.precision) errorQuda("precision mismatch: spinor %d/%d gauge %d", in.Precision(), out.Precision(), U.precision);
  if (p.x && p.x->Precision() != in.Precision()) errorQuda("xpay precision mismatch");
  if (in.VolumeCB() != U.geom.Vh || out.VolumeCB() != U.geom.Vh) errorQuda("volume mismatch: spinor %d gauge %d", in.VolumeCB(), U.geom.Vh);
  if (in.V() == out.V()) errorQuda("in and out must not alias");
  if (in.Nspin() != 4 || in.Ncolor() != 3) errorQuda("fine-grid dslash needs nSpin=4 nColor=3");
  if (p.clover && p.clover->precision != in.Precision()) errorQuda("clover precision mismatch");
  if (g_acctOn) {   // one dslash_kernel launch on an unpartitioned lattice (the case the profile tool measures)
    char tag[48];
    snprintf(tag, sizeof(tag), "level 0 prec %d mode %d%s", (int)in.Precision(), (int)p.mode, p.x ? " xpay" : "");
    acct("dslash_kernel", (double)dslashBytesPerSite(in.Precision(), (int)U.reconstruct, p.mode, p.x != nullptr) * in.VolumeCB(), tag);
  }
  switch (in.Precision()) {
    case QUDA_DOUBLE_PRECISION: dispatchRecon<double>(out, in, U, p); break;
    case QUDA_SINGLE_PRECISION: dispatchRecon<float>(out, in, U, p); break;
    case QUDA_HALF_PRECISION: dispatchRecon<short>(out, in, U, p); break;
    default: errorQuda("bad precision %d", in.Precision());
  }
}

// full (unprojected) spinors of the face x_d = face_coord of the parity_in checkerboard -> planar block of stride faceCB
template <typename T>
__global__ void __launch_bounds__(256) face_full_pack_kernel(void *dst, const void *in, const float *inNorm, int sp_stride, int X0, int X1, int X2, int X3,
                                                             int d, int face_coord, int parity_in, int faceCB) {
  using real = typename Store<T>::real;
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= faceCB) return;
  const int X[4] = {X0, X1, X2, X3};
  int c[4], L[3], o[3], n = 0;
  for (int k = 0; k < 4; k++) if (k != d) { L[n] = X[k]; o[n] = k; n++; }
  int l = 2 * f;
  const int c0 = l % L[0]; l /= L[0];
  const int c1 = l % L[1]; const int c2 = l / L[1];
  c[d] = face_coord;
  c[o[0]] = c0; c[o[1]] = c1; c[o[2]] = c2;
  c[o[0]] += (parity_in + c[0] + c[1] + c[2] + c[3]) & 1;
  const int idx = (((c[3] * X[2] + c[2]) * X[1] + c[1]) * X[0] + c[0]) >> 1;
  real psi[24];
  Planar<T, 24>::load(psi, in, sp_stride, idx, inNorm, idx);
  Planar<T, 24>::store(psi, dst, faceCB, f, nullptr, f);
}

static char *g_ffSend = nullptr, *g_ffGhost = nullptr;
static size_t g_ffBytes = 0;
void freeFullFaceBuffers() {
  if (g_ffSend) (void)hipFree(g_ffSend);
  if (g_ffGhost) (void)hipFree(g_ffGhost);
  g_ffSend = g_ffGhost = nullptr; g_ffBytes = 0;
}

template <typename T, int R> static void launchHopDir(ColorSpinorField &out, const ColorSpinorField &in, const GaugeField &U, int parity, int dir, double coef,
                                                      bool plain = false, const ColorSpinorField *x = nullptr, double xcoef = 0) {
  using real = typename Store<T>::real;
  DslashArg<real> arg;
  memset(&arg, 0, sizeof(arg));
  const LatticeGeom &g = U.geom;
  arg.out = out.V(); arg.outNorm = (float *)out.Norm();
  arg.in = in.V(); arg.inNorm = (const float *)in.Norm();
  arg.gauge = (const char *)U.parityBase(parity);
  arg.link_bytes = U.link_bytes;
  arg.sp_stride = in.Stride(); arg.g_stride = U.stride;
  arg.Vh = g.Vh; arg.Xh = g.Xh; arg.Y = g.X[1]; arg.Z = g.X[2]; arg.T = g.X[3];
  arg.dXh = g.dXh; arg.dY = g.dY; arg.dZ = g.dZ;
  arg.parity = parity; arg.sfwd = 1;
  if (x) { arg.x = x->V(); arg.xNorm = (const float *)x->Norm(); arg.xpay = 1; arg.k = (real)xcoef; }
  const bool first_t = commGrid().coords[3] == 0, last_t = commGrid().coords[3] == commGrid().dims[3] - 1;
  arg.tsign_fwd = (R != 18 && U.t_boundary == QUDA_ANTI_PERIODIC_T && last_t) ? -1 : 1;
  arg.tsign_bwd = (R != 18 && U.t_boundary == QUDA_ANTI_PERIODIC_T && first_t) ? -1 : 1;
  const int mu = dir >> 1;
  const void *ghost = nullptr;
  if (commGrid().partitioned(mu)) {
    // a forward hop needs the x_mu = 0 face of the +mu neighbour (it sends it backward), a backward hop the x_mu = L-1 face
    // of the -mu neighbour; setup-time path, so one blocking exchange on the compute stream
    const size_t bytes = (size_t)g.faceCB[mu] * 24 * sizeof(real);
    if (bytes > g_ffBytes) {
      freeFullFaceBuffers();
      HIP_CHECK(qaMalloc((void **)&g_ffSend, bytes));
      HIP_CHECK(qaMalloc((void **)&g_ffGhost, bytes));
      g_ffBytes = bytes;
    }
    const bool fwd = !(dir & 1);
    hipLaunchKernelGGL((face_full_pack_kernel<T>), dim3((g.faceCB[mu] + 255) / 256), dim3(256), 0, computeStream(), (void *)g_ffSend, in.V(), (const float *)in.Norm(),
                       in.Stride(), g.X[0], g.X[1], g.X[2], g.X[3], mu, fwd ? 0 : g.X[mu] - 1, 1 - parity, g.faceCB[mu]);
    HIP_CHECK(hipGetLastError());
    std::vector<HaloMsg> msgs;
    msgs.push_back({mu, fwd ? -1 : +1, g_ffSend, g_ffGhost, bytes});
    commExchange(msgs, computeStream());
    ghost = g_ffGhost;
  }
  hipLaunchKernelGGL((hop_dir_kernel<T, R>), dim3((g.Vh + 255) / 256), dim3(256), 0, computeStream(), arg, dir, (real)coef, ghost, g.faceCB[mu], plain ? 1 : 0);
  HIP_CHECK(hipGetLastError());
}

// out(parity) = xcoef x + coef U_dir psi(x + dhat(dir)) with NO spin projection (x may be nullptr or alias out); ghost-aware
void applyCovariantShift(ColorSpinorField &out, const ColorSpinorField &in, const GaugeField &U, int parity, int dir, double coef,
                         const ColorSpinorField *x, double xcoef) {
  if (in.Precision() != out.Precision() || in.Precision() != U.precision) errorQuda("precision mismatch");
  if (in.VolumeCB() != U.geom.Vh || out.VolumeCB() != U.geom.Vh) errorQuda("volume mismatch");
  if (x && (x->Precision() != out.Precision() || x->Stride() != out.Stride())) errorQuda("accumulation field does not match the output");
  if (in.Stride() != out.Stride()) errorQuda("stride mismatch");
  const int rr = (int)U.reconstruct;
  switch (in.Precision()) {
    case QUDA_DOUBLE_PRECISION: rr == 12 ? launchHopDir<double, 12>(out, in, U, parity, dir, coef, true, x, xcoef) : (rr == 8 ? launchHopDir<double, 8>(out, in, U, parity, dir, coef, true, x, xcoef) : launchHopDir<double, 18>(out, in, U, parity, dir, coef, true, x, xcoef)); break;
    case QUDA_SINGLE_PRECISION: rr == 12 ? launchHopDir<float, 12>(out, in, U, parity, dir, coef, true, x, xcoef) : (rr == 8 ? launchHopDir<float, 8>(out, in, U, parity, dir, coef, true, x, xcoef) : launchHopDir<float, 18>(out, in, U, parity, dir, coef, true, x, xcoef)); break;
    default: errorQuda("covariant shift: fp64/fp32 only");
  }
}

// ---- direct Galerkin construction, step 1 ("UV", reference ComputeUV lib/coarse_op.cuh:59-125): UV(x)[s, c; v] = coef U_mu(x) V(x + mu)[s, c; v]
// for ALL columns v of the transfer matrix at once — the link is read once per site instead of once per probe.  No spin projector here:
// (1 -+ gamma_mu) only permutes the spin rows of UV and multiplies them by +-1 / +-i, which galerkin_vuv_kernel does while it builds
// its operands (spin_partner_phase below) — half the bytes of a projected, chirality-split product.  V, UV in the aggregate-major order of
// the transfer operator, [aggregate][spin-colour][vector pair][site in aggregate] float4.  One work-group per aggregate, one thread per
// site of it (stores of a wave are 1 KiB contiguous), loop over the vector pairs.
__global__ void __launch_bounds__(256) galerkin_uv_kernel(const DslashArg<float> arg, const char *gaugeEven, const char *gaugeOdd, int dir, float coef, const float4 *V, float4 *UV,
                                                          const int *block_to_fine, const int *fine_to_block, int nvp, int aggOffset, int classMajor, int recon, float tsignFwd) {
  constexpr int BV = 256;
  const int A = blockIdx.x + aggOffset, Aloc = blockIdx.x, b = threadIdx.x;   // UV holds the aggregates [aggOffset, aggOffset + gridDim) only
  const int f = block_to_fine[(size_t)A * BV + b];
  const int parity = f >= arg.Vh, idx = f - parity * arg.Vh;
  const uint32_t za = arg.dXh.div((uint32_t)idx);
  const int xh = idx - (int)za * arg.Xh;
  const uint32_t zb = arg.dY.div(za);
  const int y = (int)za - (int)zb * arg.Y;
  const int tt = (int)arg.dZ.div(zb);
  const int z = (int)zb - tt * arg.Z;
  const int xodd = (y + z + tt + parity) & 1;
  const int Xh = arg.Xh, sy = Xh, sz = Xh * arg.Y, st = Xh * arg.Y * arg.Z;
  int nbr;
  switch (dir) {
    case 0: nbr = xodd ? (xh == Xh - 1 ? idx - (Xh - 1) : idx + 1) : idx; break;
    case 1: nbr = xodd ? idx : (xh == 0 ? idx + (Xh - 1) : idx - 1); break;
    case 2: nbr = y == arg.Y - 1 ? idx - (arg.Y - 1) * sy : idx + sy; break;
    case 3: nbr = y == 0 ? idx + (arg.Y - 1) * sy : idx - sy; break;
    case 4: nbr = z == arg.Z - 1 ? idx - (arg.Z - 1) * sz : idx + sz; break;
    case 5: nbr = z == 0 ? idx + (arg.Z - 1) * sz : idx - sz; break;
    case 6: nbr = tt == arg.T - 1 ? idx - (arg.T - 1) * st : idx + st; break;
    default: nbr = tt == 0 ? idx + (arg.T - 1) * st : idx - st; break;
  }
  const int posN = fine_to_block[(1 - parity) * arg.Vh + nbr];
  const int AN = posN / BV, bN = posN - AN * BV;
  // where this site's UV goes inside a (spin-colour, vector pair) row: classMajor — by its block coordinate along mu first (the class of sites one
  // wave of galerkin_vuv_kernel owns), then the other three coordinates: that wave then reads 64 consecutive entries of every row instead of
  // 16-byte pieces scattered over the whole row
  int bOut = b;
  if (classMajor) {
    const int c4[4] = {(2 * xh + xodd) & 3, y & 3, z & 3, tt & 3};
    const int mu = dir >> 1;
    int pos = 0, sh = 0;
#pragma unroll
    for (int d = 0; d < 4; d++) if (d != mu) { pos |= c4[d] << sh; sh += 2; }
    bOut = c4[mu] * 64 + pos;
  }
  float U[18];
  if (recon == 12) Link<float, 12>::load(U, (parity ? gaugeOdd : gaugeEven) + (size_t)dir * arg.link_bytes, arg.g_stride, idx, (dir == 6 && tt == arg.T - 1) ? tsignFwd : 1.f);
  else Link<float, 18>::load(U, (parity ? gaugeOdd : gaugeEven) + (size_t)dir * arg.link_bytes, arg.g_stride, idx, 1.f);
#pragma unroll
  for (int k = 0; k < 18; k++) U[k] *= coef;
  for (int vp = 0; vp < nvp; vp++) {
    float psi[2][24];
#pragma unroll
    for (int k = 0; k < 12; k++) {
      const float4 v = V[(((size_t)AN * 12 + k) * nvp + vp) * BV + bN];
      psi[0][2 * k] = v.x; psi[0][2 * k + 1] = v.y; psi[1][2 * k] = v.z; psi[1][2 * k + 1] = v.w;
    }
    float out[2][24];
#pragma unroll
    for (int vec = 0; vec < 2; vec++)
#pragma unroll
      for (int sp = 0; sp < 4; sp++) su3_mv(out[vec] + 6 * sp, U, psi[vec] + 6 * sp);
#pragma unroll
    for (int k = 0; k < 12; k++) UV[(((size_t)Aloc * 12 + k) * nvp + vp) * BV + bOut] = make_float4(out[0][2 * k], out[0][2 * k + 1], out[1][2 * k], out[1][2 * k + 1]);
  }
}
// fp32 recon-18 links (boundary condition inside the stored links), 4^4 aggregates, unpartitioned lattice (the neighbour's V would live on another rank)
void galerkinUV(float *UVout, const float *V, const GaugeField &U, int dir, double coef, const int *block_to_fine, const int *fine_to_block, int aggOffset, int nAgg, int blockVol, int nvec, bool classMajor) {
  if (U.precision != QUDA_SINGLE_PRECISION || (U.reconstruct != QUDA_RECONSTRUCT_NO && U.reconstruct != QUDA_RECONSTRUCT_12)) errorQuda("direct Galerkin construction: fp32 links (18 or 12 reals)");
  if (dir & 1) errorQuda("direct Galerkin construction: forward directions only");
  if (blockVol != 256) errorQuda("direct Galerkin construction: 4^4 aggregates");
  const LatticeGeom &g = U.geom;
  DslashArg<float> arg;
  memset(&arg, 0, sizeof(arg));
  arg.link_bytes = U.link_bytes; arg.g_stride = U.stride;
  arg.Vh = g.Vh; arg.Xh = g.Xh; arg.Y = g.X[1]; arg.Z = g.X[2]; arg.T = g.X[3];
  arg.dXh = g.dXh; arg.dY = g.dY; arg.dZ = g.dZ;
  hipLaunchKernelGGL(galerkin_uv_kernel, dim3(nAgg), dim3(256), 0, computeStream(), arg, (const char *)U.parityBase(0), (const char *)U.parityBase(1), dir, (float)coef, (const float4 *)V,
                     (float4 *)UVout, block_to_fine, fine_to_block, nvec / 2, aggOffset, classMajor ? 1 : 0, (int)U.reconstruct,
                     (U.reconstruct == QUDA_RECONSTRUCT_12 && U.t_boundary == QUDA_ANTI_PERIODIC_T && commGrid().coords[3] == commGrid().dims[3] - 1) ? -1.f : 1.f);
  HIP_CHECK(hipGetLastError());
}

// the same for the site-diagonal term of the twisted-clover operator: L(x)[s, c; v] = (A_chi(s)(x) + i a s_chi) V(x)[s, c; v] — chirality-diagonal, so
// galerkin_vuv_kernel runs it in its "local" mode (no cross-chirality columns, every site belongs to the local matrix)
__global__ void __launch_bounds__(256) galerkin_local_uv_kernel(const float4 *V, float4 *L, const void *clEven, const void *clOdd, int cl_stride, int Vh, float a,
                                                                const int *block_to_fine, int nvp, int aggOffset) {
  constexpr int BV = 256;
  const int A = blockIdx.x + aggOffset, Aloc = blockIdx.x, b = threadIdx.x;
  const int f = block_to_fine[(size_t)A * BV + b];
  const int parity = f >= Vh, idx = f - parity * Vh;
  const void *clA = parity ? clOdd : clEven;
  float C[2][36];
#pragma unroll
  for (int chi = 0; chi < 2; chi++) Planar<float, 36>::load(C[chi], (const char *)clA + (size_t)chi * 36 * sizeof(float) * cl_stride, cl_stride, idx, nullptr, 0);
  for (int vp = 0; vp < nvp; vp++) {
    float psi[2][24], o[2][24];
#pragma unroll
    for (int k = 0; k < 12; k++) {
      const float4 v = V[(((size_t)A * 12 + k) * nvp + vp) * BV + b];
      psi[0][2 * k] = v.x; psi[0][2 * k + 1] = v.y; psi[1][2 * k] = v.z; psi[1][2 * k + 1] = v.w;
    }
#pragma unroll
    for (int vec = 0; vec < 2; vec++)
#pragma unroll
      for (int chi = 0; chi < 2; chi++) {
        const float *v = psi[vec] + 12 * chi;
        float *r = o[vec] + 12 * chi;
        clover_block_mv(r, C[chi], v);
        const float sa = chi ? -a : a;
#pragma unroll
        for (int k = 0; k < 6; k++) { r[2 * k] -= sa * v[2 * k + 1]; r[2 * k + 1] += sa * v[2 * k]; }
      }
#pragma unroll
    for (int k = 0; k < 12; k++) L[(((size_t)Aloc * 12 + k) * nvp + vp) * BV + b] = make_float4(o[0][2 * k], o[0][2 * k + 1], o[1][2 * k], o[1][2 * k + 1]);
  }
}
void galerkinLocalUV(float *Lout, const float *V, const CloverField &C, double a, const int *block_to_fine, int aggOffset, int nAgg, int blockVol, int nvec) {
  if (C.precision != QUDA_SINGLE_PRECISION) errorQuda("direct Galerkin construction: fp32 clover field");
  if (blockVol != 256) errorQuda("direct Galerkin construction: 4^4 aggregates");
  hipLaunchKernelGGL(galerkin_local_uv_kernel, dim3(nAgg), dim3(256), 0, computeStream(), (const float4 *)V, (float4 *)Lout, C.A(0), C.A(1), C.stride, C.geom.Vh, (float)a, block_to_fine, nvec / 2, aggOffset);
  HIP_CHECK(hipGetLastError());
}

void applyHopDir(ColorSpinorField &out, const ColorSpinorField &in, const GaugeField &U, int parity, int dir, double coef) {
  if (in.Precision() != out.Precision() || in.Precision() != U.precision) errorQuda("precision mismatch");
  if (in.VolumeCB() != U.geom.Vh) errorQuda("volume mismatch");
  const int rr = (int)U.reconstruct;
  switch (in.Precision()) {
    case QUDA_DOUBLE_PRECISION: rr == 12 ? launchHopDir<double, 12>(out, in, U, parity, dir, coef) : (rr == 8 ? launchHopDir<double, 8>(out, in, U, parity, dir, coef) : launchHopDir<double, 18>(out, in, U, parity, dir, coef)); break;
    case QUDA_SINGLE_PRECISION: rr == 12 ? launchHopDir<float, 12>(out, in, U, parity, dir, coef) : (rr == 8 ? launchHopDir<float, 8>(out, in, U, parity, dir, coef) : launchHopDir<float, 18>(out, in, U, parity, dir, coef)); break;
    default: errorQuda("single-direction hop: fp64/fp32 only");
  }
}

// ---- neighbour shift of a 24-real fp64 planar site field: out(x) = in(x + dhat(dir)), ghost-aware like the single-direction
// hop (full faces exchanged through commExchange).  Used by the grid-decomposed clover construction, which moves 3x3 matrix
// fields (18 of the 24 reals) between neighbouring sites; setup-time only. ----
__global__ void __launch_bounds__(256) shift_kernel(double *out, const double *in, int stride, int Vh, int Xh, int Y, int Z, int T, int parity, int dir,
                                                    const void *ghost, int ghostFaceCB) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= Vh) return;
  int l = idx;
  const int xh = l % Xh; l /= Xh;
  const int y = l % Y; l /= Y;
  const int z = l % Z, t = l / Z;
  const int xodd = (y + z + t + parity) & 1, xf = 2 * xh + xodd, X0 = 2 * Xh;
  const int sy = Xh, sz = Xh * Y, st = Xh * Y * Z;
  int nbr, f;
  bool cross;
  switch (dir) {
    case 0: nbr = xodd ? (xh == Xh - 1 ? idx - (Xh - 1) : idx + 1) : idx; cross = xf == X0 - 1; f = (y + Y * (z + Z * t)) >> 1; break;
    case 1: nbr = xodd ? idx : (xh == 0 ? idx + (Xh - 1) : idx - 1); cross = xf == 0; f = (y + Y * (z + Z * t)) >> 1; break;
    case 2: nbr = y == Y - 1 ? idx - (Y - 1) * sy : idx + sy; cross = y == Y - 1; f = (xf + X0 * (z + Z * t)) >> 1; break;
    case 3: nbr = y == 0 ? idx + (Y - 1) * sy : idx - sy; cross = y == 0; f = (xf + X0 * (z + Z * t)) >> 1; break;
    case 4: nbr = z == Z - 1 ? idx - (Z - 1) * sz : idx + sz; cross = z == Z - 1; f = (xf + X0 * (y + Y * t)) >> 1; break;
    case 5: nbr = z == 0 ? idx + (Z - 1) * sz : idx - sz; cross = z == 0; f = (xf + X0 * (y + Y * t)) >> 1; break;
    case 6: nbr = t == T - 1 ? idx - (T - 1) * st : idx + st; cross = t == T - 1; f = (xf + X0 * (y + Y * z)) >> 1; break;
    default: nbr = t == 0 ? idx + (T - 1) * st : idx - st; cross = t == 0; f = (xf + X0 * (y + Y * z)) >> 1; break;
  }
  double v[24];
  if (ghost && cross) Planar<double, 24>::load(v, ghost, ghostFaceCB, f, nullptr, f);
  else Planar<double, 24>::load(v, in, stride, nbr, nullptr, nbr);
  Planar<double, 24>::store(v, out, stride, idx, nullptr, idx);
}

// out: parity `parity` block, in: the other parity's block (both [12 double2 planes][stride])
void applyShift(double *out, const double *in, const LatticeGeom &g, int stride, int parity, int dir) {
  const int mu = dir >> 1;
  const void *ghost = nullptr;
  if (commGrid().partitioned(mu)) {
    const size_t bytes = (size_t)g.faceCB[mu] * 24 * sizeof(double);
    if (bytes > g_ffBytes) {
      freeFullFaceBuffers();
      HIP_CHECK(qaMalloc((void **)&g_ffSend, bytes));
      HIP_CHECK(qaMalloc((void **)&g_ffGhost, bytes));
      g_ffBytes = bytes;
    }
    const bool fwd = !(dir & 1);
    hipLaunchKernelGGL((face_full_pack_kernel<double>), dim3((g.faceCB[mu] + 255) / 256), dim3(256), 0, computeStream(), (void *)g_ffSend, (const void *)in,
                       (const float *)nullptr, stride, g.X[0], g.X[1], g.X[2], g.X[3], mu, fwd ? 0 : g.X[mu] - 1, 1 - parity, g.faceCB[mu]);
    HIP_CHECK(hipGetLastError());
    std::vector<HaloMsg> msgs;
    msgs.push_back({mu, fwd ? -1 : +1, g_ffSend, g_ffGhost, bytes});
    commExchange(msgs, computeStream());
    ghost = g_ffGhost;
  }
  hipLaunchKernelGGL(shift_kernel, dim3((g.Vh + 255) / 256), dim3(256), 0, computeStream(), out, in, stride, g.Vh, g.Xh, g.X[1], g.X[2], g.X[3], parity, dir, ghost,
                     g.faceCB[mu]);
  HIP_CHECK(hipGetLastError());
}

template <typename T> static void launchSite(ColorSpinorField &out, const ColorSpinorField &in, SiteOp op, double a, double b,
                                             const CloverField *cl, int parity, bool inverse) {
  using real = typename Store<T>::real;
  SiteArg<real> arg;
  if (out.Stride() != in.Stride() || out.VolumeCB() != in.VolumeCB()) errorQuda("site operator: output stride %d / volume %d against input %d / %d", out.Stride(), out.VolumeCB(), in.Stride(), in.VolumeCB());
  arg.out = out.V(); arg.outNorm = (float *)out.Norm();
  arg.in = in.V(); arg.inNorm = (const float *)in.Norm();
  arg.sp_stride = in.Stride(); arg.Vh = in.VolumeCB(); arg.op = op;
  arg.a = (real)a; arg.b = (real)b;
  arg.clA = arg.clAinv = nullptr; arg.clAn = arg.clAinvN = nullptr; arg.cl_stride = 0;
  const int bs = 256, nb = (arg.Vh + bs - 1) / bs;
  if (op == SITE_TWIST) {
    hipLaunchKernelGGL((site_kernel<T, false>), dim3(nb), dim3(bs), 0, computeStream(), arg);
  } else {
    if (!cl) errorQuda("clover field required");
    if (op == SITE_CLOVER && inverse) { arg.clA = cl->Ainv(parity); arg.clAn = cl->AinvNorm(parity); }
    else { arg.clA = cl->A(parity); arg.clAn = cl->Anorm(parity); }
    arg.clAinv = cl->Ainv(parity); arg.clAinvN = cl->AinvNorm(parity);
    arg.cl_stride = cl->stride;
    hipLaunchKernelGGL((site_kernel<T, true>), dim3(nb), dim3(bs), 0, computeStream(), arg);
  }
  HIP_CHECK(hipGetLastError());
}

void applySite(ColorSpinorField &out, const ColorSpinorField &in, SiteOp op, double a, double b, const CloverField *clover,
               int parity, bool inverse) {
  if (in.Precision() != out.Precision()) errorQuda("precision mismatch");
  if (in.SiteSubset() != QUDA_PARITY_SITE_SUBSET) errorQuda("parity fields required");
  if (clover && clover->precision != in.Precision()) errorQuda("clover precision mismatch");
  if (g_acctOn) {
    const double P = in.Precision();
    acct("site_kernel", (48 * P + (op == SITE_TWIST ? 0 : (op == SITE_CLOVER_TWIST_INV ? 144 : 72) * P)) * in.VolumeCB(), "level 0");
  }
  switch (in.Precision()) {
    case QUDA_DOUBLE_PRECISION: launchSite<double>(out, in, op, a, b, clover, parity, inverse); break;
    case QUDA_SINGLE_PRECISION: launchSite<float>(out, in, op, a, b, clover, parity, inverse); break;
    case QUDA_HALF_PRECISION: launchSite<short>(out, in, op, a, b, clover, parity, inverse); break;
    default: errorQuda("bad precision %d", in.Precision());
  }
}

long long dslashFlopsPerSite(DslashMode mode, bool xpay) {
  long long f = 1320;  // lib/dslash_quda.cuh:480-495
  switch (mode) {
    case DSLASH_PLAIN: f += xpay ? 48 : 0; break;
    case DSLASH_TWIST_INV: case DSLASH_TWIST_INV_DSLASH: case DSLASH_TWIST_XPAY: f += 48 + (xpay ? 48 : 0); break;  // lib/dslash_twisted_mass.cu:144-160
    case DSLASH_CLOVER_TWIST_INV: case DSLASH_CLOVER_TWIST_XPAY: f += 552 + (xpay ? 48 : 0); break;  // lib/dslash_twisted_clover.cu:213-230
  }
  return f;
}

long long dslashBytesPerSite(QudaPrecision prec, int recon, DslashMode mode, bool xpay) {
  const long long P = prec;
  long long b = 8LL * recon * P + 24 * P + 24 * P;
  const bool x = xpay || mode == DSLASH_TWIST_XPAY || mode == DSLASH_CLOVER_TWIST_XPAY;
  if (x) b += 24 * P;
  if (mode == DSLASH_CLOVER_TWIST_INV) b += 144 * P;
  if (mode == DSLASH_CLOVER_TWIST_XPAY) b += 72 * P;
  if (prec == QUDA_HALF_PRECISION) {
    b += 4 * (2 + (x ? 1 : 0));
    if (mode == DSLASH_CLOVER_TWIST_INV) b += 16;
    if (mode == DSLASH_CLOVER_TWIST_XPAY) b += 8;
  }
  return b;
}

}  // namespace quda
