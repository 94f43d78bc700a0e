// halo.h — spinor halo (ghost zone) machinery for the fine-grid stencil when the lattice is grid-decomposed.
//
// Reference: cudaColorSpinorField::{pack,gather,sendStart,commsQuery,scatter} (lib/cuda_color_spinor_field.cu:1465-1860),
// packFaceWilsonKernel / packTwistedFaceWilsonKernel (lib/dslash_pack.cu:272, :610), the overlap policies of
// lib/dslash_policy.cuh:148-297.  Re-designed: one pack launch for every face, one grouped RCCL send/recv on the comms
// stream, the interior kernel on the compute stream meanwhile, then ONE exterior launch over the precomputed list of
// boundary sites which does all 8 hops of such a site (ghost-aware) and the fused epilogue — every site is produced by
// exactly one kernel, there is no partial-sum read-modify-write pass.
#pragma once

#include <vector>

#include "fields.h"
#include "p2p.h"

namespace quda {

struct HaloMsg {
  int dim, dir;   // dir = +1: sent to the +dim neighbour (and the matching receive comes from the -dim neighbour)
  void *send, *recv;
  size_t bytes;
};

// comm.cpp
void commExchange(const std::vector<HaloMsg> &msgs, hipStream_t s);
int commNeighborRank(int dim, int dir);
void commBarrier();

// per-precision ghost storage for spin-projected half spinors (12 reals / face site, planar, + fp32 scales for 16-bit)
struct HaloBuffers {
  QudaPrecision precision = QUDA_INVALID_PRECISION;
  int faceCB[4] = {0, 0, 0, 0};
  size_t face_bytes[4] = {0, 0, 0, 0};   // payload of one face (12 reals x faceCB) incl. norms, 16-byte aligned
  size_t norm_offset[4] = {0, 0, 0, 0};  // byte offset of the fp32 scales inside a face block (16-bit only)
  char *send[4][2] = {};                 // [dim][0: to -dim neighbour, 1: to +dim neighbour]
  char *ghost[4][2] = {};                // [dim][0: from -dim neighbour, 1: from +dim neighbour]
  char *pool = nullptr;
  size_t pool_bytes = 0;
  // peer-store transport (p2p.h): a fine-grained window holding double-buffered flag-in-data ghost zones [dim][k][buf], mapped
  // into the neighbours; peerGhost are the addresses inside THEIR windows where this rank's (dim, to_fwd) face lands
  bool p2p = false;
  char *window = nullptr;
  char *ghostBuf[4][2][2] = {};
  char *peerGhost[4][2][2] = {};   // [dim][to_fwd][buf]
  int verified = 0;                // 0: not yet (staged until then), 1: agrees with the staged transport, 3: verification in progress
  unsigned seq = 0;
  unsigned uses[4][2] = {};        // exchanges that used (dim, buf) so far = the flag carried by the words of the current one
  PeerMap map;
};

HaloBuffers &haloBuffers(const LatticeGeom &g, QudaPrecision prec);
void freeHaloBuffers();

// list of checkerboard indices (per output parity) that touch a partitioned boundary
struct BoundaryList {
  int *d_idx[2] = {nullptr, nullptr};
  int count[2] = {0, 0};
  int mask = -1;
  int X[4] = {0, 0, 0, 0};
};
const BoundaryList &boundaryList(const LatticeGeom &g, int mask);
void freeBoundaryLists();

}  // namespace quda
