// p2p.h — peer-mapped windows: a kernel of one GPU stores straight into the memory of the neighbouring GPU over xGMI and the
// neighbour's kernel finds the data there.  Fine-grid halo: the faces travel as flag-in-data vectors (dslash.hip GhostLL) that
// the neighbour's boundary sites poll; coarse-grid halo and the all-reduce of the global sums: payload + a cumulative arrival
// counter bumped by a remote atomic.  No host round trip, no collective-library launch on the critical path (a grouped RCCL
// send/recv costs ~55 us per Dslash, measured; the whole 32x16x16x16 interior kernel takes 18 us).  The reference's counterpart
// is its CUDA-IPC "p2p" policy (lib/cuda_color_spinor_field.cu:1212-1400, lib/dslash_policy.cuh:838-998: cudaIpcOpenMemHandle'd
// ghost buffers, cudaMemcpyAsync into the peer, IPC events); like the reference, the library falls back to the staged transport
// (RCCL send/recv) when peer mapping is not possible — or when the probe / the first-use verification says it does not work.
#pragma once

#include <vector>

#include "qa_core.h"

namespace quda {

// fine-grained (uncached, system-coherent) device memory: remote GPUs store into it while a local kernel polls / reads it
void *p2pAlloc(size_t bytes);
void p2pFree(void *p);

struct PeerMap {
  void *peer[8] = {};            // slot = 2 * dim + (1: the +dim neighbour, 0: the -dim neighbour); own pointer for a self neighbour
  std::vector<void *> opened;    // IPC mappings to close
};
// Collective over all ranks.  false: some rank could not export / open a handle (nothing stays mapped).
bool commMapPeers(void *local, PeerMap &m);
void commUnmapPeers(PeerMap &m);
void commAllgatherBytes(const void *mine, void *all, size_t n);
// the same for EVERY rank (all-reduce windows): byRank[r] = rank r's allocation mapped here (own pointer for r = rank)
bool commMapAllRanks(void *local, std::vector<void *> &byRank, std::vector<void *> &opened);

// Decided once, collectively, at the first halo exchange: QUDA_AMD_HALO=rccl forces the staged transport; otherwise the
// windows are mapped and a token round trip through them must succeed on every rank.
bool p2pHaloEnabled();
void p2pReset();
int p2pTransport();   // -1 not decided yet, 0 staged (RCCL send/recv), 1 direct peer stores
void p2pDisable();    // fall back to the staged transport for the rest of the run (first-use verification failed)
int p2pTakeError();   // value of the device error word, cleared

// Device error RECORD (kP2pErrInts ints) filled by the first wait that runs out — later waits see word 0 set and return at once:
//   [0] code: 1 + hop direction (fine stencil: 2 mu forward, 2 mu + 1 backward hop), 17 + hop (coarse stencil)
//   [1] face site (fine) / coarse site (coarse) whose data was missing        [2] flag / arrival count the wait expected
//   [3], [4] what was last seen there (fine: the flags of the two 8-byte halves of the first vector that did not match;
//            coarse: the arrival counter)                                       [5] exchange number of this rank's window
//   [6] buffer (parity of the exchange number)                                  [7] index of that vector inside the face site
// p2pCheck aborts with all of it spelled out; p2pDescribeError formats the same text (tools, tests).
constexpr int kP2pErrInts = 16;
int *p2pErrorWord();
void p2pCheck(const char *where);
bool p2pDescribeError(char *text, size_t n);   // false: no error recorded
// Several ranks of this run sit on ONE device (rehearsals of N ranks on a 1-GPU box).  Their kernels then compete for the same CU slots,
// and the boundary-first block order of the fused peer-store launch can fill the device with blocks that spin on faces whose senders
// cannot get a slot any more (4 ranks on one MI355X: reproduced twice, r3_reh4c3_default.log — every record says "previous use", i.e.
// never written, none shows a foreign flag — while the interior-first order passes, r3_reh4c3_interior_first.log).  launchDslash
// therefore starts with the interior blocks when this is true.  One process per GPU, the production shape, is not affected.
bool p2pDeviceShared();
// counters of this process since initQuda: [0] fine-grid exchanges through peer stores, [1] through the staged (RCCL) transport,
// [2] / [3] the same for the coarse grids, [4] global sums done inside the reduction kernel (peer windows), [5] through the
// collective library, [6] fall-backs from peer stores to the staged transport, [7] multi-right-hand-side (block) exchanges
long long *p2pStats();

// 100 MHz constant-rate counter ticks a kernel waits for a neighbour before it gives up (QUDA_AMD_P2P_TIMEOUT_S, default 10 s)
unsigned long long p2pTimeoutTicks();

}  // namespace quda
