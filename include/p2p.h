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

// device word set by a wait that timed out (a neighbour never signalled); p2pCheck aborts with a message if it is set
int *p2pErrorWord();
void p2pCheck(const char *where);

// 100 MHz constant-rate counter ticks a kernel waits for a neighbour before it gives up (QUDA_AMD_P2P_TIMEOUT_S, default 10 s)
unsigned long long p2pTimeoutTicks();

}  // namespace quda
