// qa_core.h — common definitions of the MI355X-native QUDA-compatible library.
//
// Error convention follows the reference (include/util_quda.h:51-61): errorQuda prints and ends the
// process.  Geometry conventions follow SURVEY.md section 9 (verified against tests/test_util.cpp:406-471).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "quda.h"

namespace quda {

void qa_error(const char *file, int line, const char *func, const char *fmt, ...);
void qa_printf(const char *fmt, ...);
void qa_warning(const char *fmt, ...);
QudaVerbosity getVerbosity();

void setExitLine(const char *text, int status);   // see qa_core.cpp
[[noreturn]] void abortWithExitLine(int status);
#define errorQuda(...) ::quda::qa_error(__FILE__, __LINE__, __func__, __VA_ARGS__)
#define printfQuda(...) ::quda::qa_printf(__VA_ARGS__)
#define warningQuda(...) ::quda::qa_warning(__VA_ARGS__)

#define HIP_CHECK(cmd)                                                                         \
  do {                                                                                         \
    hipError_t e_ = (cmd);                                                                     \
    if (e_ != hipSuccess) errorQuda("HIP call '%s' failed: %s", #cmd, hipGetErrorString(e_));  \
  } while (0)

// Unsigned division by a runtime-invariant divisor without the ~40-instruction software divide that
// gfx950 would otherwise emit (Granlund-Montgomery round-up form; exact for all 32-bit n, d >= 1).
struct FastDiv {
  uint32_t m, s1, s2, d;
  FastDiv() : m(0), s1(0), s2(0), d(1) {}
  explicit FastDiv(uint32_t d_) : d(d_) {
    uint32_t l = 0;
    while ((1ull << l) < d_) l++;
    m = (uint32_t)(((1ull << 32) * ((1ull << l) - d_)) / d_ + 1);
    s1 = l < 1 ? l : 1;
    s2 = l > 0 ? l - 1 : 0;
  }
  __host__ __device__ inline uint32_t div(uint32_t n) const {
#ifdef __HIP_DEVICE_COMPILE__
    uint32_t t = __umulhi(m, n);
#else
    uint32_t t = (uint32_t)(((uint64_t)m * n) >> 32);
#endif
    return (t + ((n - t) >> s1)) >> s2;
  }
};

// Local 4-D lattice geometry of one rank. X[0] is the un-halved x extent.
struct LatticeGeom {
  int X[4];
  int Xh;      // X[0]/2
  int Vh, V;   // checkerboard / full volume
  int faceCB[4];  // checkerboard sites in the face orthogonal to dim d
  FastDiv dXh, dY, dZ;
  LatticeGeom() {}
  explicit LatticeGeom(const int x[4]) {
    V = 1;
    for (int d = 0; d < 4; d++) { X[d] = x[d]; V *= x[d]; }
    Vh = V / 2;
    Xh = X[0] / 2;
    for (int d = 0; d < 4; d++) faceCB[d] = Vh / X[d];
    dXh = FastDiv((uint32_t)Xh);
    dY = FastDiv((uint32_t)X[1]);
    dZ = FastDiv((uint32_t)X[2]);
  }
  bool operator==(const LatticeGeom &o) const { return X[0] == o.X[0] && X[1] == o.X[1] && X[2] == o.X[2] && X[3] == o.X[3]; }
};

// Global state shared by the C ABI (reference keeps the same things as file-scope globals,
// lib/interface_quda.cpp:119-145).
struct CommGrid {
  int dims[4] = {1, 1, 1, 1};
  int coords[4] = {0, 0, 0, 0};
  int rank = 0, size = 1;
  bool partitioned(int d) const { return dims[d] > 1 || forced[d]; }
  bool forced[4] = {false, false, false, false};  // single-process self-neighbour emulation (reference --partition)
  QudaCommsMap user_map = nullptr;
  void *user_data = nullptr;
};
CommGrid &commGrid();

// Size-bucketed cache of device allocations for the short-lived work fields (Krylov spaces, solver temporaries, API staging
// spinors): hipMalloc / hipFree synchronise the device and occasionally stall for ~100 ms when the runtime has to grow or
// trim its memory pool (measured inside invertQuda); the reference has the same kind of pool (lib/malloc.cpp pool_device_malloc).
// Blocks return to the cache on free and are handed out again to the next request of the same size — safe because every user
// works in compute-stream order.  Emptied by endQuda.
// Launch accounting for profile post-processing (tools/mg_solve_profile.py): while switched on, every instrumented launch site appends
// (kernel base name as rocprofv3 prints it, ALGORITHMIC bytes of this launch, a tag: level / precision) to a list; the i-th record of a
// name belongs to the i-th dispatch of that name in the kernel trace.  Off by default (one branch per launch).
extern bool g_acctOn;
void acctRecord(const char *kernel, double bytes, const char *tag);
inline void acct(const char *kernel, double bytes, const char *tag = "") { if (g_acctOn) acctRecord(kernel, bytes, tag); }
void acctStart();
void acctDump(const char *path);   // JSON: [{"kernel":..., "bytes":..., "tag":...}, ...] in launch order; switches the accounting off
// SciDAC / QIO single-file container of Nvec colour-spinor fields (csrc/lime_io.cpp): fp32 host fields of the local lattice in even-odd
// site order; every rank writes / reads its own rows of the one file
bool scidacIsContainer(const char *fname);
void scidacWriteSpinors(const char *fname, const std::vector<const float *> &vecs, const int X[4], int nSpin, int nColor);
void scidacReadSpinors(const char *fname, const std::vector<float *> &vecs, const int X[4], int nSpin, int nColor);
void *poolDeviceMalloc(size_t bytes);
// hipMalloc that hands the parked pool buffers back and tries once more when the device is out of memory
hipError_t qaMallocRaw(void **p, size_t bytes);
template <typename T> inline hipError_t qaMalloc(T **p, size_t bytes) { return qaMallocRaw((void **)p, bytes); }
void poolDeviceFree(void *ptr, size_t bytes);
void poolDeviceFlush(size_t atLeast = 0);   // returns the parked buffers of at least that many bytes to the runtime (0: all)

void setLastKernel(const char *kernel, const int X[4], int prec, int recon, int block);

hipStream_t computeStream();
hipStream_t commStream();

}  // namespace quda
