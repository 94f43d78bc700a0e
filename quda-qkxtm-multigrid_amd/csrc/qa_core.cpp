// qa_core.cpp — logging / error convention, streams, process-grid state.
// Reference behaviour: include/util_quda.h:40-105 (printfQuda/warningQuda/errorQuda, verbosity),
// lib/comm_common.cpp (topology), include/quda_internal.h:314-319 (streams).
#include <string>
#include "qa_core.h"

#include <map>
#include <vector>

#include <cstring>

namespace quda {

static QudaVerbosity g_verbosity = QUDA_SUMMARIZE;
static char g_prefix[128] = "";
static FILE *g_out = nullptr;

QudaVerbosity getVerbosity() { return g_verbosity; }
void setVerbosityInternal(QudaVerbosity v, const char *prefix, FILE *f) {
  g_verbosity = v;
  if (prefix) { strncpy(g_prefix, prefix, sizeof(g_prefix) - 1); g_prefix[sizeof(g_prefix) - 1] = 0; }
  if (f) g_out = f;
}

const char *outputPrefixInternal() { return g_prefix; }
FILE *outputFileInternal() { return g_out ? g_out : stdout; }

// identification of the last stencil launch for error messages (the reference quotes its last tune-cache key there);
// recorded as raw values per launch, formatted only when an error message asks for it
static const char *g_lastKernel = "none";
static int g_lastX[4] = {0, 0, 0, 0}, g_lastPrec = 0, g_lastRecon = 0, g_lastBlock = 0;
void setLastKernel(const char *kernel, const int X[4], int prec, int recon, int block) {
  g_lastKernel = kernel; g_lastPrec = prec; g_lastRecon = recon; g_lastBlock = block;
  for (int d = 0; d < 4; d++) g_lastX[d] = X[d];
}
void lastKernelKey(char *volume, int vn, char *name, int nn, char *aux, int an) {
  snprintf(volume, vn, "%dx%dx%dx%d", g_lastX[0], g_lastX[1], g_lastX[2], g_lastX[3]);
  snprintf(name, nn, "%s", g_lastKernel);
  snprintf(aux, an, "prec=%d,recon=%d,block=%d", g_lastPrec, g_lastRecon, g_lastBlock);
}

void qa_printf(const char *fmt, ...) {
  if (commGrid().rank != 0) return;
  FILE *f = g_out ? g_out : stdout;
  fputs(g_prefix, f);
  va_list ap;
  va_start(ap, fmt);
  vfprintf(f, fmt, ap);
  va_end(ap);
  fflush(f);
}

void qa_warning(const char *fmt, ...) {
  if (g_verbosity == QUDA_SILENT || commGrid().rank != 0) return;
  FILE *f = g_out ? g_out : stdout;
  fprintf(f, "%sWARNING: ", g_prefix);
  va_list ap;
  va_start(ap, fmt);
  vfprintf(f, fmt, ap);
  va_end(ap);
  fputc('\n', f);
  fflush(f);
}

// A caller that has results to deliver (bench.py: its finished JSON line, while an optional leg is still running) can leave them
// here: an error then writes the text to stdout before the process ends, with the status the caller asked for.
// The status is never 0: a failure must reach the launcher as a failure (a status of 0 is replaced by 1).
static std::string g_exitLine;
static int g_exitStatus = 1;
void setExitLine(const char *text, int status) {
  g_exitLine = text ? text : "";
  g_exitStatus = (text && status != 0) ? status : 1;
}
// the one way out after an error: errorQuda, the util_quda.h errorQuda of linked callers (compat.cpp qudaLogError) and comm_abort
void abortWithExitLine(int status) {
  if (!g_exitLine.empty()) { fputs(g_exitLine.c_str(), stdout); fputc('\n', stdout); fflush(stdout); }
  exit(status != 0 ? status : g_exitStatus);
}

void qa_error(const char *file, int line, const char *func, const char *fmt, ...) {
  FILE *f = g_out ? g_out : stderr;
  fprintf(f, "%sERROR: ", g_prefix);
  va_list ap;
  va_start(ap, fmt);
  vfprintf(f, fmt, ap);
  va_end(ap);
  fprintf(f, " (rank %d, %s:%d in %s())\n", commGrid().rank, file, line, func);
  fflush(f);
  abortWithExitLine(g_exitStatus);  // comm_abort(1) of the reference, lib/comm_single.cpp:58-63
}

CommGrid &commGrid() {
  static CommGrid g;
  return g;
}

static hipStream_t g_compute = nullptr, g_comm = nullptr;
void createStreams() {
  if (!g_compute) HIP_CHECK(hipStreamCreateWithFlags(&g_compute, hipStreamNonBlocking));
  if (!g_comm) HIP_CHECK(hipStreamCreateWithFlags(&g_comm, hipStreamNonBlocking));
}
void destroyStreams() {
  if (g_compute) (void)hipStreamDestroy(g_compute);
  if (g_comm) (void)hipStreamDestroy(g_comm);
  g_compute = g_comm = nullptr;
}
hipStream_t computeStream() { return g_compute; }
hipStream_t commStream() { return g_comm; }

bool g_acctOn = false;
struct AcctRec { std::string kernel, tag; double bytes; };
static std::vector<AcctRec> g_acct;
void acctRecord(const char *kernel, double bytes, const char *tag) { g_acct.push_back({kernel, tag ? tag : "", bytes}); }
void acctStart() { g_acct.clear(); g_acctOn = true; }
void acctDump(const char *path) {
  g_acctOn = false;
  FILE *f = fopen(path, "w");
  if (!f) errorQuda("cannot write %s", path);
  fprintf(f, "[\n");
  for (size_t i = 0; i < g_acct.size(); i++)
    fprintf(f, " {\"kernel\": \"%s\", \"bytes\": %.0f, \"tag\": \"%s\"}%s\n", g_acct[i].kernel.c_str(), g_acct[i].bytes, g_acct[i].tag.c_str(), i + 1 < g_acct.size() ? "," : "");
  fprintf(f, "]\n");
  fclose(f);
  g_acct.clear();
}

// Size-bucketed device pool with best-fit reuse: a request is served by the smallest parked buffer of at least that size and at most 1.5 x
// it (the multigrid set-up parks 4 GB block fields and 24 GB vector matrices; the next stage's temporaries are a little smaller or equal and
// a fresh hipMalloc of that size costs 0.1-1 s); the real size of every buffer is remembered, the size passed to poolDeviceFree is ignored.
static std::multimap<size_t, void *> g_pool;
static std::map<void *, size_t> g_poolSize;

void *poolDeviceMalloc(size_t bytes) {
  auto it = g_pool.lower_bound(bytes);
  if (it != g_pool.end() && it->first <= bytes + bytes / 2) {
    void *p = it->second;
    g_pool.erase(it);
    return p;
  }
  void *p = nullptr;
  HIP_CHECK(qaMallocRaw(&p, bytes));
  g_poolSize[p] = bytes;
  return p;
}
hipError_t qaMallocRaw(void **p, size_t bytes) {
  hipError_t e = hipMalloc(p, bytes);
  if (e == hipErrorOutOfMemory) {   // memory parked in the pool (tens of GB after a 48^3 x 96 set-up) is given back before giving up
    (void)hipGetLastError();
    poolDeviceFlush();
    e = hipMalloc(p, bytes);
  }
  return e;
}
void poolDeviceFree(void *ptr, size_t) {
  if (!ptr) return;
  auto it = g_poolSize.find(ptr);
  if (it == g_poolSize.end()) errorQuda("poolDeviceFree of a pointer the pool did not hand out");
  g_pool.insert({it->second, ptr});
}
void poolDeviceFlush(size_t atLeast) {
  for (auto it = g_pool.lower_bound(atLeast); it != g_pool.end();) {
    (void)hipFree(it->second);
    g_poolSize.erase(it->second);
    it = g_pool.erase(it);
  }
}

}  // namespace quda
