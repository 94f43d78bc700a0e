// compat.cpp — host-utility symbols of the reference library that code OUTSIDE the library links against: the logging /
// error accessors of include/util_quda.h (reference lib/util_quda.cpp), the process-grid queries and host collectives of
// include/comm_quda.h (reference lib/comm_common.cpp, lib/comm_single.cpp / comm_mpi.cpp) and the commDim family of
// include/face_quda.h (reference lib/face_buffer.cpp:401-420).  The reference's own tests/*.cpp helpers and the QKXTM drivers
// call these; objects compiled against the reference's util_quda.h additionally need getOutputFile / getOutputPrefix /
// getLastTuneKey / comm_abort for their errorQuda expansions.  Everything forwards to the library's own state (qa_core.cpp,
// comm.cpp); nothing here is on a hot path.
#include <unistd.h>

#include <cstdarg>
#include <cstring>
#include <vector>

#include "blas.h"
#include "dslash.h"
#include "halo.h"
#include "interface_internal.h"
// the public header re-defines the logging macros of qa_core.h on top of the functions implemented below
#undef errorQuda
#undef printfQuda
#undef warningQuda
#include "util_quda.h"

namespace quda {
void setVerbosityInternal(QudaVerbosity v, const char *prefix, FILE *f);
const char *outputPrefixInternal();
FILE *outputFileInternal();
void commAllgatherBytes(const void *mine, void *all, size_t n);
void lastKernelKey(char *volume, int vn, char *name, int nn, char *aux, int an);
}  // namespace quda

using namespace quda;

static QudaTune g_tune = QUDA_TUNE_NO;
QudaTune getTuning() { return g_tune; }
void setTuning(QudaTune tune) { g_tune = tune; }

QudaVerbosity getVerbosity() { return quda::getVerbosity(); }
char *getOutputPrefix() { return const_cast<char *>(outputPrefixInternal()); }
FILE *getOutputFile() { return outputFileInternal(); }
void setVerbosity(const QudaVerbosity verbosity) { setVerbosityInternal(verbosity, nullptr, nullptr); }
void setOutputPrefix(const char *prefix) { setVerbosityInternal(quda::getVerbosity(), prefix, nullptr); }
void setOutputFile(FILE *outfile) { setVerbosityInternal(quda::getVerbosity(), nullptr, outfile); }

static std::vector<QudaVerbosity> g_vstack;
void pushVerbosity(QudaVerbosity verbosity) {
  g_vstack.push_back(quda::getVerbosity());
  if (g_vstack.size() > 10) qudaLogWarning("verbosity stack contains %u elements", (unsigned)g_vstack.size());
  setVerbosity(verbosity);
}
void popVerbosity() {
  if (g_vstack.empty()) qudaLogError(__FILE__, __LINE__, __func__, "popVerbosity() called with empty stack");
  setVerbosity(g_vstack.back());
  g_vstack.pop_back();
}
char *getPrintBuffer() {
  static char buf[8192];
  return buf;
}

void qudaLogPrintf(const char *fmt, ...) {
  if (commGrid().rank != 0) return;
  FILE *f = outputFileInternal();
  fputs(outputPrefixInternal(), f);
  va_list ap;
  va_start(ap, fmt);
  vfprintf(f, fmt, ap);
  va_end(ap);
  fflush(f);
}
void qudaLogWarning(const char *fmt, ...) {
  if (quda::getVerbosity() == QUDA_SILENT || commGrid().rank != 0) return;
  FILE *f = outputFileInternal();
  fprintf(f, "%sWARNING: ", outputPrefixInternal());
  va_list ap;
  va_start(ap, fmt);
  vfprintf(f, fmt, ap);
  va_end(ap);
  fputc('\n', f);
  fflush(f);
}
void qudaLogError(const char *file, int line, const char *func, const char *fmt, ...) {
  FILE *f = outputFileInternal();
  fprintf(f, "%sERROR: ", outputPrefixInternal());
  va_list ap;
  va_start(ap, fmt);
  vfprintf(f, fmt, ap);
  va_end(ap);
  const quda::TuneKey k = getLastTuneKey();
  fprintf(f, " (rank %d, host %s, %s:%d in %s())\n%s       last kernel called was (name=%s,volume=%s,aux=%s)\n", commGrid().rank, comm_hostname(), file, line,
          func, outputPrefixInternal(), k.name, k.volume, k.aux);
  fflush(f);
  comm_abort(1);
}

quda::TuneKey getLastTuneKey() {
  quda::TuneKey k;
  lastKernelKey(k.volume, quda::TuneKey::volume_n, k.name, quda::TuneKey::name_n, k.aux, quda::TuneKey::aux_n);
  return k;
}

int commDim(int dim) { return commGrid().dims[dim]; }
int commCoords(int dim) { return commGrid().coords[dim]; }
int commDimPartitioned(int dir) { return commGrid().partitioned(dir) ? 1 : 0; }
void commDimPartitionedSet(int dir) { commGrid().forced[dir] = true; }

extern "C" {

char *comm_hostname(void) {
  static char name[128] = "";
  if (!name[0]) { gethostname(name, sizeof(name) - 1); name[sizeof(name) - 1] = 0; }
  return name;
}
double comm_drand(void) {
  // the reference seeds a private generator identically on every rank (lib/comm_common.cpp:70-81): any fixed-seed LCG serves
  static unsigned long long state = 137;
  state = state * 6364136223846793005ull + 1442695040888963407ull;
  return (double)(state >> 11) * (1.0 / 9007199254740992.0);
}
int comm_rank(void) { return commGrid().rank; }
int comm_size(void) { return commGrid().size; }
int comm_gpuid(void) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  return dev;
}
int comm_dim(int dim) { return commGrid().dims[dim]; }
int comm_coord(int dim) { return commGrid().coords[dim]; }
int comm_dim_partitioned(int dim) { return commGrid().partitioned(dim) ? 1 : 0; }
void comm_dim_partitioned_set(int dim) { commGrid().forced[dim] = true; }
int comm_partitioned(void) {
  int p = 0;
  for (int d = 0; d < 4; d++) p = p || commGrid().partitioned(d);
  return p;
}
void comm_allreduce(double *data) { quda::comm_allreduce(data, 1); }
void comm_allreduce_max(double *data) { quda::comm_allreduce_max(data, 1); }
void comm_allreduce_array(double *data, size_t size) {
  for (size_t o = 0; o < size; o += 64) quda::comm_allreduce(data + o, (int)(size - o < 64 ? size - o : 64));
}
void comm_allreduce_int(int *data) {
  double d = (double)*data;
  quda::comm_allreduce(&d, 1);
  *data = (int)d;
}
void comm_broadcast(void *data, size_t nbytes) {
  const CommGrid &g = commGrid();
  if (g.size == 1) return;
  std::vector<char> all(nbytes * g.size);
  commAllgatherBytes(data, all.data(), nbytes);
  memcpy(data, all.data(), nbytes);   // rank 0's block
}
void comm_barrier(void) { commBarrier(); }
void comm_abort(int status) { quda::abortWithExitLine(status ? status : 1); }   // through the same exit path as errorQuda (exit line of qa_core.cpp)

}  // extern "C"
