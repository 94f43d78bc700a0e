/*
 * util_quda.h — logging and the error convention, for callers written against the reference's include/util_quda.h:
 * printfQuda (rank 0 only), warningQuda (unless QUDA_SILENT), errorQuda (message with rank, file:line, function and the last
 * kernel key, then comm_abort(1)); verbosity / output prefix / output file accessors.  The macros forward to functions of
 * libquda.so instead of expanding to fprintf sequences at every call site.
 */
#ifndef _UTIL_QUDA_H
#define _UTIL_QUDA_H

#include <stdio.h>
#include <stdlib.h>
#include <enum_quda.h>
#include <comm_quda.h>
#include <tune_key.h>

QudaTune getTuning();
void setTuning(QudaTune tune);

QudaVerbosity getVerbosity();
char *getOutputPrefix();
FILE *getOutputFile();
void setVerbosity(const QudaVerbosity verbosity);
void setOutputPrefix(const char *prefix);
void setOutputFile(FILE *outfile);
void pushVerbosity(QudaVerbosity verbosity);
void popVerbosity();
char *getPrintBuffer();

/* implemented in libquda.so (csrc/compat.cpp) */
void qudaLogPrintf(const char *fmt, ...);
void qudaLogWarning(const char *fmt, ...);
void qudaLogError(const char *file, int line, const char *func, const char *fmt, ...);

#define printfQuda(...) qudaLogPrintf(__VA_ARGS__)
#define warningQuda(...) qudaLogWarning(__VA_ARGS__)
#define errorQuda(...) qudaLogError(__FILE__, __LINE__, __func__, __VA_ARGS__)
#define checkCudaErrorNoSync() do { } while (0)
#define checkCudaError() do { } while (0)

#endif /* _UTIL_QUDA_H */
