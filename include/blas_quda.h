/* blas_quda.h — reference header name (include/blas_quda.h:33-144, namespace quda::blas): blas.h */
#ifndef QUDA_AMD_FWD_BLAS_QUDA_H
#define QUDA_AMD_FWD_BLAS_QUDA_H
#include <blas.h>
#endif
