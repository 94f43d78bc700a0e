/* dslash_quda.h — reference header name (include/dslash_quda.h) for the stencil launchers: dslash.h */
#ifndef QUDA_AMD_FWD_DSLASH_QUDA_H
#define QUDA_AMD_FWD_DSLASH_QUDA_H
#include <dslash.h>
#endif
