/* quda_internal.h — reference header name (include/quda_internal.h) for the library-wide definitions: qa_core.h */
#ifndef QUDA_AMD_FWD_QUDA_INTERNAL_H
#define QUDA_AMD_FWD_QUDA_INTERNAL_H
#include <qa_core.h>
#endif
