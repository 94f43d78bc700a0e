/*
 * quda_constants.h — compile-time limits that size the arrays inside the parameter structs of quda.h (values are ABI).
 * Counterpart of the reference's include/quda_constants.h:1-51; included by quda.h as there (include/quda.h:15).
 */
#ifndef _QUDA_CONSTANTS_H
#define _QUDA_CONSTANTS_H

#define QUDA_VERSION_MAJOR 0
#define QUDA_VERSION_MINOR 9
#define QUDA_VERSION_SUBMINOR 0
#define QUDA_VERSION ((QUDA_VERSION_MAJOR << 16) | (QUDA_VERSION_MINOR << 8) | QUDA_VERSION_SUBMINOR)
#define QUDA_MAX_DIM 6
#define QUDA_MAX_GEOMETRY 8
#define QUDA_MAX_MULTI_SHIFT 32
#define QUDA_MAX_DWF_LS 128
#define QUDA_MAX_MG_LEVEL 4
#define QUDA_MAX_MULTI_REDUCE 16

#endif /* _QUDA_CONSTANTS_H */
