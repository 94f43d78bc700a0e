/*
 * face_quda.h — the reference's include/face_quda.h declares the host-staged FaceBuffer exchange and the
 * commDimPartitioned family.  Faces never pass through the host in this library (csrc/halo.h: peer stores or one grouped RCCL
 * send/recv), so only the partitioning queries that drivers and tests call remain.
 */
#ifndef _FACE_QUDA_H
#define _FACE_QUDA_H

#include <sys/time.h>   /* callers of the reference header get it through quda_internal.h */
#include <quda.h>
#include <comm_quda.h>

/* C++ linkage, as in the reference header (include/face_quda.h:125-128) */
int commDim(int dim);
int commCoords(int dim);
int commDimPartitioned(int dir);
void commDimPartitionedSet(int dir);

#endif /* _FACE_QUDA_H */
