/*
 * comm_quda.h — process-grid queries and host-side collectives, the part of the reference's include/comm_quda.h that callers
 * outside the library use (tests/test_util.cpp, tests/blas_reference.cpp, the QKXTM drivers: comm_rank, comm_dim,
 * comm_coord, comm_dim_partitioned, comm_allreduce*, comm_broadcast, comm_barrier, comm_abort).  Same names, arguments and
 * meaning; the transport underneath is RCCL over xGMI (csrc/comm.cpp), bootstrapped through quda_amd_ext.h.  The reference's
 * message-handle API (comm_declare_send_relative, comm_start, ...) has no counterpart: halo traffic never passes through the
 * host here.
 */
#ifndef _COMM_QUDA_H
#define _COMM_QUDA_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int (*QudaCommsMap)(const int *coords, void *fdata);   /* as in quda.h */

char *comm_hostname(void);
double comm_drand(void);                 /* uniform [0,1), same sequence on every rank */
int comm_rank(void);
int comm_size(void);
int comm_gpuid(void);
int comm_dim(int dim);                   /* processes along dimension dim */
int comm_coord(int dim);                 /* this process' coordinate along dim */
int comm_dim_partitioned(int dim);
void comm_dim_partitioned_set(int dim);  /* single-process emulation of a partitioned dimension (tests --partition) */
int comm_partitioned(void);
void comm_allreduce(double *data);
void comm_allreduce_max(double *data);
void comm_allreduce_array(double *data, size_t size);
void comm_allreduce_int(int *data);
void comm_broadcast(void *data, size_t nbytes);   /* from rank 0 */
void comm_barrier(void);
void comm_abort(int status);

#ifdef __cplusplus
}
#endif

#endif /* _COMM_QUDA_H */
