/* oracle/qo_mg.h — TEST INFRASTRUCTURE (see qo_fields.h). Multigrid pieces of the CPU restatement. */
#ifndef QO_MG_H
#define QO_MG_H
#ifdef __cplusplus
extern "C" {
#endif
#ifdef __cplusplus
}
#endif
#endif
