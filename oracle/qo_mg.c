/* oracle/qo_mg.c — TEST INFRASTRUCTURE (see qo_fields.h). Multigrid pieces of the CPU restatement. */
#include "qo_mg.h"
