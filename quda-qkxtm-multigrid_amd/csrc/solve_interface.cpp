// solve_interface.cpp — invertQuda / newMultigridQuda / destroyMultigridQuda (reference
// lib/interface_quda.cpp:2161-2540).
#include "interface_internal.h"

using namespace quda;

extern "C" {

void invertQuda(void *, void *, QudaInvertParam *) { errorQuda("invertQuda: solver layer not built yet"); }
void *newMultigridQuda(QudaMultigridParam *) { errorQuda("newMultigridQuda: multigrid layer not built yet"); return nullptr; }
void destroyMultigridQuda(void *) {}

}
