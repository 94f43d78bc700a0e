// multigrid.h — adaptive multigrid (reference include/multigrid.h, lib/multigrid.cpp).  Filled in by multigrid.cpp.
#pragma once

#include "solver.h"

namespace quda {

class MG;

// opaque object handed out by newMultigridQuda (reference include/multigrid.h:375-411)
struct multigrid_solver {
  Solver *mg = nullptr;
};

}  // namespace quda
