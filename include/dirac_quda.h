/* dirac_quda.h — reference header name (include/dirac_quda.h:88-1032) for quda::Dirac, DiracWilson[PC], DiracTwistedMass[PC],
 * DiracTwistedClover[PC], DiracCoarse[PC] and the DiracM / DiracMdagM / DiracMdag functors: dirac.h, coarse.h */
#ifndef QUDA_AMD_FWD_DIRAC_QUDA_H
#define QUDA_AMD_FWD_DIRAC_QUDA_H
#include <dirac.h>
#include <coarse.h>
#endif
