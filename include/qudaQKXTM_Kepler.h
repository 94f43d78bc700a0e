/*
 * qudaQKXTM_Kepler.h — the multigrid entry points of the QKXTM correlator drivers under the reference's names and signatures
 * (reference include/qudaQKXTM_Kepler.h:484-508; bodies lib/interface_quda.cpp:6018-6560, :8535-9300, :7093-8530).
 *
 * What this library runs is the SOLVE LOOP each of them opens with — sources, Dirac::prepare, even-odd GCR preconditioned by the
 * multigrid hierarchies in param->preconditioner[UP|DN], Dirac::reconstruct, normalisation.  The contractions, momentum
 * projection and HDF5 / ASCII writers that follow every solve in the reference are QKXTM's own physics code and are not part
 * of this library: instead each finished solution is handed to the sink registered with qudaAmdSetSolutionSink
 * (quda_amd_ext.h), where the driver does its contractions.  Arguments that only those later stages consume (gauge for the
 * plaquette / derivative operators, file names, NUCLEON, momenta) are accepted and ignored.
 */
#ifndef _QUDAQKXTM_KEPLER_H
#define _QUDAQKXTM_KEPLER_H

#include <quda.h>
#include <qudaQKXTM_Kepler_utils.h>

/* for every source position info.sourcePosition[0 .. Nsources-1]: 12 Gaussian-smeared point sources x (up, down):
 * sink("prop_up" | "prop_dn", index = 12 * isource + spin * 3 + colour, flavour +1 | -1, source = NULL, solution) */
void calcMG_threepTwop_EvenOdd(void **gaugeSmeared, void **gauge, QudaGaugeParam *gauge_param, QudaInvertParam *param, quda::qudaQKXTMinfo_Kepler info,
                               char *filename_twop, char *filename_threep, quda::WHICHPARTICLE NUCLEON);

/* Nstoch Z4 noise sources (or, with the truncated solver method, TSM_NLP low-precision solves followed by TSM_NHP sources solved
 * both to the full and to the low precision):
 * sink("loop_stoch" | "loop_LP" | "loop_HP" | "loop_HP_LP", index = source number, flavour of param, source, solution) */
void calcMG_loop_wOneD_TSM_EvenOdd(void **gaugeToPlaquette, QudaInvertParam *param, QudaGaugeParam *gauge_param, quda::qudaQKXTM_loopInfo loopInfo,
                                   quda::qudaQKXTMinfo_Kepler info);

/* The same loop behind exact deflation.  The reference obtains the eigenvectors from ARPACK (not a dependency of this library):
 * arpackInfo.nEv must be 0 here, i.e. nothing is projected out and every source goes to the multigrid solver; nEv > 0 is an error. */
void calcMG_loop_wOneD_TSM_wExact(void **gaugeToPlaquette, QudaInvertParam *EVparam, QudaInvertParam *param, QudaGaugeParam *gauge_param,
                                  quda::qudaQKXTM_arpackInfo arpackInfo, quda::qudaQKXTM_loopInfo loopInfo, quda::qudaQKXTMinfo_Kepler info);

#endif /* _QUDAQKXTM_KEPLER_H */
