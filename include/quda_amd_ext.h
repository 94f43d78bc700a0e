/*
 * quda_amd_ext.h — C-ABI handles onto the C++ surface of the library (resident fields and operator
 * objects), for callers that cannot bind C++ classes (ctypes / cgo / JNI style FFI) and for the
 * benchmark, which — like the reference's tests/dslash_test.cpp with transfer=0 (:455-616) — times
 * Dirac::Dslash on fields that stay in HBM.
 *
 * Each function is a thin wrapper over the C++ method it cites:
 *   qudaAmdSpinor*      -> cudaColorSpinorField ctor / operator= (reference include/color_spinor_field.h:458-640,
 *                          lib/cuda_color_spinor_field.cu:513-590)
 *   qudaAmdDirac*       -> Dirac::create, Dslash, DslashXpay, M, Mdag, MdagM, prepare/reconstruct
 *                          (reference include/dirac_quda.h:88-164, :449-617; lib/interface_quda.cpp:1265, :1386)
 *   qudaAmdBlas*        -> blas::norm2 / cDotProduct / axpy (reference include/blas_quda.h:33-144)
 * Plain pointers and PODs only; errors follow quda.h (message + exit(1)).
 */
#ifndef QUDA_AMD_EXT_H
#define QUDA_AMD_EXT_H

#include <stddef.h>
#include "quda.h"

#ifdef __cplusplus
extern "C" {
#endif

/* device-resident spinor field on the local lattice loaded by loadGaugeQuda.
 * site_subset: QUDA_PARITY_SITE_SUBSET (1) or QUDA_FULL_SITE_SUBSET (2). */
void *qudaAmdSpinorCreate(QudaPrecision prec, QudaSiteSubset site_subset, QudaTwistFlavorType flavor);
void qudaAmdSpinorDestroy(void *field);
/* host <-> device with reorder, precision and gamma-basis change; host layout described by inv_param
 * (cpu_prec, dirac_order, gamma_basis) exactly as for dslashQuda */
void qudaAmdSpinorLoad(void *field, const void *h_src, const QudaInvertParam *inv_param);
void qudaAmdSpinorSave(const void *field, void *h_dst, const QudaInvertParam *inv_param);
void qudaAmdSpinorCopy(void *dst, const void *src);   /* device-device, any precision pair */
void qudaAmdSpinorSetTwist(void *field, QudaTwistFlavorType flavor);

/* operator object built from the resident gauge/clover fields.
 * pc != 0: even-odd preconditioned type (Dirac::create of *PC_DIRAC); which: 0 precise, 1 sloppy, 2 precondition */
void *qudaAmdDiracCreate(QudaInvertParam *inv_param, int pc, int which);
void qudaAmdDiracDestroy(void *dirac);
void qudaAmdDiracDslash(void *dirac, void *out, const void *in, QudaParity parity);
void qudaAmdDiracDslashXpay(void *dirac, void *out, const void *in, QudaParity parity, const void *x, double k);
void qudaAmdDiracM(void *dirac, void *out, const void *in);
void qudaAmdDiracMdag(void *dirac, void *out, const void *in);
void qudaAmdDiracMdagM(void *dirac, void *out, const void *in);
unsigned long long qudaAmdDiracFlops(void *dirac);
/* Dirac::prepare / reconstruct (include/dirac_quda.h:152-164) on resident FULL fields x, b: prepare leaves the source of the
 * preconditioned system in src_out (a parity field; a full field for un-preconditioned operators), reconstruct completes x */
void qudaAmdDiracPrepare(void *dirac, void *src_out, void *x, void *b, QudaSolutionType solution_type);
void qudaAmdDiracReconstruct(void *dirac, void *x, const void *b, QudaSolutionType solution_type);

/* niter back-to-back Dslash applications bracketed by device events on the compute stream;
 * returns seconds per application (reference tests/dslash_test.cpp:455-616). */
double qudaAmdTimeDslash(void *dirac, void *out, const void *in, QudaParity parity, int niter);
double qudaAmdTimeM(void *dirac, void *out, const void *in, int niter);

double qudaAmdBlasNorm2(const void *field);
void qudaAmdBlasCDot(const void *x, const void *y, double result[2]);
void qudaAmdBlasAxpy(double a, const void *x, void *y);
double qudaAmdTimeAxpy(double a, const void *x, void *y, int niter);   /* seconds per y += a x, device-event timed */

/* algorithmic work model of the stencil kernel behind Dirac::Dslash */
long long qudaAmdDslashBytesPerSite(QudaInvertParam *inv_param, int which, int xpay);
long long qudaAmdDslashFlopsPerSite(QudaInvertParam *inv_param, int xpay);

/* multigrid introspection: the reference's MG::verify() identities (lib/multigrid.cpp:372-486) and one preconditioner
 * application K b on host vectors (full fields, layout described by inv_param as for MatQuda) */
void qudaAmdMultigridVerify(void *mg_instance, double dev[3]);
void qudaAmdMultigridCycle(void *mg_instance, void *h_x, void *h_b, QudaInvertParam *inv_param);

/* Read-only access to a hierarchy built by newMultigridQuda — what a C++ test of the reference reaches through
 * multigrid_solver->mg (include/multigrid.h:108-330: B, transfer, diracCoarse).  Host layouts are the reference's CPU
 * orders, fp32: vectors site-major (parity*Vh + x_cb, spin, colour, re/im), DeGrand-Rossi basis on level 0; V as
 * (site, spin, colour, vector) (lib/transfer_util.cu:15-36); coarse links as QDP-ordered
 * Y[dim 0-3 backward | 4-7 forward][site][row][col] and X[site][row][col] (lib/dslash_coarse.cu:50-64) with the
 * operator's -kappa already multiplied into Y.  `level` names the finer of the two levels a transfer connects. */
/* opt-in half-precision cycle: R, P and the coarse operators stream fp16 mirrors of the null-vector matrix V and of the
 * coarse links, and the level-0 even-odd smoother iterates in 16-bit storage (work fields + a 16-bit copy of the links made
 * on the device; twisted-mass / Wilson only).  Setup, verify, residuals, the accessors below and the outer solve keep their
 * precision.  The V / link switch is process-wide; `on` also creates the mirrors of this hierarchy.  Env QUDA_AMD_MG_HALF=1
 * does the same at newMultigridQuda. */
void qudaAmdMultigridSetHalfStorage(void *mg_instance, int on);
int qudaAmdMultigridLevels(void *mg_instance);
/* set-up refinement: every level-0 null vector is replaced by (one multigrid cycle)^cycles applied to it and the hierarchy is rebuilt, `passes`
 * times (an inverse iteration through the hierarchy; for problems at a critical kappa, where the first BiCGstab set-up stops at its cap).
 * The QudaMultigridParam the hierarchy was created from need not be alive any more.  Returns the seconds spent. */
double qudaAmdMultigridRefine(void *mg_instance, int passes, int cycles);
int qudaAmdMultigridOrthoFallbackBlocks(void *mg_instance, int level); /* blocks the fp32 CholeskyQR2 block orthonormalisation handed to Gram-Schmidt (ill-conditioned) */
void qudaAmdMultigridLevelInfo(void *mg_instance, int level, int info[18]); /* Xf[4] Xc[4] fineSpin fineColor Nvec geo_bs[4] spin_bs null_vector_method (0 sequential solves / loaded, 1 lockstep on the multi-rhs fine stencil, 2 lockstep on the MFMA coarse operator) lockstep_iterations */
void qudaAmdMultigridGetNullVector(void *mg_instance, int level, int k, float *h_out);
void qudaAmdMultigridGetV(void *mg_instance, int level, float *h_out);
void qudaAmdMultigridGetCoarseLinks(void *mg_instance, int level, float *h_Y, float *h_X);
/* op 0: R (level -> level+1; lib/restrictor.cu), 1: P (level+1 -> level; lib/prolongator.cu), 2: M of `level`,
 * 3: one multigrid cycle of `level`, x = K b (MG::operator(), lib/multigrid.cpp:488-604) */
void qudaAmdMultigridApply(void *mg_instance, int level, int op, float *h_out, const float *h_in);
/* The operator of a COARSE level applied to nrhs (8, 16, 24 or 32) host vectors at once through the multi-right-hand-side kernel
 * on the matrix cores (v_mfma_f32_16x16x4_f32; the reference's multi-source coarse Dslash, lib/dslash_coarse.cu:294-333): the
 * vectors lie back to back in h_in / h_out, each in the layout of qudaAmdMultigridApply.  niter > 0 additionally times niter
 * back-to-back applications with device events and returns the seconds per application (0 otherwise).
 * qudaAmdMultigridTimeApply times the single-vector operator of a level the same way. */
double qudaAmdMultigridApplyBlock(void *mg_instance, int level, int nrhs, float *h_out, const float *h_in, int niter);
/* errorQuda ends the process (reference convention, include/util_quda.h:51-61).  A caller holding finished results can leave a text
 * here that is written to stdout first, and the exit status to use; text = NULL restores the default (nothing, status 1). */
/* this process' transport counters since initQuda: fine-grid exchanges through peer stores / staged (RCCL), the same for the coarse
 * grids, global sums inside the reduction kernel / through the collective library, fall-backs to the staged transport, block exchanges */
void qudaAmdCommStats(long long out[8]);
/* text of the device error record of a halo wait that ran out (dimension, direction, buffer, exchange number, expected and last-seen
 * flag, interpretation); returns 0 if none is recorded */
int qudaAmdDescribeHaloError(char *text, int n);
/* profile post-processing: a one-wave marker dispatch (kernel qa_profile_marker_kernel) that brackets a region of the rocprofv3 kernel
 * trace, and the launch accounting — between Start and Dump every instrumented launch is recorded with its ALGORITHMIC bytes */
void qudaAmdProfileMarker(int id);
void qudaAmdAccountStart(void);
void qudaAmdAccountDump(const char *path);
/* Nvec colour-spinor fields in the SciDAC / QIO single-file container the reference's read_spinor_field / write_spinor_field use
 * (lib/qio_field.cpp:198-328; LIME records scidac-private-file-xml ... scidac-binary-data, scidac-checksum).  V[i]: host field of the local
 * lattice X[4], even-odd site order, 2 nSpin nColor reals per site in `precision`; the file holds fp32.  Every rank moves its own rows. */
void qudaAmdWriteSpinorFields(const char *filename, void *V[], QudaPrecision precision, const int *X, int nColor, int nSpin, int Nvec);
void qudaAmdReadSpinorFields(const char *filename, void *V[], QudaPrecision precision, const int *X, int nColor, int nSpin, int Nvec);
void qudaAmdSetExitLine(const char *text, int status);
double qudaAmdMultigridTimeApply(void *mg_instance, int level, int niter);
/* The cycle from a coarse level down (smoothers, residual, R, coarsest-grid GCR, P) as ONE persistent kernel (include/coarse_cycle.h) where
 * the sub-hierarchy qualifies; on by default (QUDA_AMD_MG_FUSED=0 / SetFused(0): the kernel-per-operation path).  FusedStats: returns 1 if
 * `level` owns such a kernel and fills out[] with the last launch's device-wide barriers, coarsest-grid GCR iterations, its restarts,
 * halo exchanges and the grid size. */
void qudaAmdMultigridSetFused(int on);
/* Launch-parameter cache (include/tune.h; reference lib/tune.cpp:213-355): tunecache.tsv under QUDA_RESOURCE_PATH in the reference's text
 * format.  Load re-reads the file and returns the number of entries; Store / Lookup address one entry by the reference's key triple,
 * param = block.x y z, grid.x y z, shared_bytes, aux.x y z w; Sweeps = sweeps this process has run (0 when everything came from the file). */
int qudaAmdTuneCacheLoad(void);
void qudaAmdTuneCacheSave(void);
void qudaAmdTuneCacheStore(const char *volume, const char *name, const char *aux, const int param[11], float time, const char *comment);
int qudaAmdTuneCacheLookup(const char *volume, const char *name, const char *aux, int param[11], float *time);
long qudaAmdTuneSweeps(void);
int qudaAmdMultigridFusedStats(void *mg_instance, int level, long long out[5]);
/* counters of invertMultiSrcQuda since start-up: [0] multigrid cycles run for all sources at once (coarse levels on block fields), [1] those of
 * them whose fine-level smoothing also ran on block fields (multi-right-hand-side stencil; QUDA_AMD_MULTISRC_BLOCK_SMOOTHER=0 switches it off),
 * [2] four-source restrictor / prolongator launches (QUDA_AMD_MULTISRC_QUAD=0), [3] lockstep solves */
void qudaAmdMultiSrcStats(long long out[4]);
/* seconds per application of the restrictor (what = 0) or prolongator (what = 1) between `level` and `level + 1` */
double qudaAmdMultigridTimeTransfer(void *mg_instance, int level, int what, int niter);

/* ---- the solve loop of the QKXTM correlator drivers ----
 * calcMG_threepTwop_EvenOdd / calcMG_loop_wOneD_TSM_* (lib/interface_quda.cpp:6018-6531, :7093, :8535) open with the same
 * loop: for every spin-colour component of a point source, Gaussian-smear it with the APE-smeared links, solve for the up
 * quark (twist +, inv_param->preconditionerUP) and the down quark (twist -, preconditionerDN) with even-odd preconditioned
 * GCR, reconstruct, rescale by 2 kappa under mass normalisation.  The reference then contracts and writes HDF5 (out of
 * scope); these entry points hand the propagators back instead.
 * Host layouts are the QKXTM ones: sites lexicographic x fastest (LOCAL lattice of the calling rank), vectors
 * iv*24 + (spin*3 + colour)*2 + re/im in the UKQCD basis (lib/qudaQKXTM_Vector_Kepler.cpp:72-81), smearing links
 * gauge_APE[dir][iv*18 + (row*3 + col)*2 + re/im] (lib/qudaQKXTM_Gauge_Kepler.cpp:73-89, what mapEvenOddToNormalGauge
 * leaves in the driver, qkxtm/CalcMG_2pt3pt_EvenOdd.cpp:684-686); fp64. */
typedef struct QudaAmdSourceParam_s {
  int sourcePosition[4];   /* GLOBAL (x, y, z, t): qudaQKXTMinfo_Kepler::sourcePosition[i] (include/qudaQKXTM_Kepler_utils.h:51) */
  int nsmearGauss;         /* qudaQKXTMinfo_Kepler::nsmearGauss (:46); 0: point source, gauge_APE may be NULL */
  double alphaGauss;       /* qudaQKXTMinfo_Kepler::alphaGauss (:48) */
} QudaAmdSourceParam;
/* The smeared field that performAPEnStep leaves inside the library (the reference keeps it in the file-scope gaugeSmeared and has
 * no accessor): host copy, fp64, either in QDP order (lexicographic = 0: even sites then odd, as loadGaugeQuda takes it) or in
 * the QKXTM lexicographic order gauge_APE uses (lexicographic = 1).  The two functions below also accept gauge_APE = NULL and
 * then smear with that resident field — what the drivers' read-smeared-configuration step supplies otherwise. */
void qudaAmdSaveSmearedGauge(void **h_gauge, int lexicographic);
/* h_out = smear^nsmear(h_in) (QKXTM_Vector_Kepler::gaussianSmearing, lib/qudaQKXTM_Vector_Kepler.cpp:386-421); the lattice is
 * that of the resident gauge field */
void qudaAmdGaussianSmear(void *h_out, const void *h_in, void **gauge_APE, int nsmear, double alpha);
/* h_prop_up / h_prop_dn: 12 vectors each (index isc = spin*3 + colour of the source, as the reference's loop), V*24 doubles per
 * vector.  inv_param as the reference requires it (:6041-6054): QUDA_DIRECT_PC_SOLVE, QUDA_GCR_INVERTER, UKQCD basis,
 * QUDA_DIRAC_ORDER, symmetric even-even / odd-odd preconditioning, QUDA_MAT_SOLUTION; iter / secs / gflops are summed over
 * the 24 solves; twist_flavor and preconditioner are left at their last values (minus / DN), as in the reference. */
void qudaAmdCalcMGPropagators(void *h_prop_up, void *h_prop_dn, void **gauge_APE, QudaInvertParam *inv_param, const QudaAmdSourceParam *source);

/* Sink for the solutions of the QKXTM entry points of qudaQKXTM_Kepler.h (calcMG_threepTwop_EvenOdd, calcMG_loop_wOneD_TSM_*):
 * called once per finished solve with host vectors in the QKXTM layout (lexicographic sites of the LOCAL lattice, UKQCD spin,
 * V * 24 doubles; h_source may be NULL) that are valid only during the call.  This is where a driver runs the contractions the
 * reference does inside those functions.  No sink registered: the solutions are dropped after their norm has been printed. */
typedef void (*QudaAmdSolutionSink)(void *ctx, const char *kind, int index, int twist_flavor, const double *h_source, const double *h_solution, size_t nreal);
void qudaAmdSetSolutionSink(QudaAmdSolutionSink sink, void *ctx);

/* ILDG gauge configurations in LIME containers (the step in front of loadGaugeQuda in the QKXTM drivers).  qudaAmdReadLimeGauge
 * has the semantics of readLimeGauge / readLimeGaugeSmeared (qkxtm/QKXTM_read_conf.h:107-400, :819-835): every rank reads the
 * sub-block of its grid coordinates from the "ildg-binary-data" record into the even-odd QDP arrays gauge[4] (fp64, allocated
 * by the caller for the LOCAL volume), sets param->X to the local extents from "ildg-format", compares kappa of "xlf-info" with
 * inv_param (may be NULL) and applies no boundary condition.  The container format is restated in csrc/lime_io.cpp (c-lime is
 * not a dependency).  qudaAmdWriteLimeGauge writes such a file from one rank (xlf_info may be NULL). */
void qudaAmdReadLimeGauge(void **gauge, const char *fname, QudaGaugeParam *param, QudaInvertParam *inv_param, const int gridSize[4]);
void qudaAmdWriteLimeGauge(void **gauge, const char *fname, const QudaGaugeParam *param, const char *xlf_info);

/* RCCL bootstrap (the transport that replaces the reference's MPI layer, lib/comm_mpi.cpp:50-155): rank 0 obtains a
 * 128-byte id, the launcher broadcasts it out of band, every rank calls qudaAmdCommInit BEFORE initCommsGridQuda / initQuda. */
void qudaAmdCommGetUniqueId(void *out128);
void qudaAmdCommInit(const void *id128, int rank, int size);
int qudaAmdCommRank(void);
int qudaAmdCommSize(void);
void qudaAmdCommCoords(int coords[4]);
void qudaAmdCommBarrier(void);
void qudaAmdCommAllreduce(double *data, int n);   /* sum over ranks, in place */
void qudaAmdCommAllreduceMax(double *data, int n);

/* single-process emulation of a partitioned dimension (reference tests --partition, commDimPartitionedSet) */
void qudaAmdSetPartitionMask(int mask);

/* Raw device images of the resident fields, for checks of the layout contract (reference lib/color_spinor_field.cpp:129-216:
 * stride = volumeCB + pad, bytes per parity rounded up to 1 KiB, odd half at bytes / 2, fp32 norm array for 16-bit fields):
 * spinor info = {volume, volumeCB, stride, pad, nSpin, nColor, precision, fieldOrder, siteSubset, gammaBasis, bytes, norm_bytes,
 *   device address of v, of norm, byte offset of the odd half, of the odd norms, reals per plane entry (2 fp64 | 4 fp32 | 8 16-bit),
 *   twistFlavor, x[0], location};
 * gauge info (which: 0 precise, 1 sloppy, 2 precondition) = {address, bytes, stride, bytes of one (parity, direction) block,
 *   precision, reconstruct, Vh, boundary sign folded into the links, t_boundary};
 * clover info = {address of A, of the inverse, of the A norms, of the inverse norms, stride, bytes per parity, norm bytes per parity,
 *   precision, bytes, Vh, twisted}.  qudaAmdRawDeviceCopy copies `bytes` from a device address to the host. */
void qudaAmdSpinorRawInfo(const void *field, long long info[20]);
void qudaAmdGaugeRawInfo(int which, long long info[12]);
void qudaAmdCloverRawInfo(int which, long long info[12]);
void qudaAmdRawDeviceCopy(void *h_dst, long long device_address, size_t bytes);

/* stream the kernels are launched on (hipStream_t), for callers that bracket work with their own events */
void *qudaAmdComputeStream(void);
void qudaAmdDeviceSynchronize(void);
/* halo transport in use: -1 not decided yet (no partitioned Dslash so far), 0 staged RCCL send/recv, 1 direct peer stores
 * into IPC-mapped ghost zones (the reference's "p2p" vs staged comms policies, lib/dslash_policy.cuh:838-998) */
int qudaAmdHaloTransport(void);
int qudaAmdHaloWireFormat(void);   /* wire format of the peer-store ghost zones in use: 0 flag-in-data {word, flag} halves, 1 self-validating 16-byte atoms {3 words, flag} (dslash.h haloWireFormat) */
/* launch geometry of the fine-grid stencil kernel, the counterpart of the reference's autotuner entries for the dslash kernels
 * (lib/tune.cpp, TuneParam block / grid): key = "block" (threads per block, 0 automatic), "remap" (XCD-aware block mapping),
 * "order" (legacy slab order), "tiled" / "nxz" / "tz" / "tt" (plane-tiled block order: XCDs along z, tile extents),
 * "store_aux", "link_aux", "lds_pad".  Results never depend on these. */
void qudaAmdSetDslashTune(const char *key, int value);

#ifdef __cplusplus
}
#endif
#endif
