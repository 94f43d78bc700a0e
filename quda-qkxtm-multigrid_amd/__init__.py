"""quda-qkxtm-multigrid_amd — MI355X-native twisted-mass Dslash / multigrid library behind the QUDA C ABI.

This package is only the Python-side binding used by tests and bench.py: ctypes mirrors of the PODs in
include/quda.h (field-for-field, the ABI of reference include/quda.h:25-409) and thin wrappers over the
extern "C" entry points of lib/libquda.so (built from csrc/ by `make`, see __graft_entry__.build()).
There is NO CPU fallback: importing works without a GPU (so the symbol table can be checked), but any
compute call needs a gfx950 device, and a missing library raises immediately.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("QUDA_AMD_LIBRARY") or os.path.join(_HERE, "lib", "libquda.so")   # override: A/B timing of two builds
INCLUDE_DIR = os.path.join(os.path.dirname(_HERE), "include")

# ---- enum values (include/quda.h) ----
QUDA_INVALID_ENUM = -(2**31)
QUDA_CPU_FIELD_LOCATION, QUDA_CUDA_FIELD_LOCATION = 1, 2
QUDA_HALF_PRECISION, QUDA_SINGLE_PRECISION, QUDA_DOUBLE_PRECISION = 2, 4, 8
QUDA_RECONSTRUCT_NO, QUDA_RECONSTRUCT_12, QUDA_RECONSTRUCT_8 = 18, 12, 8
QUDA_WILSON_LINKS = 0
QUDA_QDP_GAUGE_ORDER = 5
QUDA_ANTI_PERIODIC_T, QUDA_PERIODIC_T = -1, 1
QUDA_GAUGE_FIXED_NO = 0
QUDA_WILSON_DSLASH, QUDA_TWISTED_MASS_DSLASH, QUDA_TWISTED_CLOVER_DSLASH = 0, 7, 8
QUDA_CG_INVERTER, QUDA_BICGSTAB_INVERTER, QUDA_GCR_INVERTER, QUDA_MR_INVERTER, QUDA_MG_INVERTER = 0, 1, 2, 3, 15
QUDA_MAT_SOLUTION, QUDA_MATDAG_MAT_SOLUTION, QUDA_MATPC_SOLUTION, QUDA_MATPC_DAG_SOLUTION, QUDA_MATPCDAG_MATPC_SOLUTION = 0, 1, 2, 3, 4
QUDA_DIRECT_SOLVE, QUDA_NORMOP_SOLVE, QUDA_DIRECT_PC_SOLVE, QUDA_NORMOP_PC_SOLVE = 0, 1, 2, 3
QUDA_MG_CYCLE_VCYCLE, QUDA_MG_CYCLE_FCYCLE, QUDA_MG_CYCLE_WCYCLE, QUDA_MG_CYCLE_RECURSIVE = 0, 1, 2, 3
QUDA_ADDITIVE_SCHWARZ = 0
QUDA_L2_RELATIVE_RESIDUAL = 1
QUDA_MATPC_EVEN_EVEN, QUDA_MATPC_ODD_ODD, QUDA_MATPC_EVEN_EVEN_ASYMMETRIC, QUDA_MATPC_ODD_ODD_ASYMMETRIC = 0, 1, 2, 3
MATPC = {"ee": 0, "oo": 1, "eeasym": 2, "ooasym": 3}
QUDA_DAG_NO, QUDA_DAG_YES = 0, 1
QUDA_KAPPA_NORMALIZATION, QUDA_MASS_NORMALIZATION, QUDA_ASYMMETRIC_MASS_NORMALIZATION = 0, 1, 2
QUDA_DEFAULT_NORMALIZATION = 0
QUDA_PRESERVE_SOURCE_NO, QUDA_PRESERVE_SOURCE_YES = 0, 1
QUDA_DIRAC_ORDER, QUDA_QDP_DIRAC_ORDER = 1, 2
QUDA_PACKED_CLOVER_ORDER = 5
QUDA_SILENT, QUDA_SUMMARIZE, QUDA_VERBOSE, QUDA_DEBUG_VERBOSE = 0, 1, 2, 3
QUDA_TUNE_NO = 0
QUDA_EVEN_PARITY, QUDA_ODD_PARITY = 0, 1
QUDA_PARITY_SITE_SUBSET, QUDA_FULL_SITE_SUBSET = 1, 2
QUDA_DEGRAND_ROSSI_GAMMA_BASIS, QUDA_UKQCD_GAMMA_BASIS = 0, 1
QUDA_TWIST_MINUS, QUDA_TWIST_PLUS, QUDA_TWIST_NO = -1, 1, 0
QUDA_USE_INIT_GUESS_NO, QUDA_USE_INIT_GUESS_YES = 0, 1
QUDA_COMPUTE_NULL_VECTOR_NO, QUDA_COMPUTE_NULL_VECTOR_YES = 0, 1
QUDA_BOOLEAN_NO, QUDA_BOOLEAN_YES = 0, 1

QUDA_MAX_DIM, QUDA_MAX_MULTI_SHIFT, QUDA_MAX_DWF_LS, QUDA_MAX_MG_LEVEL = 6, 32, 128, 4

_i, _d, _p = C.c_int, C.c_double, C.c_void_p


class QudaGaugeParam(C.Structure):
    _fields_ = [("location", _i), ("X", _i * 4), ("anisotropy", _d), ("tadpole_coeff", _d), ("scale", _d), ("type", _i),
                ("gauge_order", _i), ("t_boundary", _i), ("cpu_prec", _i), ("cuda_prec", _i), ("reconstruct", _i),
                ("cuda_prec_sloppy", _i), ("reconstruct_sloppy", _i), ("cuda_prec_precondition", _i),
                ("reconstruct_precondition", _i), ("gauge_fix", _i), ("ga_pad", _i), ("site_ga_pad", _i), ("staple_pad", _i),
                ("llfat_ga_pad", _i), ("mom_ga_pad", _i), ("gaugeGiB", _d), ("preserve_gauge", _i),
                ("staggered_phase_type", _i), ("staggered_phase_applied", _i), ("i_mu", _d), ("overlap", _i),
                ("overwrite_mom", _i), ("use_resident_gauge", _i), ("use_resident_mom", _i), ("make_resident_gauge", _i),
                ("make_resident_mom", _i), ("return_result_gauge", _i), ("return_result_mom", _i)]


class QudaInvertParam(C.Structure):
    _fields_ = [("input_location", _i), ("output_location", _i), ("dslash_type", _i), ("inv_type", _i), ("mass", _d),
                ("kappa", _d), ("m5", _d), ("Ls", _i), ("b_5", _d * QUDA_MAX_DWF_LS), ("c_5", _d * QUDA_MAX_DWF_LS), ("mu", _d),
                ("epsilon", _d), ("twist_flavor", _i), ("tol", _d), ("tol_restart", _d), ("tol_hq", _d), ("true_res", _d),
                ("true_res_hq", _d), ("maxiter", _i), ("reliable_delta", _d), ("use_sloppy_partial_accumulator", _i),
                ("max_res_increase", _i), ("max_res_increase_total", _i), ("heavy_quark_check", _i), ("pipeline", _i),
                ("num_offset", _i), ("num_src", _i), ("overlap", _i), ("offset", _d * QUDA_MAX_MULTI_SHIFT),
                ("tol_offset", _d * QUDA_MAX_MULTI_SHIFT), ("tol_hq_offset", _d * QUDA_MAX_MULTI_SHIFT),
                ("true_res_offset", _d * QUDA_MAX_MULTI_SHIFT), ("iter_res_offset", _d * QUDA_MAX_MULTI_SHIFT),
                ("true_res_hq_offset", _d * QUDA_MAX_MULTI_SHIFT), ("solution_type", _i), ("solve_type", _i), ("matpc_type", _i),
                ("dagger", _i), ("mass_normalization", _i), ("solver_normalization", _i), ("preserve_source", _i),
                ("cpu_prec", _i), ("cuda_prec", _i), ("cuda_prec_sloppy", _i), ("cuda_prec_precondition", _i),
                ("dirac_order", _i), ("gamma_basis", _i), ("clover_location", _i), ("clover_cpu_prec", _i),
                ("clover_cuda_prec", _i), ("clover_cuda_prec_sloppy", _i), ("clover_cuda_prec_precondition", _i),
                ("clover_order", _i), ("use_init_guess", _i), ("clover_coeff", _d), ("compute_clover_trlog", _i),
                ("trlogA", _d * 2), ("compute_clover", _i), ("compute_clover_inverse", _i), ("return_clover", _i),
                ("return_clover_inverse", _i), ("verbosity", _i), ("sp_pad", _i), ("cl_pad", _i), ("iter", _i),
                ("spinorGiB", _d), ("cloverGiB", _d), ("gflops", _d), ("secs", _d), ("tune", _i), ("Nsteps", _i),
                ("gcrNkrylov", _i), ("inv_type_precondition", _i), ("preconditioner", _p), ("preconditionerUP", _p),
                ("preconditionerDN", _p), ("dslash_type_precondition", _i), ("verbosity_precondition", _i),
                ("tol_precondition", _d), ("maxiter_precondition", _i), ("omega", _d), ("precondition_cycle", _i),
                ("schwarz_type", _i), ("residual_type", _i), ("cuda_prec_ritz", _i), ("nev", _i), ("max_search_dim", _i),
                ("rhs_idx", _i), ("deflation_grid", _i), ("use_reduced_vector_set", _i), ("eigenval_tol", _d),
                ("use_cg_updates", _i), ("cg_iterref_tol", _d), ("eigcg_max_restarts", _i), ("max_restart_num", _i),
                ("inc_tol", _d), ("make_resident_solution", _i), ("use_resident_solution", _i)]


class QudaMultigridParam(C.Structure):
    _L = QUDA_MAX_MG_LEVEL
    _fields_ = [("invert_param", C.POINTER(QudaInvertParam)), ("n_level", _i), ("geo_block_size", (_i * QUDA_MAX_DIM) * _L),
                ("spin_block_size", _i * _L), ("n_vec", _i * _L), ("smoother", _i * _L), ("coarse_grid_solution_type", _i * _L),
                ("smoother_solve_type", _i * _L), ("cycle_type", _i * _L), ("nu_pre", _i * _L), ("nu_post", _i * _L),
                ("smoother_tol", _d * _L), ("setup_maxiter", _i), ("setup_tol", _d), ("omega", _d * _L),
                ("global_reduction", _i * _L), ("location", _i * _L), ("compute_null_vector", _i), ("generate_all_levels", _i),
                ("run_verify", _i), ("vec_infile", C.c_char * 256), ("vec_outfile", C.c_char * 256), ("gflops", _d), ("secs", _d),
                ("delta_muPR", _d), ("delta_kappaPR", _d), ("delta_cswPR", _d), ("delta_muCG", _d), ("delta_kappaCG", _d),
                ("delta_cswCG", _d)]


# every extern "C" symbol include/quda.h and include/quda_amd_ext.h declare
QUDA_H_SYMBOLS = ["setVerbosityQuda", "initCommsGridQuda", "initQudaDevice", "initQudaMemory", "initQuda", "endQuda",
                  "newQudaGaugeParam", "newQudaInvertParam", "newQudaMultigridParam", "printQudaGaugeParam", "printQudaInvertParam",
                  "printQudaMultigridParam", "loadGaugeQuda", "freeGaugeQuda", "loadCloverQuda", "freeCloverQuda", "invertQuda", "invertMultiSrcQuda",
                  "newMultigridQuda", "destroyMultigridQuda", "dslashQuda", "cloverQuda", "MatQuda", "MatDagMatQuda", "openMagma",
                  "closeMagma"]
EXT_H_SYMBOLS = ["qudaAmdSpinorCreate", "qudaAmdSpinorDestroy", "qudaAmdSpinorLoad", "qudaAmdSpinorSave", "qudaAmdSpinorCopy",
                 "qudaAmdSpinorSetTwist", "qudaAmdDiracCreate", "qudaAmdDiracDestroy", "qudaAmdDiracDslash", "qudaAmdDiracDslashXpay",
                 "qudaAmdDiracM", "qudaAmdDiracMdag", "qudaAmdDiracMdagM", "qudaAmdDiracFlops", "qudaAmdTimeDslash", "qudaAmdTimeM",
                 "qudaAmdBlasNorm2", "qudaAmdBlasCDot", "qudaAmdBlasAxpy", "qudaAmdDslashBytesPerSite", "qudaAmdDslashFlopsPerSite",
                 "qudaAmdSetPartitionMask", "qudaAmdComputeStream", "qudaAmdDeviceSynchronize", "qudaAmdHaloTransport", "qudaAmdHaloWireFormat", "qudaAmdSetDslashTune", "qudaAmdCommGetUniqueId",
                 "qudaAmdCommInit", "qudaAmdCommRank", "qudaAmdCommSize", "qudaAmdCommCoords", "qudaAmdCommBarrier", "qudaAmdCommAllreduce", "qudaAmdCommAllreduceMax",
                 "qudaAmdMultigridVerify", "qudaAmdMultigridCycle", "qudaAmdTimeAxpy", "qudaAmdMultigridLevels", "qudaAmdMultigridLevelInfo",
                 "qudaAmdMultigridSetHalfStorage", "qudaAmdMultigridGetNullVector", "qudaAmdMultigridGetV", "qudaAmdMultigridGetCoarseLinks", "qudaAmdMultigridApply", "qudaAmdMultigridApplyBlock",
                 "qudaAmdMultigridTimeApply", "qudaAmdMultigridTimeTransfer", "qudaAmdSetExitLine", "qudaAmdDiracPrepare", "qudaAmdDiracReconstruct", "qudaAmdSpinorRawInfo", "qudaAmdGaugeRawInfo", "qudaAmdCloverRawInfo", "qudaAmdRawDeviceCopy",
                 "qudaAmdSetSolutionSink", "qudaAmdCommStats", "qudaAmdDescribeHaloError", "qudaAmdMultigridOrthoFallbackBlocks", "qudaAmdProfileMarker", "qudaAmdAccountStart", "qudaAmdAccountDump", "qudaAmdWriteSpinorFields", "qudaAmdReadSpinorFields", "qudaAmdMultigridRefine", "qudaAmdMultigridSetFused", "qudaAmdMultigridFusedStats", "qudaAmdMultiSrcStats"]

_lib = None


class QudaAmdSourceParam(C.Structure):
    """include/quda_amd_ext.h: the source description of the QKXTM solve loop"""
    _fields_ = [("sourcePosition", C.c_int * 4), ("nsmearGauss", C.c_int), ("alphaGauss", C.c_double)]


def lib():
    """The loaded libquda.so; raises if the HIP extension has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("%s is missing: build it with `make -C %s` (or __graft_entry__.build()); "
                               "there is no CPU fallback" % (LIB_PATH, _HERE))
        L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        L.newQudaGaugeParam.restype = QudaGaugeParam
        L.newQudaInvertParam.restype = QudaInvertParam
        L.newQudaMultigridParam.restype = QudaMultigridParam
        L.newMultigridQuda.restype = _p
        L.newMultigridQuda.argtypes = [C.POINTER(QudaMultigridParam)]
        L.destroyMultigridQuda.argtypes = [_p]
        for name in ("qudaAmdSpinorCreate", "qudaAmdDiracCreate", "qudaAmdComputeStream"):
            getattr(L, name).restype = _p
        L.qudaAmdSpinorCreate.argtypes = [_i, _i, _i]
        L.qudaAmdSpinorDestroy.argtypes = [_p]
        L.qudaAmdSpinorLoad.argtypes = [_p, _p, C.POINTER(QudaInvertParam)]
        L.qudaAmdSpinorSave.argtypes = [_p, _p, C.POINTER(QudaInvertParam)]
        L.qudaAmdSpinorCopy.argtypes = [_p, _p]
        L.qudaAmdSpinorSetTwist.argtypes = [_p, _i]
        L.qudaAmdDiracCreate.argtypes = [C.POINTER(QudaInvertParam), _i, _i]
        L.qudaAmdDiracDestroy.argtypes = [_p]
        L.qudaAmdDiracDslash.argtypes = [_p, _p, _p, _i]
        L.qudaAmdDiracDslashXpay.argtypes = [_p, _p, _p, _i, _p, _d]
        for name in ("qudaAmdDiracM", "qudaAmdDiracMdag", "qudaAmdDiracMdagM"):
            getattr(L, name).argtypes = [_p, _p, _p]
        L.qudaAmdDiracFlops.argtypes = [_p]
        L.qudaAmdDiracFlops.restype = C.c_ulonglong
        L.qudaAmdTimeDslash.argtypes = [_p, _p, _p, _i, _i]
        L.qudaAmdTimeDslash.restype = _d
        L.qudaAmdTimeM.argtypes = [_p, _p, _p, _i]
        L.qudaAmdTimeM.restype = _d
        L.qudaAmdBlasNorm2.argtypes = [_p]
        L.qudaAmdBlasNorm2.restype = _d
        L.qudaAmdBlasCDot.argtypes = [_p, _p, C.POINTER(_d)]
        L.qudaAmdBlasAxpy.argtypes = [_d, _p, _p]
        L.qudaAmdDslashBytesPerSite.argtypes = [C.POINTER(QudaInvertParam), _i, _i]
        L.qudaAmdDslashBytesPerSite.restype = C.c_longlong
        L.qudaAmdDslashFlopsPerSite.argtypes = [C.POINTER(QudaInvertParam), _i]
        L.qudaAmdDslashFlopsPerSite.restype = C.c_longlong
        L.dslashQuda.argtypes = [_p, _p, C.POINTER(QudaInvertParam), _i]
        L.MatQuda.argtypes = [_p, _p, C.POINTER(QudaInvertParam)]
        L.MatDagMatQuda.argtypes = [_p, _p, C.POINTER(QudaInvertParam)]
        L.invertQuda.argtypes = [_p, _p, C.POINTER(QudaInvertParam)]
        L.invertMultiSrcQuda.argtypes = [_p, _p, C.POINTER(QudaInvertParam)]
        L.invertMultiSrcQuda.restype = None
        L.cloverQuda.argtypes = [_p, _p, C.POINTER(QudaInvertParam), C.POINTER(_i), _i]
        L.loadGaugeQuda.argtypes = [_p, C.POINTER(QudaGaugeParam)]
        L.loadCloverQuda.argtypes = [_p, _p, C.POINTER(QudaInvertParam)]
        L.qudaAmdCommInit.argtypes = [_p, _i, _i]
        L.qudaAmdCommGetUniqueId.argtypes = [_p]
        L.initCommsGridQuda.argtypes = [_i, C.POINTER(_i), _p, _p]
        L.qudaAmdSetPartitionMask.argtypes = [_i]
        L.qudaAmdSpinorRawInfo.argtypes = [_p, C.POINTER(C.c_longlong)]
        L.qudaAmdGaugeRawInfo.argtypes = [_i, C.POINTER(C.c_longlong)]
        L.qudaAmdCloverRawInfo.argtypes = [_i, C.POINTER(C.c_longlong)]
        L.qudaAmdRawDeviceCopy.argtypes = [_p, C.c_longlong, C.c_size_t]
        L.qudaAmdSetDslashTune.argtypes = [C.c_char_p, _i]
        L.qudaAmdMultigridVerify.argtypes = [_p, C.POINTER(_d)]
        L.qudaAmdMultigridCycle.argtypes = [_p, _p, _p, C.POINTER(QudaInvertParam)]
        L.qudaAmdTimeAxpy.argtypes = [_d, _p, _p, _i]
        L.qudaAmdTimeAxpy.restype = _d
        L.qudaAmdMultigridSetHalfStorage.argtypes = [_p, _i]
        L.qudaAmdMultigridLevels.argtypes = [_p]
        L.qudaAmdMultigridOrthoFallbackBlocks.argtypes = [_p, _i]
        L.qudaAmdMultigridRefine.argtypes = [_p, _i, _i]
        L.qudaAmdMultigridRefine.restype = _d
        L.qudaAmdProfileMarker.argtypes = [_i]
        L.qudaAmdAccountDump.argtypes = [C.c_char_p]
        L.qudaAmdCommStats.argtypes = [C.POINTER(C.c_longlong)]
        L.qudaAmdDescribeHaloError.argtypes = [C.c_char_p, _i]
        L.qudaAmdMultigridLevelInfo.argtypes = [_p, _i, C.POINTER(_i)]
        L.qudaAmdMultigridGetNullVector.argtypes = [_p, _i, _i, _p]
        L.qudaAmdMultigridGetV.argtypes = [_p, _i, _p]
        L.qudaAmdMultigridGetCoarseLinks.argtypes = [_p, _i, _p, _p]
        L.qudaAmdMultigridApply.argtypes = [_p, _i, _i, _p, _p]
        L.qudaAmdMultigridSetFused.argtypes = [_i]
        L.qudaAmdMultigridSetFused.restype = None
        L.qudaAmdMultigridFusedStats.argtypes = [_p, _i, _p]
        L.qudaAmdMultigridFusedStats.restype = _i
        L.qudaAmdMultigridApplyBlock.argtypes = [_p, _i, _i, _p, _p, _i]
        L.qudaAmdMultigridApplyBlock.restype = _d
        L.qudaAmdMultigridTimeApply.argtypes = [_p, _i, _i]
        L.qudaAmdMultigridTimeApply.restype = _d
        L.qudaAmdMultigridTimeTransfer.argtypes = [_p, _i, _i, _i]
        L.qudaAmdMultigridTimeTransfer.restype = _d
        L.qudaAmdSetExitLine.argtypes = [C.c_char_p, _i]
        L.qudaAmdSetExitLine.restype = None
        L.qudaAmdDiracPrepare.argtypes = [_p, _p, _p, _p, _i]
        L.qudaAmdDiracPrepare.restype = None
        L.qudaAmdDiracReconstruct.argtypes = [_p, _p, _p, _i]
        L.qudaAmdDiracReconstruct.restype = None
        L.qudaAmdReadLimeGauge.argtypes = [C.POINTER(_p), C.c_char_p, C.POINTER(QudaGaugeParam), C.POINTER(QudaInvertParam), C.POINTER(_i)]
        L.qudaAmdWriteLimeGauge.argtypes = [C.POINTER(_p), C.c_char_p, C.POINTER(QudaGaugeParam), C.c_char_p]
        L.plaqQuda.argtypes = [C.POINTER(_d)]
        L.performAPEnStep.argtypes = [C.c_uint, _d]
        L.saveGaugeQuda.argtypes = [_p, C.POINTER(QudaGaugeParam)]
        L.qudaAmdSaveSmearedGauge.argtypes = [C.POINTER(_p), _i]
        L.qudaAmdGaussianSmear.argtypes = [_p, _p, C.POINTER(_p), _i, _d]
        L.qudaAmdCalcMGPropagators.argtypes = [_p, _p, C.POINTER(_p), C.POINTER(QudaInvertParam), C.POINTER(QudaAmdSourceParam)]
        _lib = L
    return _lib


def _vp(a):
    return a.ctypes.data_as(_p) if a is not None else None


# ---- helpers written the way the reference's tests set things up (tests/dslash_test.cpp:84-200) ----
def gauge_param(X, cpu_prec=QUDA_DOUBLE_PRECISION, cuda_prec=QUDA_DOUBLE_PRECISION, recon=QUDA_RECONSTRUCT_NO,
                prec_sloppy=None, recon_sloppy=None, prec_precondition=None, t_boundary=QUDA_ANTI_PERIODIC_T, recon_precondition=None):
    gp = lib().newQudaGaugeParam()
    for d in range(4):
        gp.X[d] = int(X[d])
    gp.anisotropy = 1.0
    gp.type = QUDA_WILSON_LINKS
    gp.gauge_order = QUDA_QDP_GAUGE_ORDER
    gp.t_boundary = t_boundary
    gp.cpu_prec = cpu_prec
    gp.cuda_prec = cuda_prec
    gp.reconstruct = recon
    gp.cuda_prec_sloppy = prec_sloppy or cuda_prec
    gp.reconstruct_sloppy = recon_sloppy or recon
    gp.cuda_prec_precondition = prec_precondition or gp.cuda_prec_sloppy
    gp.reconstruct_precondition = recon_precondition or gp.reconstruct_sloppy
    gp.gauge_fix = QUDA_GAUGE_FIXED_NO
    gp.ga_pad = 0
    return gp


def invert_param(dslash_type=QUDA_TWISTED_MASS_DSLASH, kappa=0.1, mu=0.01, flavor=QUDA_TWIST_PLUS, matpc="ee", dagger=0,
                 cpu_prec=QUDA_DOUBLE_PRECISION, cuda_prec=QUDA_DOUBLE_PRECISION, prec_sloppy=None, prec_precondition=None,
                 solution_type=QUDA_MATPC_SOLUTION, gamma_basis=QUDA_DEGRAND_ROSSI_GAMMA_BASIS, dirac_order=QUDA_DIRAC_ORDER):
    ip = lib().newQudaInvertParam()
    ip.dslash_type = dslash_type
    ip.kappa = kappa
    ip.mu = mu
    ip.epsilon = 0.0
    ip.mass = 0.5 / kappa - 4.0
    ip.twist_flavor = flavor if dslash_type != QUDA_WILSON_DSLASH else QUDA_TWIST_NO
    ip.matpc_type = MATPC[matpc] if isinstance(matpc, str) else matpc
    ip.dagger = dagger
    ip.solution_type = solution_type
    ip.solve_type = QUDA_DIRECT_PC_SOLVE
    ip.mass_normalization = QUDA_KAPPA_NORMALIZATION
    ip.cpu_prec = cpu_prec
    ip.cuda_prec = cuda_prec
    ip.cuda_prec_sloppy = prec_sloppy or cuda_prec
    ip.cuda_prec_precondition = prec_precondition or ip.cuda_prec_sloppy
    ip.gamma_basis = gamma_basis
    ip.dirac_order = dirac_order
    ip.clover_cpu_prec = cpu_prec
    ip.clover_cuda_prec = cuda_prec
    ip.clover_cuda_prec_sloppy = ip.cuda_prec_sloppy
    ip.clover_cuda_prec_precondition = ip.cuda_prec_precondition
    ip.clover_order = QUDA_PACKED_CLOVER_ORDER
    ip.clover_coeff = 0.0
    ip.input_location = ip.output_location = QUDA_CPU_FIELD_LOCATION
    ip.tune = QUDA_TUNE_NO
    ip.sp_pad = ip.cl_pad = 0
    ip.verbosity = QUDA_SILENT
    ip.inv_type = QUDA_GCR_INVERTER
    ip.inv_type_precondition = QUDA_INVALID_ENUM
    ip.tol = 1e-10
    ip.maxiter = 1000
    ip.reliable_delta = 1e-4
    ip.gcrNkrylov = 20
    ip.use_init_guess = QUDA_USE_INIT_GUESS_NO
    ip.preserve_source = QUDA_PRESERVE_SOURCE_YES
    ip.residual_type = QUDA_L2_RELATIVE_RESIDUAL
    return ip


COMM_STATS_KEYS = ("fine_peer_store_exchanges", "fine_staged_exchanges", "coarse_peer_store_exchanges", "coarse_staged_exchanges",
                   "in_kernel_allreduces", "collective_allreduces", "fallbacks_to_staged", "block_exchanges")


def comm_stats():
    """this process' transport counters since initQuda (include/quda_amd_ext.h qudaAmdCommStats)"""
    a = (C.c_longlong * 8)()
    lib().qudaAmdCommStats(a)
    return dict(zip(COMM_STATS_KEYS, [int(v) for v in a]))


def halo_error_text():
    buf = C.create_string_buffer(1024)
    return buf.value.decode() if lib().qudaAmdDescribeHaloError(buf, 1024) else None


def init(device=0, verbosity=QUDA_SILENT):
    lib().setVerbosityQuda(verbosity, b"", None)
    lib().initQuda(int(device))


def end():
    lib().endQuda()


def load_gauge(gauge, gp):
    """gauge: (4, V*18) float64/float32 array in QDP order (even sites then odd)."""
    gauge = np.ascontiguousarray(gauge)
    ptrs = (_p * 4)(*[gauge[d].ctypes.data_as(_p) for d in range(4)])
    lib().loadGaugeQuda(C.cast(ptrs, _p), C.byref(gp))
    return gp


def load_clover(clover, clover_inv, ip):
    """loadCloverQuda; clover = clover_inv = None: the clover term is built on the device from the resident links with
    ip.clover_coeff (reference lib/interface_quda.cpp:743-747)"""
    lib().loadCloverQuda(_vp(clover) if clover is not None else None, _vp(clover_inv) if clover_inv is not None else None, C.byref(ip))


def dslash(h_in, ip, parity):
    out = np.empty_like(h_in)
    lib().dslashQuda(_vp(out), _vp(h_in), C.byref(ip), int(parity))
    return out


def mat(h_in, ip):
    out = np.empty_like(h_in)
    lib().MatQuda(_vp(out), _vp(h_in), C.byref(ip))
    return out


def matdagmat(h_in, ip):
    out = np.empty_like(h_in)
    lib().MatDagMatQuda(_vp(out), _vp(h_in), C.byref(ip))
    return out


def invert(h_b, ip, out=None):
    """invertQuda; out: an existing solution array of the caller (as a C caller has one) — a fresh numpy array is untouched memory, and its page
    faults (0.1 s for the 2 GB solution of a 48^3 x 96 lattice) would be charged to the download of the solution"""
    x = np.zeros_like(h_b) if out is None else out
    lib().invertQuda(_vp(x), _vp(h_b), C.byref(ip))
    return x


def multi_src_stats():
    """qudaAmdMultiSrcStats: counters of the lockstep multi-source path since start-up"""
    a = (C.c_longlong * 4)()
    lib().qudaAmdMultiSrcStats(a)
    return dict(block_cycles=int(a[0]), block_smoothed=int(a[1]), quad_transfers=int(a[2]), solves=int(a[3]))


def invert_multi_src(h_bs, ip, out=None):
    """invertMultiSrcQuda: the sources h_bs[i] through one lockstep solve; returns the list of solutions (ip.num_src is set here).
    out: a list of arrays to receive the solutions (as invert(..., out=))"""
    bs = [np.ascontiguousarray(b) for b in h_bs]
    xs = out if out is not None else [np.zeros_like(b) for b in bs]
    if len(xs) != len(bs) or any(x.shape != b.shape or x.dtype != b.dtype or not x.flags.c_contiguous for x, b in zip(xs, bs)):
        raise ValueError("out must hold one contiguous array per source, of the source's shape and type")
    n = len(bs)
    ip.num_src = n
    pb = (_p * n)(*[_vp(b) for b in bs])
    px = (_p * n)(*[_vp(x) for x in xs])
    lib().invertMultiSrcQuda(px, pb, C.byref(ip))
    return xs


def read_lime_gauge(fname, gp, grid=(1, 1, 1, 1), ip=None, local_volume=None):
    """qudaAmdReadLimeGauge: this rank's sub-block of an ILDG configuration as (4, V_local*18) QDP even-odd arrays; sets gp.X"""
    if local_volume is None:
        raise ValueError("local_volume (sites of this rank's sub-lattice) is needed to size the arrays")
    out = np.zeros((4, int(local_volume) * 18))
    ptr = (_p * 4)(*[_vp(out[d]) for d in range(4)])
    lib().qudaAmdReadLimeGauge(ptr, str(fname).encode(), C.byref(gp), C.byref(ip) if ip is not None else None, (_i * 4)(*[int(v) for v in grid]))
    return out


def write_lime_gauge(fname, gauge, gp, xlf_info=None):
    keep = [np.ascontiguousarray(gauge[d], dtype=np.float64) for d in range(4)]
    ptr = (_p * 4)(*[_vp(a) for a in keep])
    lib().qudaAmdWriteLimeGauge(ptr, str(fname).encode(), C.byref(gp), xlf_info.encode() if xlf_info else None)


def plaquette():
    """plaqQuda: (total, spatial, temporal) plaquette averages of the resident links"""
    pl = (_d * 3)()
    lib().plaqQuda(pl)
    return tuple(pl)


def perform_ape(n_steps, alpha):
    lib().performAPEnStep(int(n_steps), float(alpha))


def save_gauge(gp):
    """saveGaugeQuda: the resident links in host QDP order, (4, V*18) float64"""
    V = int(np.prod([gp.X[d] for d in range(4)]))
    out = np.zeros((4, V * 18))
    ptr = (_p * 4)(*[_vp(out[d]) for d in range(4)])
    lib().saveGaugeQuda(C.cast(ptr, _p), C.byref(gp))
    return out


def save_smeared_gauge(local_volume, lexicographic=False):
    out = np.zeros((4, int(local_volume) * 18))
    ptr = (_p * 4)(*[_vp(out[d]) for d in range(4)])
    lib().qudaAmdSaveSmearedGauge(ptr, int(bool(lexicographic)))
    return out


def _lex_links(gauge_lex):
    if gauge_lex is None:
        return None, None
    keep = [np.ascontiguousarray(gauge_lex[d], dtype=np.float64) for d in range(4)]
    return keep, (_p * 4)(*[_vp(a) for a in keep])


def gaussian_smear(vec_lex, gauge_lex, nsmear, alpha):
    """QKXTM_Vector_Kepler::gaussianSmearing on a lexicographic UKQCD host vector (local lattice of the resident gauge field)"""
    vec_lex = np.ascontiguousarray(vec_lex, dtype=np.float64)
    out = np.empty_like(vec_lex)
    keep, links = _lex_links(gauge_lex)
    lib().qudaAmdGaussianSmear(_vp(out), _vp(vec_lex), links, int(nsmear), float(alpha))
    return out


def calc_mg_propagators(gauge_lex, ip, source_position, nsmear, alpha, local_volume):
    """the solve loop of calcMG_threepTwop_EvenOdd: returns (prop_up, prop_dn), each (12, V*24) lexicographic UKQCD vectors"""
    sp = QudaAmdSourceParam()
    for k in range(4):
        sp.sourcePosition[k] = int(source_position[k])
    sp.nsmearGauss, sp.alphaGauss = int(nsmear), float(alpha)
    up = np.zeros((12, int(local_volume) * 24))
    dn = np.zeros((12, int(local_volume) * 24))
    keep, links = _lex_links(gauge_lex)
    lib().qudaAmdCalcMGPropagators(_vp(up), _vp(dn), links, C.byref(ip), C.byref(sp))
    return up, dn


class Spinor:
    """Device-resident ColorSpinorField handle (quda_amd_ext.h)."""

    def __init__(self, prec, subset=QUDA_PARITY_SITE_SUBSET, flavor=QUDA_TWIST_PLUS):
        self.h = lib().qudaAmdSpinorCreate(int(prec), int(subset), int(flavor))
        self.prec, self.subset = prec, subset

    def load(self, host, ip):
        lib().qudaAmdSpinorLoad(self.h, _vp(np.ascontiguousarray(host)), C.byref(ip))
        return self

    def save(self, ip, like):
        out = np.empty_like(like)
        lib().qudaAmdSpinorSave(self.h, _vp(out), C.byref(ip))
        return out

    def norm2(self):
        return lib().qudaAmdBlasNorm2(self.h)

    def raw_info(self):
        """qudaAmdSpinorRawInfo as a dict"""
        a = (C.c_longlong * 20)()
        lib().qudaAmdSpinorRawInfo(self.h, a)
        keys = ("volume", "volumeCB", "stride", "pad", "nSpin", "nColor", "precision", "fieldOrder", "siteSubset", "gammaBasis", "bytes", "norm_bytes",
                "v", "norm", "odd_offset", "odd_norm_offset", "N", "twistFlavor", "x0", "location")
        return dict(zip(keys, [int(v) for v in a]))

    def free(self):
        if self.h:
            lib().qudaAmdSpinorDestroy(self.h)
            self.h = None


def raw_device_copy(address, nbytes):
    """bytes of device memory as a uint8 array (layout tests)"""
    out = np.zeros(int(nbytes), dtype=np.uint8)
    lib().qudaAmdRawDeviceCopy(_vp(out), int(address), int(nbytes))
    return out


def gauge_raw_info(which=0):
    a = (C.c_longlong * 12)()
    lib().qudaAmdGaugeRawInfo(int(which), a)
    return dict(zip(("data", "bytes", "stride", "link_bytes", "precision", "reconstruct", "Vh", "tbc_folded", "t_boundary"), [int(v) for v in a][:9]))


def clover_raw_info(which=0):
    a = (C.c_longlong * 12)()
    lib().qudaAmdCloverRawInfo(int(which), a)
    return dict(zip(("A", "Ainv", "norm", "invNorm", "stride", "parity_bytes", "parity_norm_bytes", "precision", "bytes", "Vh", "twisted"), [int(v) for v in a][:11]))


class Dirac:
    """Operator handle: Dirac::create over the resident gauge/clover (quda_amd_ext.h)."""

    def __init__(self, ip, pc=True, which=0):
        self.h = lib().qudaAmdDiracCreate(C.byref(ip), int(pc), int(which))

    def dslash(self, out, inp, parity):
        lib().qudaAmdDiracDslash(self.h, out.h, inp.h, int(parity))

    def dslash_xpay(self, out, inp, parity, x, k):
        lib().qudaAmdDiracDslashXpay(self.h, out.h, inp.h, int(parity), x.h, float(k))

    def M(self, out, inp):
        lib().qudaAmdDiracM(self.h, out.h, inp.h)

    def Mdag(self, out, inp):
        lib().qudaAmdDiracMdag(self.h, out.h, inp.h)

    def MdagM(self, out, inp):
        lib().qudaAmdDiracMdagM(self.h, out.h, inp.h)

    def prepare(self, src_out, x, b, solution_type):
        """Dirac::prepare on resident full fields; the source of the (preconditioned) system is copied into src_out"""
        lib().qudaAmdDiracPrepare(self.h, src_out.h, x.h, b.h, int(solution_type))

    def reconstruct(self, x, b, solution_type):
        lib().qudaAmdDiracReconstruct(self.h, x.h, b.h, int(solution_type))

    def time_dslash(self, out, inp, parity, niter):
        return lib().qudaAmdTimeDslash(self.h, out.h, inp.h, int(parity), int(niter))

    def time_M(self, out, inp, niter):
        return lib().qudaAmdTimeM(self.h, out.h, inp.h, int(niter))

    def free(self):
        if self.h:
            lib().qudaAmdDiracDestroy(self.h)
            self.h = None


def multigrid_param(ip, n_level=2, geo_block=(4, 4, 4, 4), n_vec=24, nu_pre=2, nu_post=2, cycle=QUDA_MG_CYCLE_RECURSIVE, smoother_tol=0.25,
                    setup_maxiter=500, setup_tol=5e-6, generate_all_levels=True, omega=0.85, smoother_pc=False, coarse_matpc=False):
    """QudaMultigridParam filled the way the reference harness does (tests/multigrid_invert_test.cpp:195-290), with the
    smoother on the full operator (QUDA_DIRECT_SOLVE) or, smoother_pc=True, the reference default QUDA_DIRECT_PC_SOLVE."""
    mp = lib().newQudaMultigridParam()
    mp.invert_param = C.pointer(ip)
    mp.n_level = n_level
    blocks = geo_block if isinstance(geo_block[0], (tuple, list)) else [geo_block] * n_level
    nv = n_vec if isinstance(n_vec, (tuple, list)) else [n_vec] * n_level
    for i in range(n_level):
        for d in range(4):
            mp.geo_block_size[i][d] = int(blocks[i][d])
        for d in range(4, QUDA_MAX_DIM):
            mp.geo_block_size[i][d] = 1
        mp.spin_block_size[i] = 2 if i == 0 else 1
        mp.n_vec[i] = int(nv[i])
        mp.nu_pre[i], mp.nu_post[i] = nu_pre, nu_post
        mp.cycle_type[i] = cycle
        mp.smoother[i] = QUDA_MR_INVERTER
        mp.smoother_tol[i] = smoother_tol
        mp.global_reduction[i] = QUDA_BOOLEAN_YES
        mp.smoother_solve_type[i] = QUDA_DIRECT_PC_SOLVE if smoother_pc else QUDA_DIRECT_SOLVE
        # QUDA_MATPC_SOLUTION: single-parity injection, what the harness pairs with an outer even-odd solve (multigrid_invert_test.cpp:246-252)
        mp.coarse_grid_solution_type[i] = QUDA_MATPC_SOLUTION if coarse_matpc else QUDA_MAT_SOLUTION
        mp.omega[i] = omega
        mp.location[i] = QUDA_CUDA_FIELD_LOCATION
    mp.setup_maxiter, mp.setup_tol = setup_maxiter, setup_tol
    mp.compute_null_vector = QUDA_COMPUTE_NULL_VECTOR_YES
    mp.generate_all_levels = QUDA_BOOLEAN_YES if generate_all_levels else QUDA_BOOLEAN_NO
    mp.run_verify = QUDA_BOOLEAN_NO
    return mp


class Multigrid:
    """newMultigridQuda / destroyMultigridQuda handle"""

    def __init__(self, mp):
        self.mp = mp
        self.h = lib().newMultigridQuda(C.byref(mp))

    def verify(self):
        dev = (_d * 3)()
        lib().qudaAmdMultigridVerify(self.h, dev)
        return [dev[0], dev[1], dev[2]]

    def cycle(self, h_b, ip):
        x = np.zeros_like(h_b)
        lib().qudaAmdMultigridCycle(self.h, _vp(x), _vp(h_b), C.byref(ip))
        return x

    def set_half_storage(self, on=True):
        lib().qudaAmdMultigridSetHalfStorage(self.h, int(bool(on)))

    # ---- introspection (include/quda_amd_ext.h): reference CPU orders, fp32 complex ----
    def levels(self):
        return lib().qudaAmdMultigridLevels(self.h)

    def level_info(self, level):
        a = (_i * 18)()
        lib().qudaAmdMultigridLevelInfo(self.h, level, a)
        v = list(a)
        return dict(Xf=v[0:4], Xc=v[4:8], fineSpin=v[8], fineColor=v[9], Nvec=v[10], geo_bs=v[11:15], spin_bs=v[15], null_method=v[16], null_iters=v[17])

    def refine(self, passes=1, cycles=1):
        """set-up refinement: inverse iteration of the null vectors through the current hierarchy, hierarchy rebuilt; seconds spent"""
        return float(lib().qudaAmdMultigridRefine(self.h, int(passes), int(cycles)))

    def ortho_fallback_blocks(self, level):
        return int(lib().qudaAmdMultigridOrthoFallbackBlocks(self.h, int(level)))

    def null_vector(self, level, k):
        i = self.level_info(level)
        out = np.zeros((int(np.prod(i["Xf"])), i["fineSpin"], i["fineColor"]), dtype=np.complex64)
        lib().qudaAmdMultigridGetNullVector(self.h, level, k, _vp(out))
        return out

    def V(self, level):
        i = self.level_info(level)
        out = np.zeros((int(np.prod(i["Xf"])), i["fineSpin"], i["fineColor"], i["Nvec"]), dtype=np.complex64)
        lib().qudaAmdMultigridGetV(self.h, level, _vp(out))
        return out

    def coarse_links(self, level):
        i = self.level_info(level)
        Vc, n = int(np.prod(i["Xc"])), 2 * i["Nvec"]
        Y = np.zeros((8, Vc, n, n), dtype=np.complex64)
        X = np.zeros((Vc, n, n), dtype=np.complex64)
        lib().qudaAmdMultigridGetCoarseLinks(self.h, level, _vp(Y), _vp(X))
        return Y, X

    def apply_block(self, level, h_in, niter=0):
        """M of `level` on a batch (nrhs, sites, spin, colour) complex64 through the multi-right-hand-side kernels — the MFMA coarse
        operator on a coarse level, the 8/16/24/32-right-hand-side stencil (Wilson / twisted mass / twisted clover) on level 0;
        returns (out, seconds per application if niter > 0)"""
        h_in = np.ascontiguousarray(h_in, dtype=np.complex64)
        out = np.zeros_like(h_in)
        secs = lib().qudaAmdMultigridApplyBlock(self.h, int(level), int(h_in.shape[0]), _vp(out), _vp(h_in), int(niter))
        return out, secs

    def time_apply(self, level, niter=20):
        return lib().qudaAmdMultigridTimeApply(self.h, int(level), int(niter))

    def time_transfer(self, level, what, niter=20):
        """seconds per restriction (what = 'R') or prolongation ('P') between level and level + 1"""
        return lib().qudaAmdMultigridTimeTransfer(self.h, int(level), 1 if what == 'P' else 0, int(niter))

    def apply(self, level, op, h_in):
        """op 'R' (level -> level+1), 'P' (level+1 -> level), 'M' (operator of `level`); fields as (sites, spin, colour) complex64"""
        i = self.level_info(level) if op not in ("M", "K") or level < self.levels() - 1 else None
        if i is None:
            j = self.level_info(level - 1)
            fine_shape = (int(np.prod(j["Xc"])), 2, j["Nvec"])
            coarse_shape = None
        else:
            fine_shape = (int(np.prod(i["Xf"])), i["fineSpin"], i["fineColor"])
            coarse_shape = (int(np.prod(i["Xc"])), 2, i["Nvec"])
        shape_in, shape_out = {"R": (fine_shape, coarse_shape), "P": (coarse_shape, fine_shape), "M": (fine_shape, fine_shape), "K": (fine_shape, fine_shape)}[op]
        h_in = np.ascontiguousarray(h_in, dtype=np.complex64).reshape(shape_in)
        out = np.zeros(shape_out, dtype=np.complex64)
        lib().qudaAmdMultigridApply(self.h, level, {"R": 0, "P": 1, "M": 2, "K": 3}[op], _vp(out), _vp(h_in))
        return out

    def fused_stats(self, level):
        """None if `level` does not run its cycle as one persistent kernel (coarse_cycle.h), else the last launch's counters"""
        a = (C.c_longlong * 5)()
        if not lib().qudaAmdMultigridFusedStats(self.h, int(level), a):
            return None
        return dict(barriers=int(a[0]), gcr_iters=int(a[1]), gcr_restarts=int(a[2]), halo_exchanges=int(a[3]), grid=int(a[4]))

    def free(self):
        if self.h:
            lib().destroyMultigridQuda(self.h)
            self.h = None
