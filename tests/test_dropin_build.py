"""The drop-in boundary, checked without a GPU: callers written against the reference compile against include/ and link
against libquda.so.

  * the reference's OWN test program tests/multigrid_invert_test.cpp and the helper sources it links, unmodified, from where
    they lie under /root/reference (build container only; skipped where the tree is absent) — compiled with include/ in place of
    the reference's include/ (oracle/Makefile target `dropin`), and syntax-checked with both include directories on the path
    (the combination in which a folded enum_quda.h used to clash);
  * the committed plain-C driver and C++ consumer under tests/consumer/ (run on the GPU by tests/test_dropin_gpu.py).
"""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INC = os.path.join(ROOT, "include")
LIBDIR = os.path.join(ROOT, "quda-qkxtm-multigrid_amd", "lib")
REF = "/root/reference"

needs_lib = pytest.mark.skipif(not os.path.exists(os.path.join(LIBDIR, "libquda.so")), reason="libquda.so not built")
needs_ref = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "tests")), reason="reference tree not present (GPU box)")


def build_consumers(outdir):
    """gcc / g++ builds of tests/consumer/* against include/ + libquda.so; returns the two executables"""
    rpath = ["-Wl,-rpath," + LIBDIR, "-Wl,-rpath,/opt/rocm/lib"]
    cdrv = os.path.join(outdir, "c_driver")
    subprocess.run(["gcc", "-std=c99", "-O1", "-Wall", "-Werror", "-I", INC, os.path.join(ROOT, "tests", "consumer", "c_driver.c"), "-o", cdrv,
                    "-L" + LIBDIR, "-lquda", "-lm"] + rpath, check=True, capture_output=True, text=True)
    cxx = os.path.join(outdir, "cxx_consumer")
    subprocess.run(["g++", "-std=c++17", "-O1", "-D__HIP_PLATFORM_AMD__", "-I", INC, "-I", "/opt/rocm/include",
                    os.path.join(ROOT, "tests", "consumer", "cxx_consumer.cpp"), "-o", cxx, "-L" + LIBDIR, "-lquda", "-L/opt/rocm/lib", "-lamdhip64"] + rpath,
                   check=True, capture_output=True, text=True)
    return cdrv, cxx


def build_qkxtm_driver(outdir):
    """tests/consumer/qkxtm_driver.cpp: includes <qudaQKXTM_Kepler.h> and calls the reference's calcMG_* entry points by name"""
    exe = os.path.join(outdir, "qkxtm_driver")
    subprocess.run(["g++", "-std=c++17", "-O1", "-D__HIP_PLATFORM_AMD__", "-I", INC, "-I", "/opt/rocm/include",
                    os.path.join(ROOT, "tests", "consumer", "qkxtm_driver.cpp"), "-o", exe, "-L" + LIBDIR, "-lquda", "-L/opt/rocm/lib", "-lamdhip64",
                    "-Wl,-rpath," + LIBDIR, "-Wl,-rpath,/opt/rocm/lib"], check=True, capture_output=True, text=True)
    return exe


@needs_lib
def test_committed_c_and_cxx_consumers_build(tmp_path):
    cdrv, cxx = build_consumers(str(tmp_path))
    assert os.path.exists(cdrv) and os.path.exists(cxx)
    assert os.path.exists(build_qkxtm_driver(str(tmp_path)))


@needs_lib
def test_qkxtm_entry_points_are_exported_under_the_reference_names():
    """calcMG_threepTwop_EvenOdd / calcMG_loop_wOneD_TSM_EvenOdd / calcMG_loop_wOneD_TSM_wExact with the reference's C++ signatures
    (include/qudaQKXTM_Kepler.h:484-508): the mangled names a driver compiled against the reference header would ask for"""
    sym = subprocess.run(["nm", "-DC", os.path.join(LIBDIR, "libquda.so")], capture_output=True, text=True).stdout
    assert "calcMG_threepTwop_EvenOdd(void**, void**, QudaGaugeParam_s*, QudaInvertParam_s*, quda::qudaQKXTMinfo_Kepler, char*, char*, quda::WHICHPARTICLE)" in sym
    assert "calcMG_loop_wOneD_TSM_EvenOdd(void**, QudaInvertParam_s*, QudaGaugeParam_s*, quda::qudaQKXTM_loopInfo, quda::qudaQKXTMinfo_Kepler)" in sym
    assert "calcMG_loop_wOneD_TSM_wExact(void**, QudaInvertParam_s*, QudaInvertParam_s*, QudaGaugeParam_s*, quda::qudaQKXTM_arpackInfo, quda::qudaQKXTM_loopInfo, quda::qudaQKXTMinfo_Kepler)" in sym


def test_public_headers_are_self_contained_c(tmp_path):
    """quda.h / enum_quda.h / quda_constants.h / comm_quda.h / quda_amd_ext.h each compile on their own as C99"""
    for h in ("quda.h", "enum_quda.h", "quda_constants.h", "comm_quda.h", "quda_amd_ext.h"):
        src = tmp_path / ("t_" + h.replace(".", "_") + ".c")
        src.write_text("#include <%s>\nint main(void) { return 0; }\n" % h)
        subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-fsyntax-only", "-I", INC, str(src)], check=True, capture_output=True, text=True)


@needs_ref
@needs_lib
def test_reference_mg_test_program_builds_against_this_library():
    r = subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "dropin"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    exe = os.path.join(ROOT, "oracle", "_ref", "mg_invert_test")
    out = subprocess.run([exe, "--help"], capture_output=True, text=True)
    assert "--mg-levels" in out.stdout
    # no symbol of the program is left to a library other than libquda / libc / libstdc++ / libm
    und = subprocess.run(["nm", "-uC", exe], capture_output=True, text=True).stdout
    for sym in ("newMultigridQuda", "invertQuda", "loadGaugeQuda", "comm_dim_partitioned"):
        assert sym in und


@needs_ref
def test_reference_mg_test_source_with_both_include_dirs():
    """repo include/ in front of the reference's include/: <enum_quda.h> and <quda_constants.h> must resolve to ONE definition set"""
    try:
        import triton
        cuda_inc = os.path.join(os.path.dirname(triton.__file__), "backends", "nvidia", "include")
    except Exception:
        pytest.skip("no CUDA headers for the reference's own include files")
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    r = subprocess.run(["g++", "-std=c++11", "-fsyntax-only", "-w", "-I", INC, "-I", os.path.join(REF, "include"), "-I", os.path.join(REF, "tests"), "-I", cuda_inc,
                        os.path.join(REF, "tests", "multigrid_invert_test.cpp")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
