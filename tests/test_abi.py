"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol include/*.h declares,
and the param structs have the reference's binary layout."""
import ctypes as C
import importlib
import json
import os
import re
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
qa = importlib.import_module("quda-qkxtm-multigrid_amd")
GOLD = os.path.join(ROOT, "tests", "golden", "abi_layout.json")

PROBE = r'''
#include <quda.h>
#include <stddef.h>
#include <stdio.h>
#define O(S,f) printf("\"" #S "." #f "\": %zu,\n", offsetof(S,f))
int main(){
 printf("{\n\"sizeof.QudaGaugeParam\": %zu,\n\"sizeof.QudaInvertParam\": %zu,\n\"sizeof.QudaMultigridParam\": %zu,\n\"sizeof.QudaEigParam\": %zu,\n", sizeof(QudaGaugeParam), sizeof(QudaInvertParam), sizeof(QudaMultigridParam), sizeof(QudaEigParam));
 O(QudaGaugeParam,X);O(QudaGaugeParam,anisotropy);O(QudaGaugeParam,type);O(QudaGaugeParam,t_boundary);O(QudaGaugeParam,cpu_prec);O(QudaGaugeParam,reconstruct);O(QudaGaugeParam,reconstruct_precondition);O(QudaGaugeParam,ga_pad);O(QudaGaugeParam,gaugeGiB);O(QudaGaugeParam,i_mu);O(QudaGaugeParam,return_result_mom);
 O(QudaInvertParam,dslash_type);O(QudaInvertParam,kappa);O(QudaInvertParam,Ls);O(QudaInvertParam,c_5);O(QudaInvertParam,mu);O(QudaInvertParam,twist_flavor);O(QudaInvertParam,tol);O(QudaInvertParam,true_res);O(QudaInvertParam,maxiter);O(QudaInvertParam,reliable_delta);O(QudaInvertParam,pipeline);O(QudaInvertParam,offset);O(QudaInvertParam,true_res_hq_offset);
 O(QudaInvertParam,solution_type);O(QudaInvertParam,solve_type);O(QudaInvertParam,matpc_type);O(QudaInvertParam,dagger);O(QudaInvertParam,mass_normalization);O(QudaInvertParam,cpu_prec);O(QudaInvertParam,cuda_prec_precondition);O(QudaInvertParam,dirac_order);O(QudaInvertParam,gamma_basis);O(QudaInvertParam,clover_cpu_prec);O(QudaInvertParam,clover_order);O(QudaInvertParam,use_init_guess);O(QudaInvertParam,clover_coeff);
 O(QudaInvertParam,trlogA);O(QudaInvertParam,return_clover_inverse);O(QudaInvertParam,verbosity);O(QudaInvertParam,iter);O(QudaInvertParam,spinorGiB);O(QudaInvertParam,secs);O(QudaInvertParam,tune);O(QudaInvertParam,gcrNkrylov);O(QudaInvertParam,inv_type_precondition);O(QudaInvertParam,preconditioner);O(QudaInvertParam,preconditionerUP);O(QudaInvertParam,preconditionerDN);
 O(QudaInvertParam,dslash_type_precondition);O(QudaInvertParam,tol_precondition);O(QudaInvertParam,omega);O(QudaInvertParam,precondition_cycle);O(QudaInvertParam,residual_type);O(QudaInvertParam,eigenval_tol);O(QudaInvertParam,inc_tol);O(QudaInvertParam,use_resident_solution);
 O(QudaMultigridParam,invert_param);O(QudaMultigridParam,n_level);O(QudaMultigridParam,geo_block_size);O(QudaMultigridParam,spin_block_size);O(QudaMultigridParam,n_vec);O(QudaMultigridParam,smoother);O(QudaMultigridParam,coarse_grid_solution_type);O(QudaMultigridParam,smoother_solve_type);O(QudaMultigridParam,cycle_type);O(QudaMultigridParam,nu_post);O(QudaMultigridParam,smoother_tol);O(QudaMultigridParam,setup_maxiter);O(QudaMultigridParam,setup_tol);O(QudaMultigridParam,omega);
 O(QudaMultigridParam,global_reduction);O(QudaMultigridParam,location);O(QudaMultigridParam,compute_null_vector);O(QudaMultigridParam,run_verify);O(QudaMultigridParam,vec_infile);O(QudaMultigridParam,vec_outfile);O(QudaMultigridParam,gflops);O(QudaMultigridParam,delta_muPR);O(QudaMultigridParam,delta_cswCG);
 printf("\"enum.QUDA_TWISTED_CLOVER_DSLASH\": %d,\n\"enum.QUDA_MG_INVERTER\": %d,\n\"enum.QUDA_QDP_GAUGE_ORDER\": %d,\n\"enum.QUDA_TWISTED_CLOVERPC_DIRAC\": %d,\n\"enum.QUDA_COARSEPC_DIRAC\": %d,\n\"enum.QUDA_SPACE_SPIN_COLOR_FIELD_ORDER\": %d,\n\"enum.QUDA_PACKED_CLOVER_ORDER\": %d,\n\"enum.QUDA_MATPC_ODD_ODD_ASYMMETRIC\": %d,\n\"enum.QUDA_MG_CYCLE_RECURSIVE\": %d,\n\"enum.QUDA_DIRECT_PC_SOLVE\": %d,\n\"enum.QUDA_MATPCDAG_MATPC_SOLUTION\": %d,\n\"enum.QUDA_UKQCD_GAMMA_BASIS\": %d,\n\"enum.QUDA_TWIST_DEG_DOUBLET\": %d,\n\"enum.QUDA_DEBUG_VERBOSE\": %d,\n\"enum.QUDA_INVALID_ENUM\": %d\n}\n",
   QUDA_TWISTED_CLOVER_DSLASH, QUDA_MG_INVERTER, QUDA_QDP_GAUGE_ORDER, QUDA_TWISTED_CLOVERPC_DIRAC, QUDA_COARSEPC_DIRAC, QUDA_SPACE_SPIN_COLOR_FIELD_ORDER, QUDA_PACKED_CLOVER_ORDER, QUDA_MATPC_ODD_ODD_ASYMMETRIC, QUDA_MG_CYCLE_RECURSIVE, QUDA_DIRECT_PC_SOLVE, QUDA_MATPCDAG_MATPC_SOLUTION, QUDA_UKQCD_GAMMA_BASIS, QUDA_TWIST_DEG_DOUBLET, QUDA_DEBUG_VERBOSE, QUDA_INVALID_ENUM);
 return 0;}
'''


def _probe(include_dir):
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "p.c")
        open(src, "w").write(PROBE)
        subprocess.check_call(["gcc", "-I", include_dir, src, "-o", os.path.join(d, "p")])
        return json.loads(subprocess.check_output([os.path.join(d, "p")]).decode())


def test_struct_layout_matches_reference_fixture():
    """tests/golden/abi_layout.json was produced by running the probe against the REFERENCE's include/quda.h
    (see test_regenerate_fixture_from_reference); this repo's header must give identical offsets/values."""
    want = json.load(open(GOLD))
    got = _probe(os.path.join(ROOT, "include"))
    assert got == want


@pytest.mark.skipif(not os.path.isdir("/root/reference/include"), reason="reference tree not present on this box")
def test_regenerate_fixture_from_reference():
    got = _probe("/root/reference/include")
    if os.environ.get("QUDA_AMD_WRITE_FIXTURES"):
        json.dump(got, open(GOLD, "w"), indent=1, sort_keys=True)
    assert got == json.load(open(GOLD))


def test_ctypes_mirrors_have_the_same_size():
    want = json.load(open(GOLD))
    assert C.sizeof(qa.QudaGaugeParam) == want["sizeof.QudaGaugeParam"]
    assert C.sizeof(qa.QudaInvertParam) == want["sizeof.QudaInvertParam"]
    assert C.sizeof(qa.QudaMultigridParam) == want["sizeof.QudaMultigridParam"]
    assert qa.QudaInvertParam.preconditioner.offset == want["QudaInvertParam.preconditioner"]
    assert qa.QudaInvertParam.use_resident_solution.offset == want["QudaInvertParam.use_resident_solution"]
    assert qa.QudaMultigridParam.delta_cswCG.offset == want["QudaMultigridParam.delta_cswCG"]
    assert qa.QudaGaugeParam.gaugeGiB.offset == want["QudaGaugeParam.gaugeGiB"]


def _declared(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    txt = re.sub(r"^\s*typedef[^;]*\(\s*\*\s*\w+\s*\)[^;]*;", "", txt, flags=re.M)   # function-pointer typedefs declare no symbol
    return set(re.findall(r"^\s*(?:[A-Za-z_][\w\s\*]*?)\b(\w+)\s*\([^;{]*\)\s*;", txt, flags=re.M)) - {"defined"}


def test_library_exports_every_declared_symbol():
    """No compute call here (no GPU needed): dlopen + dlsym of everything the headers declare."""
    L = qa.lib()
    declared = (_declared("quda.h") | _declared("quda_amd_ext.h")) - {"QudaCommsMap", "int"}
    assert set(qa.QUDA_H_SYMBOLS) <= declared and set(qa.EXT_H_SYMBOLS) <= declared
    missing = [s for s in sorted(declared) if not hasattr(L, s)]
    assert not missing, missing
    assert len(declared) >= 49


def test_param_constructors_use_invalid_sentinels():
    gp = qa.lib().newQudaGaugeParam()
    ip = qa.lib().newQudaInvertParam()
    mp = qa.lib().newQudaMultigridParam()
    assert gp.cpu_prec == qa.QUDA_INVALID_ENUM and gp.X[0] == qa.QUDA_INVALID_ENUM
    assert ip.dslash_type == qa.QUDA_INVALID_ENUM and ip.kappa != ip.kappa  # NaN
    assert mp.n_level == qa.QUDA_INVALID_ENUM and mp.delta_muPR == 1.0
