"""SURVEY 8a-7, the ColorSpinorField layout contract, pinned on the RAW device image (not through a load -> save round trip):
a field is uploaded through the C ABI, its device bytes are copied back untouched (qudaAmdRawDeviceCopy) and every real is looked
up where the reference's formulas put it (lib/color_spinor_field.cpp:129-216; planar orders include/color_spinor_field_order.h /
lib/io_spinor.h:1-62):

  stride = volumeCB + pad; per-parity bytes rounded up to TEX_ALIGN_REQ = 1 KiB; odd half at bytes / 2; real (spin s, colour c,
  re/im r) of checkerboard site x at plane-entry index  k = ((s * 3 + c) * 2 + r),  plane k / N, slot k % N:
      element index  (k / N) * stride * N + x * N + k % N          N = 2 (fp64, FLOAT2), 4 (fp32, FLOAT4)
  16-bit: int16 fixed point with one fp32 scale per site in a norm array of stride floats per parity (norm_bytes rounded to 1 KiB).

Two documented deviations: the device spin basis is DeGrand-Rossi, so a host field in that basis lands
unrotated; 16-bit planes hold N = 8 values (16-byte vectors) instead of the reference's short4.
Also pinned: the bidirectional link blocks (fields.h) and the clover planes (two chiral blocks of 36 reals per site)."""
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def qa():
    mod = importlib.import_module("quda-qkxtm-multigrid_amd")
    mod.init(0)
    yield mod
    mod.end()


def _align(n, a=1024):
    return (n + a - 1) // a * a


@pytest.mark.parametrize("prec", [8, 4, 2])
@pytest.mark.parametrize("subset", ["parity", "full"])
def test_spinor_raw_layout(qa, oracle, prec, subset):
    X = [6, 4, 2, 8]
    gauge, spinor, _ = oracle.make_fields(X, seed=5, clover=False)
    qa.load_gauge(gauge, qa.gauge_param(X, cuda_prec=prec))
    V = int(np.prod(X))
    Vh = V // 2
    full = subset == "full"
    ip = qa.invert_param(qa.QUDA_TWISTED_MASS_DSLASH, 0.1, 0.01, +1, "ee", 0, cuda_prec=prec,
                         solution_type=qa.QUDA_MAT_SOLUTION if full else qa.QUDA_MATPC_SOLUTION)
    host = (np.random.default_rng(8).random(V * 24 if full else Vh * 24) - 0.5)
    f = qa.Spinor(prec, qa.QUDA_FULL_SITE_SUBSET if full else qa.QUDA_PARITY_SITE_SUBSET)
    try:
        f.load(host, ip)
        i = f.raw_info()
        nsub = 2 if full else 1
        N = {8: 2, 4: 4, 2: 8}[prec]
        assert (i["volume"], i["volumeCB"], i["stride"], i["pad"]) == (Vh * nsub, Vh, Vh, 0)
        assert (i["nSpin"], i["nColor"], i["precision"], i["N"]) == (4, 3, prec, N)
        assert i["gammaBasis"] == qa.QUDA_DEGRAND_ROSSI_GAMMA_BASIS
        assert i["fieldOrder"] == (2 if prec == 8 else 4)                      # QUDA_FLOAT2_FIELD_ORDER / QUDA_FLOAT4_FIELD_ORDER
        half_bytes = _align(Vh * 24 * prec)
        assert i["bytes"] == nsub * half_bytes and i["odd_offset"] == (half_bytes if full else 0)
        assert i["v"] % 1024 == 0 and (i["v"] + i["odd_offset"]) % 1024 == 0     # both halves TEX_ALIGN_REQ-aligned
        if prec == 2:
            half_norm = _align(Vh * 4)
            assert i["norm_bytes"] == nsub * half_norm and i["odd_norm_offset"] == (half_norm if full else 0) and i["norm"] % 256 == 0
        else:
            assert i["norm_bytes"] == 0 and i["norm"] == 0
        raw = qa.raw_device_copy(i["v"], i["bytes"])
        norms = qa.raw_device_copy(i["norm"], i["norm_bytes"]).view(np.float32) if prec == 2 else None
        dt = {8: np.float64, 4: np.float32, 2: np.int16}[prec]
        for par in range(nsub):
            img = raw[par * half_bytes: par * half_bytes + Vh * 24 * prec].view(dt)
            want = host.reshape(nsub, Vh, 24)[par]                        # host: site-major (spin, colour, re/im), DeGrand-Rossi
            # gather what the formula says: element (k / N) * stride * N + x * N + k % N
            k = np.arange(24)
            idx = (k // N)[None, :] * (Vh * N) + np.arange(Vh)[:, None] * N + (k % N)[None, :]
            got = img[idx]
            if prec == 8:
                assert np.array_equal(got, want)
            elif prec == 4:
                assert np.array_equal(got, want.astype(np.float32))
            else:
                nrm = norms[par * (half_norm // 4): par * (half_norm // 4) + Vh]
                assert np.allclose(nrm, np.max(np.abs(want), axis=1).astype(np.float32), rtol=1e-6)
                assert np.max(np.abs(got.astype(np.float64) * nrm[:, None] / 32767.0 - want)) <= np.max(nrm) / 32767.0 * 0.51 + 1e-7
                assert np.max(np.abs(got)) == 32767                                       # fixed point uses the full int16 range
    finally:
        f.free()


@pytest.mark.parametrize("prec,recon", [(8, 18), (4, 18), (4, 12), (2, 18)])
def test_gauge_raw_layout(qa, oracle, prec, recon):
    """[parity][direction 0..7][planes][site]: W[p][2 mu](x) = U_mu(x), W[p][2 mu + 1](x) = U_mu(x - mu)^dagger — the eight matrices the
    stencil multiplies with at x, stored at x (fields.h); recon-12 keeps rows 0 and 1"""
    X = [4, 6, 2, 4]
    gauge, _, _ = oracle.make_fields(X, seed=6, antiperiodic_t=False, clover=False)
    qa.load_gauge(gauge, qa.gauge_param(X, cuda_prec=prec, recon=recon, t_boundary=qa.QUDA_PERIODIC_T))
    i = qa.gauge_raw_info(0)
    V = int(np.prod(X)); Vh = V // 2
    N = {8: 2, 4: 4, 2: 8}[prec]
    assert (i["stride"], i["Vh"], i["precision"], i["reconstruct"]) == (Vh, Vh, prec, recon)
    assert i["link_bytes"] >= Vh * recon * prec and i["bytes"] == 16 * i["link_bytes"] and i["data"] % 256 == 0
    raw = qa.raw_device_copy(i["data"], i["bytes"])
    dt = {8: np.float64, 4: np.float32, 2: np.int16}[prec]
    U = gauge.reshape(4, 2, Vh, 3, 3, 2)   # host QDP: [mu][parity][x_cb][row][col][re/im]
    Uc = U[..., 0] + 1j * U[..., 1]
    nbr = oracle.neighbor_table(X) if hasattr(oracle, "neighbor_table") else None
    scale = 1.0 / 32767.0 if prec == 2 else 1.0
    tol = {8: 0.0, 4: 1e-7, 2: 1.0 / 32767.0}[prec]
    for p in range(2):
        for mu in range(4):
            blk = raw[(p * 8 + 2 * mu) * i["link_bytes"]: (p * 8 + 2 * mu) * i["link_bytes"] + Vh * recon * prec].view(dt).astype(np.float64) * scale
            k = np.arange(recon)
            full_planes = (recon // N) * N
            idx = np.where(k[None, :] < full_planes, (k // N)[None, :] * (Vh * N) + np.arange(Vh)[:, None] * N + (k % N)[None, :],
                           full_planes * Vh + np.arange(Vh)[:, None] * (recon - full_planes) + (k - full_planes)[None, :])
            got = blk[idx].reshape(Vh, recon // 6, 3, 2)
            want = U[mu, p][:, : recon // 6]                                  # forward link of the site itself, rows 0.. (all three, or two)
            assert np.max(np.abs(got - want)) <= tol, (p, mu)
    # backward blocks: checked through the operator in tests/test_dslash_gpu.py; here their unitarity partner relation at one site
    p, mu, x = 0, 1, 3
    if recon == 18 and prec == 8:
        blk = raw[(p * 8 + 2 * mu + 1) * i["link_bytes"]: (p * 8 + 2 * mu + 1) * i["link_bytes"] + Vh * 18 * 8].view(np.float64)
        k = np.arange(18)
        w = blk[(k // 2) * (Vh * 2) + x * 2 + k % 2].reshape(3, 3, 2)
        W = w[..., 0] + 1j * w[..., 1]
        # it is the dagger of SOME forward link of the other parity (the one that ends at x)
        cands = Uc[mu, 1 - p]
        assert np.min(np.max(np.abs(cands.conj().transpose(0, 2, 1) - W[None]), axis=(1, 2))) == 0.0


@pytest.mark.parametrize("prec", [8, 4, 2])
def test_clover_raw_layout(qa, oracle, prec):
    """[parity][chiral block 0,1][planes of the 36 packed reals][site] (+ one fp32 scale per (site, block) for 16-bit), host packed order
    tests/clover_reference.cpp:45-53, values un-halved"""
    X = [4, 4, 2, 6]
    gauge, _, clover = oracle.make_fields(X, seed=9, clover=True)
    qa.load_gauge(gauge, qa.gauge_param(X, cuda_prec=prec))
    ip = qa.invert_param(qa.QUDA_TWISTED_CLOVER_DSLASH, 0.1, 0.02, +1, "ee", 0, cuda_prec=prec)
    qa.load_clover(clover, None, ip)
    i = qa.clover_raw_info(0)
    V = int(np.prod(X)); Vh = V // 2
    N = {8: 2, 4: 4, 2: 8}[prec]
    assert (i["stride"], i["Vh"], i["precision"]) == (Vh, Vh, prec) and i["parity_bytes"] >= Vh * 72 * prec and i["A"] % 256 == 0
    raw = qa.raw_device_copy(i["A"], 2 * i["parity_bytes"])
    dt = {8: np.float64, 4: np.float32, 2: np.int16}[prec]
    norms = qa.raw_device_copy(i["norm"], 2 * 2 * Vh * 4).view(np.float32).reshape(2, 2, Vh) if prec == 2 else None
    host = clover.reshape(2, Vh, 2, 36)
    for p in range(2):
        for chi in range(2):
            base = p * i["parity_bytes"] + chi * 36 * prec * Vh
            blk = raw[base: base + Vh * 36 * prec].view(dt)
            k = np.arange(36)
            full_planes = (36 // N) * N
            idx = np.where(k[None, :] < full_planes, (k // N)[None, :] * (Vh * N) + np.arange(Vh)[:, None] * N + (k % N)[None, :],
                           full_planes * Vh + np.arange(Vh)[:, None] * (36 - full_planes) + (k - full_planes)[None, :])
            got = blk[idx].astype(np.float64)
            want = host[p, :, chi, :]
            if prec == 2:
                got = got * norms[p, chi][:, None] / 32767.0
                assert np.max(np.abs(got - want)) <= np.max(norms) / 32767.0 * 0.51 + 1e-7
            else:
                assert np.max(np.abs(got - want)) <= (0.0 if prec == 8 else 1e-7)
