"""Launch-parameter cache (csrc/tune.cpp; reference lib/tune.cpp:213-355): the table is persisted as tunecache.tsv under QUDA_RESOURCE_PATH
in the reference's text format — header line `tunecache <version> <gitversion> <hash> # Last updated ...`, a blank line, the column
description, then one tab-separated entry per line (volume, name, aux, block.x y z, grid.x y z, shared_bytes, aux.x y z w, time, comment).
Host-only code: runs without a GPU.  The GPU half (a sweep fills the table, a second start reads it and sweeps nothing) is
tests/test_dslash_gpu.py::test_tune_cache_is_filled_by_a_sweep_and_read_back."""
import ctypes as C
import importlib
import os

import pytest


@pytest.fixture()
def qa():
    return importlib.import_module("quda-qkxtm-multigrid_amd")


def _proto(L):
    L.qudaAmdTuneCacheLoad.restype = C.c_int
    L.qudaAmdTuneCacheStore.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_int), C.c_float, C.c_char_p]
    L.qudaAmdTuneCacheLookup.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_float)]
    L.qudaAmdTuneCacheLookup.restype = C.c_int


def test_tunecache_file_format_and_round_trip(qa, tmp_path, monkeypatch):
    L = qa.lib()
    _proto(L)
    monkeypatch.setenv("QUDA_RESOURCE_PATH", str(tmp_path))
    assert L.qudaAmdTuneCacheLoad() == 0
    entries = {
        (b"32x32x32x32", b"dslash_kernel", b"prec=8,recon=18,mode=1,xpay=0,dagger=0,comm=0000"): ([256, 1, 1, 2048, 1, 1, 0, 1, 0, 2, 1], 1.42e-4, b"# 142.00 us; yg1 st0=146.1 yg1 st2=142.0"),
        (b"48x48x48x96", b"dslash_kernel<clover>", b"prec=2,recon=12,mode=3,xpay=1,dagger=1,comm=0111"): ([192, 1, 1, 27648, 1, 1, 0, 2, 2, 0, 0], 3.81e-4, b"# with spaces and\ttabs kept out"),
    }
    for (v, n, a), (prm, t, cm) in entries.items():
        L.qudaAmdTuneCacheStore(v, n, a, (C.c_int * 11)(*prm), t, cm.replace(b"\t", b" "))
    L.qudaAmdTuneCacheSave()
    path = tmp_path / "tunecache.tsv"
    assert path.exists() and not (tmp_path / "tunecache.lock").exists()
    lines = path.read_text().split("\n")
    head = lines[0].split("\t")
    assert head[0] == "tunecache" and len(head) >= 5 and head[4].startswith("# Last updated")
    assert lines[1] == ""                                  # ctime's own newline plus std::endl, as the reference writes it
    cols = lines[2].split("\t")
    assert cols[0].strip() == "volume" and cols[1:] == ["name", "aux", "block.x", "block.y", "block.z", "grid.x", "grid.y", "grid.z", "shared_bytes", "aux.x", "aux.y", "aux.z", "aux.w", "time", "comment"]
    body = [ln.split("\t") for ln in lines[3:] if ln]
    assert len(body) == 2
    for row in body:
        key = (row[0].strip().encode(), row[1].encode(), row[2].encode())
        prm, t, cm = entries[key]
        assert len(row[0]) == 16                            # std::setw(16), right-aligned
        assert [int(x) for x in row[3:14]] == prm
        assert abs(float(row[14]) - t) < 1e-9 and row[15].encode() == cm.replace(b"\t", b" ")
    # a fresh load (what the next process start does) finds the same table
    assert L.qudaAmdTuneCacheLoad() == 2
    for (v, n, a), (prm, t, cm) in entries.items():
        out, tt = (C.c_int * 11)(), C.c_float()
        assert L.qudaAmdTuneCacheLookup(v, n, a, out, C.byref(tt)) == 1
        assert list(out) == prm and abs(tt.value - t) < 1e-9
    assert L.qudaAmdTuneCacheLookup(b"8x8x8x8", b"dslash_kernel", b"nothing", (C.c_int * 11)(), None) == 0
    # a table of another build is ignored, not trusted (the reference aborts and asks for its deletion)
    path.write_text(path.read_text().replace("gfx950-dslash", "other-build", 1))
    assert L.qudaAmdTuneCacheLoad() == 0
    monkeypatch.delenv("QUDA_RESOURCE_PATH")
    assert L.qudaAmdTuneCacheLoad() == 0
