"""ILDG / LIME configuration reader and writer (csrc/lime_io.cpp; reference qkxtm/QKXTM_read_conf.h on top of c-lime).  Host-only
code, so it runs without a GPU: the reader is fed a file assembled byte by byte here from the published container format
(144-byte big-endian record headers, data padded to 8 bytes) and the ILDG site order, independently of the library's writer;
then the library's own writer must produce a file with the same payload bytes."""
import importlib
import struct

import numpy as np
import pytest


@pytest.fixture(scope="module")
def qa():
    return importlib.import_module("quda-qkxtm-multigrid_amd")


def lime_record(rtype, data, mb, me):
    head = struct.pack(">IHHQ", 0x456789AB, 1, (0x8000 if mb else 0) | (0x4000 if me else 0), len(data)) + rtype.encode().ljust(128, b"\0")
    assert len(head) == 144
    return head + data + b"\0" * (-len(data) % 8)


def ildg_payload(X):
    """value of every real number from its ILDG coordinates: site (t, z, y, x) with x fastest, then mu, then 18 reals"""
    t, z, y, x, mu, k = np.meshgrid(*[np.arange(n) for n in (X[3], X[2], X[1], X[0], 4, 18)], indexing="ij")
    return (((t * 100 + z) * 100 + y) * 100 + x) * 100.0 + mu * 20 + k + 0.25


def expected_qdp(vals, X):
    """QDP even-odd arrays of loadGaugeQuda: gauge[mu][(parity*Vh + lexicographic/2)*18 + k] (tests/test_util.cpp:419-443)"""
    V = int(np.prod(X))
    out = np.zeros((4, V * 18))
    lex = vals.reshape(V, 4, 18)
    t, z, y, x = np.meshgrid(*[np.arange(n) for n in (X[3], X[2], X[1], X[0])], indexing="ij")
    parity = ((t + z + y + x) & 1).reshape(-1)
    iv = np.arange(V)
    dst = parity * (V // 2) + iv // 2
    for mu in range(4):
        o = out[mu].reshape(V, 18)
        o[dst] = lex[:, mu]
    return out


@pytest.mark.parametrize("X", [(4, 4, 4, 4), (6, 4, 2, 8)])
def test_reader_against_hand_built_file(qa, tmp_path, X):
    vals = ildg_payload(X)
    xml = ("<?xml version=\"1.0\" encoding=\"UTF-8\"?><ildgFormat><version>1.0</version><field>su3gauge</field><precision>64</precision>"
           "<lx>%d</lx><ly>%d</ly><lz>%d</lz><lt>%d</lt></ildgFormat>" % tuple(X)).encode()
    blob = (lime_record("xlf-info", b"plaquette = 0.5, kappa = 0.137000, mu = 0.0040", True, True)
            + lime_record("ildg-format", xml, True, False)
            + lime_record("ildg-binary-data", vals.astype(">f8").tobytes(), False, True)
            + lime_record("scidac-checksum", b"<scidacChecksum/>", True, True))
    path = tmp_path / "conf.lime"
    path.write_bytes(blob)
    gp = qa.lib().newQudaGaugeParam()
    ip = qa.lib().newQudaInvertParam()
    ip.kappa = 0.137
    got = qa.read_lime_gauge(path, gp, (1, 1, 1, 1), ip, int(np.prod(X)))
    assert [gp.X[d] for d in range(4)] == list(X)
    assert np.array_equal(got, expected_qdp(vals, X))


def test_writer_produces_the_same_payload(qa, tmp_path):
    X = (4, 6, 2, 4)
    vals = ildg_payload(X)
    gauge = expected_qdp(vals, X)
    gp = qa.lib().newQudaGaugeParam()
    for d in range(4):
        gp.X[d] = X[d]
    path = tmp_path / "out.lime"
    qa.write_lime_gauge(path, gauge, gp, "kappa = 0.125000, mu = 0.01")
    raw = path.read_bytes()
    # walk the records with the format description
    pos, records = 0, {}
    while pos < len(raw):
        magic, version, flags, nbytes = struct.unpack(">IHHQ", raw[pos:pos + 16])
        assert magic == 0x456789AB and version == 1
        rtype = raw[pos + 16:pos + 144].split(b"\0")[0].decode()
        records[rtype] = raw[pos + 144:pos + 144 + nbytes]
        pos += 144 + (nbytes + 7) // 8 * 8
    assert pos == len(raw) and list(records) == ["xlf-info", "ildg-format", "ildg-binary-data"]
    assert records["ildg-binary-data"] == vals.astype(">f8").tobytes()
    assert b"<precision>64</precision>" in records["ildg-format"] and b"<lx>4</lx><ly>6</ly><lz>2</lz><lt>4</lt>" in records["ildg-format"]
    # and back through the reader
    gp2 = qa.lib().newQudaGaugeParam()
    assert np.array_equal(qa.read_lime_gauge(path, gp2, (1, 1, 1, 1), None, int(np.prod(X))), gauge)


# ---- vector files: the SciDAC / QIO single-file container the reference's write_spinor_field / read_spinor_field use
# (lib/qio_field.cpp:198-328; MG::saveVectors / loadVectors, lib/multigrid.cpp:607-691) ----
def _walk(raw):
    pos, records = 0, []
    while pos < len(raw):
        magic, version, flags, nbytes = struct.unpack(">IHHQ", raw[pos:pos + 16])
        assert magic == 0x456789AB and version == 1
        records.append((raw[pos + 16:pos + 144].split(b"\0")[0].decode(), raw[pos + 144:pos + 144 + nbytes], flags))
        pos += 144 + (nbytes + 7) // 8 * 8
    assert pos == len(raw)
    return records


def _vectors(X, nvec, nreal, seed):
    """even-odd ordered host fields whose every real encodes (vector, global lexicographic site, component)"""
    V = int(np.prod(X))
    t, z, y, x = np.meshgrid(*[np.arange(n) for n in (X[3], X[2], X[1], X[0])], indexing="ij")
    lex = (((t * X[2] + z) * X[1] + y) * X[0] + x).reshape(-1)
    eo = ((t + z + y + x) & 1).reshape(-1) * (V // 2) + lex // 2
    out = []
    for v in range(nvec):
        f = np.zeros((V, nreal), dtype=np.float32)
        f[eo] = (lex[:, None] * 8 + v) + np.arange(nreal)[None, :] / 128.0 + seed
        out.append(f)
    return out, lex, eo


@pytest.mark.parametrize("X,nspin,ncolor,nvec", [((4, 4, 4, 4), 4, 3, 3), ((6, 4, 2, 8), 2, 8, 5)])
def test_scidac_vector_file_records_and_round_trip(qa, tmp_path, X, nspin, ncolor, nvec):
    import ctypes as C
    import zlib
    nreal = 2 * nspin * ncolor
    V = int(np.prod(X))
    fields, lex, eo = _vectors(X, nvec, nreal, 0.5)
    path = str(tmp_path / "vecs_level_0").encode()
    ptrs = (C.c_void_p * nvec)(*[f.ctypes.data for f in fields])
    Xc = (C.c_int * 4)(*X)
    qa.lib().qudaAmdWriteSpinorFields(path, ptrs, 4, Xc, ncolor, nspin, nvec)
    recs = _walk(open(path, "rb").read())
    assert [r[0] for r in recs] == ["scidac-private-file-xml", "scidac-file-xml", "scidac-private-record-xml", "scidac-record-xml", "scidac-binary-data", "scidac-checksum"]
    assert ("<dims>%d %d %d %d </dims>" % tuple(X)).encode() in recs[0][1] and b"<volfmt>0</volfmt>" in recs[0][1]
    assert recs[1][1].rstrip(b"\0") == b"Dummy user file XML"
    assert ("<datatype>QUDA_FNs%dNc%d_ColorSpinorField</datatype><precision>F</precision><colors>%d</colors><spins>%d</spins><typesize>%d</typesize><datacount>%d</datacount>"
            % (nspin, ncolor, ncolor, nspin, 4 * nreal, nvec)).encode() in recs[2][1]
    # payload: global lexicographic sites, per site vector 0 .. nvec-1, big-endian fp32 — built here independently of the writer
    want = np.zeros((V, nvec, nreal), dtype=">f4")
    for v in range(nvec):
        want[lex, v] = fields[v][eo]
    assert recs[4][1] == want.tobytes()
    # checksum as QIO accumulates it: CRC-32 of every site's bytes rotated by (site rank mod 29 / mod 31), XORed
    suma = sumb = 0
    for s in range(V):
        c = zlib.crc32(want[s].tobytes()) & 0xFFFFFFFF
        r29, r31 = s % 29, s % 31
        suma ^= ((c << r29) | (c >> (32 - r29))) & 0xFFFFFFFF
        sumb ^= ((c << r31) | (c >> (32 - r31))) & 0xFFFFFFFF
    assert ("<suma>%x</suma><sumb>%x</sumb>" % (suma, sumb)).encode() in recs[5][1]
    # back through the reader, fewer vectors than the file holds, into fp64 memory
    back = [np.zeros((V, nreal)) for _ in range(nvec - 1)]
    bptrs = (C.c_void_p * (nvec - 1))(*[f.ctypes.data for f in back])
    qa.lib().qudaAmdReadSpinorFields(path, bptrs, 8, Xc, ncolor, nspin, nvec - 1)
    for v in range(nvec - 1):
        assert np.array_equal(back[v], fields[v].astype(np.float64))


def test_scidac_double_precision_records(qa, tmp_path):
    """fp64 fields are written as 'D' records (datatype QUDA_DNs.., typesize 8 * reals: reference lib/qio_field.cpp:305-314) and a 'D' file
    — also one assembled here byte by byte — loads into fp32 or fp64 memory (the file's precision comes from its record, :73-125)"""
    import ctypes as C
    X, nspin, ncolor, nvec = (4, 4, 2, 4), 4, 3, 2
    nreal = 2 * nspin * ncolor
    V = int(np.prod(X))
    fields32, lex, eo = _vectors(X, nvec, nreal, 0.25)
    fields = [f.astype(np.float64) + 1e-9 * (1 + i) for i, f in enumerate(fields32)]   # not representable in fp32
    path = str(tmp_path / "dvecs").encode()
    Xc = (C.c_int * 4)(*X)
    ptrs = (C.c_void_p * nvec)(*[f.ctypes.data for f in fields])
    qa.lib().qudaAmdWriteSpinorFields(path, ptrs, 8, Xc, ncolor, nspin, nvec)
    recs = _walk(open(path, "rb").read())
    assert ("<datatype>QUDA_DNs%dNc%d_ColorSpinorField</datatype><precision>D</precision><colors>%d</colors><spins>%d</spins><typesize>%d</typesize><datacount>%d</datacount>"
            % (nspin, ncolor, ncolor, nspin, 8 * nreal, nvec)).encode() in recs[2][1]
    want = np.zeros((V, nvec, nreal), dtype=">f8")
    for v in range(nvec):
        want[lex, v] = fields[v][eo]
    assert recs[4][1] == want.tobytes()
    back = [np.zeros((V, nreal)) for _ in range(nvec)]
    bptrs = (C.c_void_p * nvec)(*[f.ctypes.data for f in back])
    qa.lib().qudaAmdReadSpinorFields(path, bptrs, 8, Xc, ncolor, nspin, nvec)
    for v in range(nvec):
        assert np.array_equal(back[v], fields[v])          # bit-exact in fp64
    back32 = [np.zeros((V, nreal), dtype=np.float32) for _ in range(nvec)]
    bptrs = (C.c_void_p * nvec)(*[f.ctypes.data for f in back32])
    qa.lib().qudaAmdReadSpinorFields(path, bptrs, 4, Xc, ncolor, nspin, nvec)
    for v in range(nvec):
        assert np.array_equal(back32[v], fields[v].astype(np.float32))
    # the same file put together here from the format description, without the library's writer
    hand = str(tmp_path / "dvecs_by_hand").encode()
    with open(hand, "wb") as f:
        for name, data, mb, me in [("scidac-private-file-xml", recs[0][1], True, False), ("scidac-file-xml", b"user\0", False, True),
                                   ("scidac-private-record-xml", recs[2][1], True, False), ("scidac-record-xml", b"rec\0", False, False),
                                   ("scidac-binary-data", want.tobytes(), False, False), ("scidac-checksum", recs[5][1], False, True)]:
            f.write(lime_record(name, data, mb, me))
    back = [np.zeros((V, nreal)) for _ in range(nvec)]
    bptrs = (C.c_void_p * nvec)(*[f.ctypes.data for f in back])
    qa.lib().qudaAmdReadSpinorFields(hand, bptrs, 8, Xc, ncolor, nspin, nvec)
    for v in range(nvec):
        assert np.array_equal(back[v], fields[v])
