// lime_io.cpp — ILDG gauge configurations in LIME containers, host side (the I/O half).
//
// Reference: readLimeGauge / readLimeGaugeSmeared (qkxtm/QKXTM_read_conf.h:107-400, :819-835) on top of the c-lime library
// (a third-party dependency that is not in the reference tree and not in this image).  The container is restated here from
// its published format (USQCD c-lime 1.3, lime_header.h / lime_fixed_types.h):
//   a LIME file is a sequence of records; every record starts with a 144-byte header, all fields big-endian:
//     bytes 0-3 magic 0x456789ab | 4-5 version (1) | 6-7 flags: bit 15 message-begin, bit 14 message-end |
//     8-15 data length in bytes | 16-143 NUL-padded ASCII type,
//   followed by the data, zero-padded to a multiple of 8 bytes.
// What the reference reads from it (QKXTM_read_conf.h:153-222): the record "ildg-format" (XML: <precision>, <lx> <ly> <lz>
// <lt>), optionally "xlf-info" (kappa / mu, compared with the parameters and only warned about) and the payload
// "ildg-binary-data": big-endian doubles, index (((t*LZ + z)*LY + y)*LX + x)*72 + mu*18 + (row*3 + col)*2 + re/im, i.e. sites
// with x fastest and t slowest, the four directions x, y, z, t inside a site (:321-323, :341-373).  Every rank reads the
// sub-block of its own coordinates (the reference: an MPI subarray view; here plain seeks, one x-row at a time) into the
// even-odd QDP arrays loadGaugeQuda takes, and param->X is set to the local extents (:196-214).  No boundary condition is
// applied (:395-397).  "PARITY UNPINNED" against the reference's own reader (it needs c-lime and MPI-IO); the unit test
// reads a file assembled byte by byte in Python from the format description above.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "qa_core.h"
#include "blas.h"
#include "halo.h"
#include "quda_amd_ext.h"

namespace quda {

static uint64_t be64(const unsigned char *p) { uint64_t v = 0; for (int i = 0; i < 8; i++) v = (v << 8) | p[i]; return v; }
static uint32_t be32(const unsigned char *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
static void put_be(unsigned char *p, uint64_t v, int n) { for (int i = n - 1; i >= 0; i--) { p[i] = (unsigned char)(v & 0xff); v >>= 8; } }

struct LimeRecord { std::string type; uint64_t bytes = 0; long data_offset = 0; };

// next record header at the current position; false at end of file
static bool limeNext(FILE *f, LimeRecord &r, const char *fname) {
  unsigned char h[144];
  const size_t n = fread(h, 1, 144, f);
  if (n == 0) return false;
  if (n != 144) errorQuda("%s: truncated LIME header", fname);
  if (be32(h) != 0x456789abu) errorQuda("%s: not a LIME record (magic %08x)", fname, be32(h));
  r.bytes = be64(h + 8);
  char type[129];
  memcpy(type, h + 16, 128); type[128] = 0;
  r.type = type;
  r.data_offset = ftell(f);
  return true;
}
static void limeSkip(FILE *f, const LimeRecord &r) {
  const uint64_t padded = (r.bytes + 7) / 8 * 8;
  if (fseek(f, r.data_offset + (long)padded, SEEK_SET) != 0) errorQuda("seek failed");
}
static bool xmlInt(const std::string &xml, const char *tag, int &v) {
  const size_t p = xml.find(tag);
  if (p == std::string::npos) return false;
  return sscanf(xml.c_str() + p + strlen(tag), "%d", &v) == 1;
}
static void swap8(double *d, size_t n) {
  unsigned char *p = (unsigned char *)d;
  for (size_t i = 0; i < n; i++, p += 8) { for (int k = 0; k < 4; k++) { const unsigned char t = p[k]; p[k] = p[7 - k]; p[7 - k] = t; } }
}
static bool hostIsBigEndian() { const uint16_t v = 1; return *(const unsigned char *)&v == 0; }

// ================================================================================================
// SciDAC / QIO "single file" container of colour-spinor fields — the container the reference hands its null vectors to
// (MG::saveVectors / loadVectors -> write_spinor_field / read_spinor_field, lib/multigrid.cpp:607-691, lib/qio_field.cpp:198-328:
// QIO_write of ONE field record with datacount = Nvec, QIO_SINGLEFILE, QIO_PARALLEL).  QIO and c-lime are not in the reference tree nor
// in this image; the layout is restated from the published QIO 2.x file format: LIME records
//   scidac-private-file-xml   <scidacFile><version>1.1</version><spacetime>4</spacetime><dims>X Y Z T </dims><volfmt>0</volfmt></scidacFile>
//   scidac-file-xml           the user file string ("Dummy user file XML", lib/qio_field.cpp:38)
//   scidac-private-record-xml <scidacRecord>... <datatype>QUDA_FNs4Nc3_ColorSpinorField</datatype><precision>F</precision><colors>3</colors>
//                             <spins>4</spins><typesize>96</typesize><datacount>Nvec</datacount></scidacRecord>   (lib/qio_field.cpp:243-245, :314)
//   scidac-record-xml         the user record string ("Dummy user record XML for SU(N) field", :219)
//   scidac-binary-data        for every GLOBAL site in lexicographic order (x fastest): datacount x typesize bytes, big-endian — the 24
//                             (2 nSpin nColor) reals of the site in vector 0, then vector 1, ... (vgetM, lib/qio_field.cpp)
//   scidac-checksum           <scidacChecksum><version>1.0</version><suma>%x</suma><sumb>%x</sumb></scidacChecksum>: the CRC-32 of every site's
//                             bytes, rotated left by (site rank mod 29) resp. (mod 31), XORed over the sites (QIO DML_checksum_accum)
// The fields are the level's vectors in the host order the reference holds them in: even-odd site order locally (node_index,
// lib/layout_hyper.c:215-230), (spin, colour, re/im) per site.  "PARITY UNPINNED" against QIO itself (it cannot be built here): pinned
// by a record-by-record check of the bytes in tests/test_lime_io.py and by the round trip.  Every rank writes / reads the rows of
// its own sub-lattice of the one file (plain seeks; QIO_PARALLEL in the reference).
// ================================================================================================
static uint32_t crc32_bytes(const unsigned char *p, size_t n) {
  static uint32_t table[256];
  static bool init = false;
  if (!init) {
    for (uint32_t i = 0; i < 256; i++) { uint32_t c = i; for (int k = 0; k < 8; k++) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1; table[i] = c; }
    init = true;
  }
  uint32_t c = 0xFFFFFFFFu;
  for (size_t i = 0; i < n; i++) c = table[(c ^ p[i]) & 0xff] ^ (c >> 8);
  return c ^ 0xFFFFFFFFu;
}
static void limeWriteHeader(FILE *f, const char *type, uint64_t bytes, bool mb, bool me) {
  unsigned char h[144];
  memset(h, 0, sizeof(h));
  put_be(h, 0x456789abu, 4); put_be(h + 4, 1, 2); put_be(h + 6, (mb ? 0x8000u : 0u) | (me ? 0x4000u : 0u), 2); put_be(h + 8, bytes, 8);
  strncpy((char *)h + 16, type, 127);
  if (fwrite(h, 1, 144, f) != 144) errorQuda("short write of a LIME header");
}
static void limeWriteRecord(FILE *f, const char *type, const void *data, uint64_t bytes, bool mb, bool me) {
  limeWriteHeader(f, type, bytes, mb, me);
  if (fwrite(data, 1, bytes, f) != bytes) errorQuda("short write of a %s record", type);
  const unsigned char zero[8] = {0};
  if (bytes % 8 && fwrite(zero, 1, 8 - bytes % 8, f) != 8 - bytes % 8) errorQuda("short write");
}
static void swap4(float *d, size_t n) {
  unsigned char *p = (unsigned char *)d;
  for (size_t i = 0; i < n; i++, p += 4) { unsigned char t = p[0]; p[0] = p[3]; p[3] = t; t = p[1]; p[1] = p[2]; p[2] = t; }
}
// global checksum from the per-rank partial XORs (an XOR all-reduce through a sum over rank-indexed slots)
static void xorAllreduce(uint32_t &a, uint32_t &b) {
  const CommGrid &g = commGrid();
  if (g.size == 1) return;
  if (2 * g.size > 64) errorQuda("checksum reduction over %d ranks exceeds the host collective's buffer", g.size);
  std::vector<double> v(2 * (size_t)g.size, 0.0);
  v[2 * g.rank] = (double)a; v[2 * g.rank + 1] = (double)b;
  comm_allreduce(v.data(), 2 * g.size);
  a = b = 0;
  for (int r = 0; r < g.size; r++) { a ^= (uint32_t)v[2 * r]; b ^= (uint32_t)v[2 * r + 1]; }
}

bool scidacIsContainer(const char *fname) {
  FILE *f = fopen(fname, "rb");
  if (!f) return false;
  unsigned char h[4];
  const bool ok = fread(h, 1, 4, f) == 4 && be32(h) == 0x456789abu;
  fclose(f);
  return ok;
}

// vecs[i]: host field of the LOCAL lattice X in `prec` (fp32 -> 'F' records, fp64 -> 'D' records, as the reference's write_spinor_field picks
// the file precision from the field, lib/qio_field.cpp:305-314), even-odd site order, nReal = 2 nSpin nColor reals per site
void scidacWriteSpinorsPrec(const char *fname, const std::vector<const void *> &vecs, QudaPrecision prec, const int X[4], int nSpin, int nColor) {
  const CommGrid &g = commGrid();
  const int nvec = (int)vecs.size(), nReal = 2 * nSpin * nColor;
  const size_t es = prec == QUDA_DOUBLE_PRECISION ? sizeof(double) : sizeof(float);
  int G[4], off[4];
  for (int d = 0; d < 4; d++) { G[d] = X[d] * g.dims[d]; off[d] = X[d] * g.coords[d]; }
  const uint64_t gvol = (uint64_t)G[0] * G[1] * G[2] * G[3], siteBytes = (uint64_t)nvec * nReal * es;
  long dataOffset = 0;
  if (g.rank == 0) {
    FILE *f = fopen(fname, "wb");
    if (!f) errorQuda("cannot open %s for writing", fname);
    char xml[1024];
    snprintf(xml, sizeof(xml), "<?xml version=\"1.0\" encoding=\"UTF-8\"?><scidacFile><version>1.1</version><spacetime>4</spacetime><dims>%d %d %d %d </dims><volfmt>0</volfmt></scidacFile>", G[0], G[1], G[2], G[3]);
    limeWriteRecord(f, "scidac-private-file-xml", xml, strlen(xml) + 1, true, false);
    const char *userFile = "Dummy user file XML";
    limeWriteRecord(f, "scidac-file-xml", userFile, strlen(userFile) + 1, false, true);
    snprintf(xml, sizeof(xml), "<?xml version=\"1.0\" encoding=\"UTF-8\"?><scidacRecord><version>1.1</version><date>unknown</date><recordtype>0</recordtype>"
             "<datatype>QUDA_%cNs%dNc%d_ColorSpinorField</datatype><precision>%c</precision><colors>%d</colors><spins>%d</spins><typesize>%d</typesize><datacount>%d</datacount></scidacRecord>",
             es == 8 ? 'D' : 'F', nSpin, nColor, es == 8 ? 'D' : 'F', nColor, nSpin, (int)(nReal * es), nvec);
    limeWriteRecord(f, "scidac-private-record-xml", xml, strlen(xml) + 1, true, false);
    const char *userRec = "Dummy user record XML for SU(N) field";
    limeWriteRecord(f, "scidac-record-xml", userRec, strlen(userRec) + 1, false, false);
    limeWriteHeader(f, "scidac-binary-data", gvol * siteBytes, false, false);
    dataOffset = ftell(f);
    // the payload area exists before any rank seeks into it
    const uint64_t padded = (gvol * siteBytes + 7) / 8 * 8;
    if (fseek(f, dataOffset + (long)padded - 1, SEEK_SET) != 0 || fputc(0, f) == EOF) errorQuda("cannot extend %s to %llu payload bytes", fname, (unsigned long long)padded);
    fclose(f);
  }
  { double o = (double)dataOffset; comm_allreduce(&o, 1); dataOffset = (long)o; }   // rank 0's value (the others contribute 0); also a barrier
  FILE *f = fopen(fname, "r+b");
  if (!f) errorQuda("cannot reopen %s", fname);
  const long Vh = (long)X[0] * X[1] * X[2] * X[3] / 2;
  const size_t rowReals = (size_t)X[0] * nvec * nReal;
  std::vector<unsigned char> row(rowReals * es);
  const bool swap = !hostIsBigEndian();
  uint32_t suma = 0, sumb = 0;
  for (int t = 0; t < X[3]; t++)
    for (int z = 0; z < X[2]; z++)
      for (int y = 0; y < X[1]; y++) {
        for (int x = 0; x < X[0]; x++) {
          const int par = (x + y + z + t) & 1;   // local parity (local extents are even, so it equals the global one)
          const long cb = ((((long)t * X[2] + z) * X[1] + y) * X[0] + x) >> 1;
          for (int v = 0; v < nvec; v++) memcpy(&row[((size_t)x * nvec + v) * nReal * es], (const unsigned char *)vecs[v] + ((size_t)par * Vh + cb) * nReal * es, nReal * es);
        }
        if (swap) { if (es == 8) swap8((double *)row.data(), rowReals); else swap4((float *)row.data(), rowReals); }
        const uint64_t rank0 = (((uint64_t)(t + off[3]) * G[2] + (z + off[2])) * G[1] + (y + off[1])) * G[0] + off[0];
        for (int x = 0; x < X[0]; x++) {
          const uint32_t c = crc32_bytes(&row[(size_t)x * siteBytes], siteBytes);
          const unsigned r29 = (unsigned)((rank0 + x) % 29), r31 = (unsigned)((rank0 + x) % 31);
          suma ^= (c << r29) | (r29 ? c >> (32 - r29) : 0u);
          sumb ^= (c << r31) | (r31 ? c >> (32 - r31) : 0u);
        }
        if (fseek(f, dataOffset + (long)(rank0 * siteBytes), SEEK_SET) != 0 || fwrite(row.data(), 1, row.size(), f) != row.size()) errorQuda("short write on %s", fname);
      }
  fclose(f);
  xorAllreduce(suma, sumb);
  if (g.rank == 0) {
    f = fopen(fname, "r+b");
    if (!f) errorQuda("cannot reopen %s", fname);
    fseek(f, 0, SEEK_END);
    char xml[512];
    snprintf(xml, sizeof(xml), "<?xml version=\"1.0\" encoding=\"UTF-8\"?><scidacChecksum><version>1.0</version><suma>%x</suma><sumb>%x</sumb></scidacChecksum>", suma, sumb);
    limeWriteRecord(f, "scidac-checksum", xml, strlen(xml) + 1, false, true);
    fclose(f);
  }
  commBarrier();
}
void scidacWriteSpinors(const char *fname, const std::vector<const float *> &vecs, const int X[4], int nSpin, int nColor) {
  std::vector<const void *> v(vecs.begin(), vecs.end());
  scidacWriteSpinorsPrec(fname, v, QUDA_SINGLE_PRECISION, X, nSpin, nColor);
}

// fills vecs[0 .. n-1] (n <= datacount of the file) with this rank's sub-lattice; the file's lattice, site size and checksum are checked
void scidacReadSpinorsPrec(const char *fname, const std::vector<void *> &vecs, QudaPrecision memPrec, const int X[4], int nSpin, int nColor) {
  const CommGrid &g = commGrid();
  const int nvec = (int)vecs.size(), nReal = 2 * nSpin * nColor;
  int G[4], off[4];
  for (int d = 0; d < 4; d++) { G[d] = X[d] * g.dims[d]; off[d] = X[d] * g.coords[d]; }
  FILE *f = fopen(fname, "rb");
  if (!f) errorQuda("cannot open %s", fname);
  LimeRecord r;
  long payload = -1;
  int fileCount = 0, typesize = 0, fd[4] = {0, 0, 0, 0};
  uint32_t wantA = 0, wantB = 0;
  bool haveSum = false;
  char prec = 0;
  while (limeNext(f, r, fname)) {
    if (r.type == "scidac-private-file-xml" || r.type == "scidac-private-record-xml" || r.type == "scidac-checksum") {
      std::string data(r.bytes, 0);
      if (fread(&data[0], 1, r.bytes, f) != r.bytes) errorQuda("%s: truncated %s record", fname, r.type.c_str());
      if (r.type == "scidac-private-file-xml") {
        const size_t p = data.find("<dims>");
        if (p == std::string::npos || sscanf(data.c_str() + p + 6, "%d %d %d %d", &fd[0], &fd[1], &fd[2], &fd[3]) != 4) errorQuda("%s: no <dims> in the file record", fname);
      } else if (r.type == "scidac-private-record-xml") {
        if (!xmlInt(data, "<typesize>", typesize) || !xmlInt(data, "<datacount>", fileCount)) errorQuda("%s: no typesize / datacount in the record description", fname);
        const size_t p = data.find("<precision>");
        if (p != std::string::npos) prec = data[p + 11];
      } else {
        const size_t pa = data.find("<suma>"), pb = data.find("<sumb>");
        if (pa != std::string::npos && pb != std::string::npos && sscanf(data.c_str() + pa + 6, "%x", &wantA) == 1 && sscanf(data.c_str() + pb + 6, "%x", &wantB) == 1) haveSum = true;
      }
    } else if (r.type == "scidac-binary-data") {
      payload = r.data_offset;
    }
    limeSkip(f, r);
  }
  if (payload < 0) errorQuda("%s: no scidac-binary-data record", fname);
  for (int d = 0; d < 4; d++) if (fd[d] != G[d]) errorQuda("%s holds a %d x %d x %d x %d lattice, this run has %d x %d x %d x %d", fname, fd[0], fd[1], fd[2], fd[3], G[0], G[1], G[2], G[3]);
  // the file's precision comes from its record description, as the reference's read_field takes it (lib/qio_field.cpp:73-125): 'F' or 'D'
  if ((prec != 'F' && prec != 'D') || typesize != (int)(nReal * (prec == 'D' ? sizeof(double) : sizeof(float))))
    errorQuda("%s: records of precision %c with %d bytes per site and vector (expected F with %d or D with %d bytes)", fname, prec ? prec : '?', typesize, (int)(nReal * sizeof(float)), (int)(nReal * sizeof(double)));
  const size_t es = prec == 'D' ? sizeof(double) : sizeof(float), ms = memPrec == QUDA_DOUBLE_PRECISION ? sizeof(double) : sizeof(float);
  if (fileCount < nvec) errorQuda("%s holds %d vectors, %d are needed", fname, fileCount, nvec);
  if (fileCount > nvec) warningQuda("%s holds %d vectors, this level uses the first %d", fname, fileCount, nvec);
  const uint64_t siteBytes = (uint64_t)fileCount * nReal * es;
  const long Vh = (long)X[0] * X[1] * X[2] * X[3] / 2;
  const size_t rowReals = (size_t)X[0] * fileCount * nReal;
  std::vector<unsigned char> row(rowReals * es);
  const bool swap = !hostIsBigEndian();
  uint32_t suma = 0, sumb = 0;
  for (int t = 0; t < X[3]; t++)
    for (int z = 0; z < X[2]; z++)
      for (int y = 0; y < X[1]; y++) {
        const uint64_t rank0 = (((uint64_t)(t + off[3]) * G[2] + (z + off[2])) * G[1] + (y + off[1])) * G[0] + off[0];
        if (fseek(f, payload + (long)(rank0 * siteBytes), SEEK_SET) != 0 || fread(row.data(), 1, row.size(), f) != row.size()) errorQuda("short read on %s", fname);
        for (int x = 0; x < X[0]; x++) {
          const uint32_t c = crc32_bytes(&row[(size_t)x * siteBytes], siteBytes);
          const unsigned r29 = (unsigned)((rank0 + x) % 29), r31 = (unsigned)((rank0 + x) % 31);
          suma ^= (c << r29) | (r29 ? c >> (32 - r29) : 0u);
          sumb ^= (c << r31) | (r31 ? c >> (32 - r31) : 0u);
        }
        if (swap) { if (es == 8) swap8((double *)row.data(), rowReals); else swap4((float *)row.data(), rowReals); }
        for (int x = 0; x < X[0]; x++) {
          const int par = (x + y + z + t) & 1;
          const long cb = ((((long)t * X[2] + z) * X[1] + y) * X[0] + x) >> 1;
          for (int v = 0; v < nvec; v++) {
            const unsigned char *src = &row[((size_t)x * fileCount + v) * nReal * es];
            unsigned char *dst = (unsigned char *)vecs[v] + ((size_t)par * Vh + cb) * nReal * ms;
            if (es == ms) memcpy(dst, src, nReal * es);
            else if (es == 8) for (int k = 0; k < nReal; k++) ((float *)dst)[k] = (float)((const double *)src)[k];
            else for (int k = 0; k < nReal; k++) ((double *)dst)[k] = (double)((const float *)src)[k];
          }
        }
      }
  fclose(f);
  xorAllreduce(suma, sumb);
  if (haveSum && (suma != wantA || sumb != wantB)) errorQuda("%s: checksum mismatch (file %x %x, data %x %x)", fname, wantA, wantB, suma, sumb);
  if (!haveSum) warningQuda("%s carries no scidac-checksum record", fname);
}
void scidacReadSpinors(const char *fname, const std::vector<float *> &vecs, const int X[4], int nSpin, int nColor) {
  std::vector<void *> v(vecs.begin(), vecs.end());
  scidacReadSpinorsPrec(fname, v, QUDA_SINGLE_PRECISION, X, nSpin, nColor);
}

}  // namespace quda

using namespace quda;

extern "C" {

// read_spinor_field / write_spinor_field of the reference (include/qio_field.h, lib/qio_field.cpp:198-328): Nvec host fields V[i] of the
// local lattice X in even-odd site order, 2 nSpin nColor reals per site, fp32 or fp64 in memory; fp32 in the file
void qudaAmdWriteSpinorFields(const char *filename, void *V[], QudaPrecision precision, const int *X, int nColor, int nSpin, int Nvec) {
  if (precision != QUDA_DOUBLE_PRECISION && precision != QUDA_SINGLE_PRECISION) errorQuda("Error, file_prec=%d not supported", precision);
  std::vector<const void *> ptrs(V, V + Nvec);
  scidacWriteSpinorsPrec(filename, ptrs, precision, X, nSpin, nColor);   // fp64 fields -> 'D' records (QUDA_DNs..), fp32 -> 'F'
}
void qudaAmdReadSpinorFields(const char *filename, void *V[], QudaPrecision precision, const int *X, int nColor, int nSpin, int Nvec) {
  if (precision != QUDA_DOUBLE_PRECISION && precision != QUDA_SINGLE_PRECISION) errorQuda("Error, cpu precision %d not supported", precision);
  std::vector<void *> ptrs(V, V + Nvec);
  scidacReadSpinorsPrec(filename, ptrs, precision, X, nSpin, nColor);     // the record's precision ('F' or 'D') is converted to the caller's
}


void qudaAmdReadLimeGauge(void **gauge, const char *fname, QudaGaugeParam *param, QudaInvertParam *inv_param, const int gridSize[4]) {
  FILE *f = fopen(fname, "rb");
  if (!f) errorQuda("Error reading configuration! Could not open %s for reading", fname);
  int ln[4] = {0, 0, 0, 0}, precision = 0;
  long payload = -1;
  uint64_t payload_bytes = 0;
  LimeRecord r;
  while (limeNext(f, r, fname)) {
    if (r.type == "ildg-binary-data") { payload = r.data_offset; payload_bytes = r.bytes; break; }
    if (r.type == "ildg-format" || r.type == "xlf-info") {
      std::string data(r.bytes, '\0');
      if (fread(&data[0], 1, r.bytes, f) != r.bytes) errorQuda("%s: truncated %s record", fname, r.type.c_str());
      if (r.type == "ildg-format") {
        if (!xmlInt(data, "<precision>", precision) || !xmlInt(data, "<lx>", ln[0]) || !xmlInt(data, "<ly>", ln[1]) || !xmlInt(data, "<lz>", ln[2]) ||
            !xmlInt(data, "<lt>", ln[3]))
          errorQuda("%s: ildg-format record without precision / lx / ly / lz / lt", fname);
      } else if (inv_param) {
        // reference :159-176: report, and warn if kappa differs
        const size_t p = data.find("kappa =");
        double kappa = 0;
        if (p != std::string::npos && sscanf(data.c_str() + p + 7, "%lf", &kappa) == 1) {
          if (getVerbosity() >= QUDA_SUMMARIZE) printfQuda("Kappa given is: %10.8f \t Kappa conf is: %10.8f\n", inv_param->kappa, kappa);
          if (inv_param->kappa != kappa) warningQuda("Kappa given and kappa from configuration do not agree!");
        }
      }
    }
    limeSkip(f, r);
  }
  if (payload < 0) errorQuda("%s: no ildg-binary-data record", fname);
  if (precision == 32) errorQuda("Unsupported precision 32 bits");   // reference :230-233
  if (precision != 64) errorQuda("%s: ildg-format precision %d", fname, precision);
  const CommGrid &cg = commGrid();
  int X[4];
  for (int d = 0; d < 4; d++) {
    if (gridSize[d] < 1 || ln[d] % gridSize[d]) errorQuda("%s: extent %d of dimension %d does not divide over %d ranks", fname, ln[d], d, gridSize[d]);
    if (gridSize[d] != cg.dims[d]) errorQuda("gridSize[%d] = %d but the process grid has %d ranks there", d, gridSize[d], cg.dims[d]);
    X[d] = ln[d] / gridSize[d];
    param->X[d] = X[d];
  }
  const uint64_t lvol = (uint64_t)ln[0] * ln[1] * ln[2] * ln[3];
  if (lvol == 0) errorQuda("Zero volume");
  if (payload_bytes != lvol * 72 * sizeof(double)) errorQuda("%s: ildg-binary-data holds %llu bytes, %llu expected", fname, (unsigned long long)payload_bytes, (unsigned long long)(lvol * 576));
  if (getVerbosity() >= QUDA_SUMMARIZE) {
    printfQuda("Volume:   \t%ix%ix%ix%i\n", ln[0], ln[1], ln[2], ln[3]);
    printfQuda("Subvolume:\t%ix%ix%ix%i\n", X[0], X[1], X[2], X[3]);
  }
  const long nvh = (long)X[0] * X[1] * X[2] * X[3] / 2;
  double *res[4];
  for (int mu = 0; mu < 4; mu++) { res[mu] = (double *)gauge[mu]; if (!res[mu]) errorQuda("gauge[%d] is NULL", mu); }
  const int s0 = cg.coords[0] * X[0], s1 = cg.coords[1] * X[1], s2 = cg.coords[2] * X[2], s3 = cg.coords[3] * X[3];
  std::vector<double> row((size_t)X[0] * 72);
  const bool swap = !hostIsBigEndian();
  for (int t = 0; t < X[3]; t++)
    for (int z = 0; z < X[2]; z++)
      for (int y = 0; y < X[1]; y++) {
        const uint64_t first = ((((uint64_t)(s3 + t) * ln[2] + (s2 + z)) * ln[1] + (s1 + y)) * ln[0] + s0) * 72;
        if (fseek(f, payload + (long)(first * sizeof(double)), SEEK_SET) != 0) errorQuda("%s: seek failed", fname);
        if (fread(row.data(), sizeof(double), row.size(), f) != row.size()) errorQuda("Error, could not read proper amount of data");
        if (swap) swap8(row.data(), row.size());
        for (int x = 0; x < X[0]; x++) {
          const int oddBit = (s0 + x + s1 + y + s2 + z + s3 + t) & 1;   // parity of the GLOBAL coordinates, as the reference (:347)
          const long iy = ((long)x + (long)y * X[0] + (long)z * X[1] * X[0] + (long)t * X[0] * X[1] * X[2]) / 2;
          for (int mu = 0; mu < 4; mu++) memcpy(res[mu] + ((long)oddBit * nvh + iy) * 18, &row[(size_t)x * 72 + mu * 18], 18 * sizeof(double));
        }
      }
  fclose(f);
}

// single-rank writer of the same container (ildg-format + optional xlf-info + ildg-binary-data), so files can be produced
// where no other ILDG tool exists; gauge in the even-odd QDP order of loadGaugeQuda, fp64
void qudaAmdWriteLimeGauge(void **gauge, const char *fname, const QudaGaugeParam *param, const char *xlf_info) {
  if (commGrid().size != 1) errorQuda("qudaAmdWriteLimeGauge: single rank only");
  FILE *f = fopen(fname, "wb");
  if (!f) errorQuda("could not open %s for writing", fname);
  const int *X = param->X;
  auto record = [&](const char *type, const void *data, uint64_t bytes, bool mb, bool me) {
    unsigned char h[144];
    memset(h, 0, sizeof(h));
    put_be(h, 0x456789abu, 4); put_be(h + 4, 1, 2); put_be(h + 6, (mb ? 0x8000u : 0u) | (me ? 0x4000u : 0u), 2); put_be(h + 8, bytes, 8);
    strncpy((char *)h + 16, type, 127);
    fwrite(h, 1, 144, f);
    if (data) {
      fwrite(data, 1, bytes, f);
      const unsigned char zero[8] = {0};
      if (bytes % 8) fwrite(zero, 1, 8 - bytes % 8, f);
    }
  };
  char xml[512];
  snprintf(xml, sizeof(xml),
           "<?xml version=\"1.0\" encoding=\"UTF-8\"?><ildgFormat xmlns=\"http://www.lqcd.org/ildg\"><version>1.0</version><field>su3gauge</field>"
           "<precision>64</precision><lx>%d</lx><ly>%d</ly><lz>%d</lz><lt>%d</lt></ildgFormat>", X[0], X[1], X[2], X[3]);
  if (xlf_info) record("xlf-info", xlf_info, strlen(xlf_info), true, true);
  record("ildg-format", xml, strlen(xml), true, false);
  const uint64_t lvol = (uint64_t)X[0] * X[1] * X[2] * X[3];
  record("ildg-binary-data", nullptr, lvol * 576, false, true);
  const long nvh = (long)(lvol / 2);
  std::vector<double> row((size_t)X[0] * 72);
  const bool swap = !hostIsBigEndian();
  for (int t = 0; t < X[3]; t++)
    for (int z = 0; z < X[2]; z++)
      for (int y = 0; y < X[1]; y++) {
        for (int x = 0; x < X[0]; x++) {
          const int oddBit = (x + y + z + t) & 1;
          const long iy = ((long)x + (long)y * X[0] + (long)z * X[1] * X[0] + (long)t * X[0] * X[1] * X[2]) / 2;
          for (int mu = 0; mu < 4; mu++) memcpy(&row[(size_t)x * 72 + mu * 18], (const double *)gauge[mu] + ((long)oddBit * nvh + iy) * 18, 18 * sizeof(double));
        }
        if (swap) swap8(row.data(), row.size());
        fwrite(row.data(), sizeof(double), row.size(), f);
      }
  fclose(f);
}

}
