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

}  // namespace quda

using namespace quda;

extern "C" {

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
