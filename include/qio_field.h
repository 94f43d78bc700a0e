/*
 * qio_field.h — file I/O helpers under the names the reference's tests use (include/qio_field.h: read_gauge_field,
 * read_spinor_field, write_spinor_field; QIO-backed there, lib/qio_field.cpp).  QIO is not a dependency of this library:
 * gauge configurations are read from ILDG / LIME containers by the library's own reader (qudaAmdReadLimeGauge,
 * csrc/lime_io.cpp), and vector files are the SciDAC records QIO writes, restated from the published format in the same file
 * (qudaAmdReadSpinorFields / qudaAmdWriteSpinorFields) — what the multigrid object uses for vec_infile / vec_outfile too.
 */
#ifndef _GAUGE_QIO_H
#define _GAUGE_QIO_H

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <quda.h>
#include <quda_amd_ext.h>

/* gauge[4]: even-odd QDP-ordered links of the LOCAL lattice X[4] (allocated by the caller), precision prec */
inline void read_gauge_field(const char *filename, void *gauge[], QudaPrecision prec, const int *X, int argc, char *argv[]) {
  (void)argc; (void)argv;
  QudaGaugeParam p = newQudaGaugeParam();
  size_t V = 1;
  for (int d = 0; d < 4; d++) { p.X[d] = X[d]; V *= (size_t)X[d]; }
  int grid[4] = {1, 1, 1, 1};
  if (prec == QUDA_DOUBLE_PRECISION) {
    qudaAmdReadLimeGauge(gauge, filename, &p, NULL, grid);
  } else {
    double *tmp[4];
    for (int mu = 0; mu < 4; mu++) tmp[mu] = (double *)malloc(V * 18 * sizeof(double));
    qudaAmdReadLimeGauge((void **)tmp, filename, &p, NULL, grid);
    for (int mu = 0; mu < 4; mu++) {
      for (size_t i = 0; i < V * 18; i++) ((float *)gauge[mu])[i] = (float)tmp[mu][i];
      free(tmp[mu]);
    }
  }
  for (int d = 0; d < 4; d++)
    if (p.X[d] != X[d]) { fprintf(stderr, "read_gauge_field: %s holds a %dx%dx%dx%d lattice\n", filename, p.X[0], p.X[1], p.X[2], p.X[3]); exit(1); }
}
/* V[Nvec]: host fields of the LOCAL lattice X[4] in even-odd site order, 2 nSpin nColor reals per site in `precision`; the file is the
 * SciDAC / QIO single-file container of the reference (lib/qio_field.cpp:198-328), written and read by the library's own code */
inline void read_spinor_field(const char *filename, void *V[], QudaPrecision precision, const int *X, int nColor, int nSpin, int Nvec, int argc, char *argv[]) {
  (void)argc; (void)argv;
  qudaAmdReadSpinorFields(filename, V, precision, X, nColor, nSpin, Nvec);
}
inline void write_spinor_field(const char *filename, void *V[], QudaPrecision precision, const int *X, int nColor, int nSpin, int Nvec, int argc, char *argv[]) {
  (void)argc; (void)argv;
  qudaAmdWriteSpinorFields(filename, V, precision, X, nColor, nSpin, Nvec);
}

#endif /* _GAUGE_QIO_H */
