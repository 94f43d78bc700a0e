// blas.h — the fused BLAS-1 / reduction set the GCR / MR / BiCGstab / MG path uses
// (reference include/blas_quda.h:33-144; CPU twins lib/blas_cpu.cpp:10-358 define the semantics).
#pragma once

#include <complex>
#include <vector>

#include "fields.h"

namespace quda {

typedef std::complex<double> Complex;
struct double3_t { double x, y, z; };

namespace blas {

extern unsigned long long flops;
extern unsigned long long bytes;

void init();
void end();
// when false, reductions stay rank-local (reference global_reduction switch, lib/face_buffer.cpp:409)
void setGlobalReduction(bool on);
bool globalReduction();

void zero(ColorSpinorField &a);
void copy(ColorSpinorField &dst, const ColorSpinorField &src);

double norm2(const ColorSpinorField &a);
double reDotProduct(const ColorSpinorField &x, const ColorSpinorField &y);
Complex cDotProduct(const ColorSpinorField &x, const ColorSpinorField &y);           // sum conj(x) y
double3_t cDotProductNormA(const ColorSpinorField &x, const ColorSpinorField &y);    // (re, im, |x|^2)
double3_t cDotProductNormB(const ColorSpinorField &x, const ColorSpinorField &y);    // (re, im, |y|^2)

void ax(const double &a, ColorSpinorField &x);                                        // x = a x
void axpy(const double &a, const ColorSpinorField &x, ColorSpinorField &y);           // y = a x + y
void xpy(const ColorSpinorField &x, ColorSpinorField &y);                             // y = x + y
void xpay(const ColorSpinorField &x, const double &a, ColorSpinorField &y);           // y = x + a y
void mxpy(const ColorSpinorField &x, ColorSpinorField &y);                            // y = y - x
void axpby(const double &a, const ColorSpinorField &x, const double &b, ColorSpinorField &y);  // y = a x + b y
double xmyNorm(const ColorSpinorField &x, ColorSpinorField &y);                       // y = x - y ; |y|^2
double axpyNorm(const double &a, const ColorSpinorField &x, ColorSpinorField &y);     // y = a x + y ; |y|^2

void caxpy(const Complex &a, const ColorSpinorField &x, ColorSpinorField &y);         // y = a x + y
void caxpby(const Complex &a, const ColorSpinorField &x, const Complex &b, ColorSpinorField &y);
void xmyz(const ColorSpinorField &x, const ColorSpinorField &y, ColorSpinorField &z);   // z = x - y (z write-only; y may be z)
void cxpaypbz(const ColorSpinorField &x, const Complex &a, const ColorSpinorField &y, const Complex &b, ColorSpinorField &z);  // z = x + a y + b z
double caxpyNorm(const Complex &a, const ColorSpinorField &x, ColorSpinorField &y);
void caxpyXmaz(const Complex &a, ColorSpinorField &x, ColorSpinorField &y, const ColorSpinorField &z);   // y += a x ; x -= a z
void caxpyXmazMR(const Complex &a, ColorSpinorField &x, ColorSpinorField &y, const ColorSpinorField &z);
double caxpyXmazNormX(const Complex &a, ColorSpinorField &x, ColorSpinorField &y, const ColorSpinorField &z);
void cabxpyAx(const double &a, const Complex &b, ColorSpinorField &x, ColorSpinorField &y);             // x = a x ; y += b x
double cabxpyAxNorm(const double &a, const Complex &b, ColorSpinorField &x, ColorSpinorField &y);
// device-side scalars (rank-local reductions only): (x, y) and |x|^2 stay in device memory, and the three MR updates take
// alpha = omega (x, y) / |x|^2 from there — no host round trip between the operator application and the update
bool deviceScalars();
void cDotProductNormADev(const ColorSpinorField &x, const ColorSpinorField &y);
void caxpyXmazDev(double omega, ColorSpinorField &x, ColorSpinorField &y, const ColorSpinorField &z);
void caxXmazDev(double omega, ColorSpinorField &x, ColorSpinorField &y, const ColorSpinorField &z);
void caxInitDev(double omega, const ColorSpinorField &x, ColorSpinorField &y, const ColorSpinorField &z, ColorSpinorField &w);
void caxXmaz(const Complex &a, ColorSpinorField &x, ColorSpinorField &y, const ColorSpinorField &z);   // y = a x ; x -= a z
void caxInit(const Complex &a, const ColorSpinorField &x, ColorSpinorField &y, const ColorSpinorField &z, ColorSpinorField &w);   // y = a x ; w = x - a z
Complex caxpyDotzy(const Complex &a, const ColorSpinorField &x, ColorSpinorField &y, const ColorSpinorField &z);  // y += a x ; (z,y)
void caxpbypzYmbw(const Complex &a, const ColorSpinorField &x, const Complex &b, ColorSpinorField &y, ColorSpinorField &z,
                  const ColorSpinorField &w);                                         // z += a x + b y ; y -= b w
double3_t HeavyQuarkResidualNorm(const ColorSpinorField &x, const ColorSpinorField &r);
// multi-field forms for the blocked orthogonalisation of GCR (reference lib/inv_gcr_quda.cpp:53-84, :103-121: N dots / N caxpys per pass),
// k <= 20 fields of fp64 / fp32 (multiSupported):
//   multiDot:            beta[i] = (f_i, y) for i < k, yr = (y, r), ynorm = |y|^2                       — one sweep
//   multiCaxpyResidual:  y <- scale (y + sum_i c_i f_i) ; r <- r - a y ; r2 = |r|^2, y2 = |y|^2        — one sweep
//   multiCaxpy:          y <- y + sum_i c_i f_i
bool multiSupported(const ColorSpinorField &x, int k);
void multiDot(Complex *beta, Complex &yr, double &ynorm, const std::vector<ColorSpinorField *> &f, int k, const ColorSpinorField &y, const ColorSpinorField &r);
void multiCaxpyResidual(double &r2, double &y2, const Complex *c, const std::vector<ColorSpinorField *> &f, int k, double scale, ColorSpinorField &y, const Complex &a, ColorSpinorField &r);
void multiCaxpy(const Complex *c, const std::vector<ColorSpinorField *> &f, int k, ColorSpinorField &y);

// vectorised forms used by GCR's orthogonalisation / solution update
void caxpy(const Complex *a, std::vector<ColorSpinorField *> &x, std::vector<ColorSpinorField *> &y);
void cDotProduct(Complex *result, std::vector<ColorSpinorField *> &a, std::vector<ColorSpinorField *> &b);

}  // namespace blas

// rank reductions (comm layer)
void comm_allreduce(double *data, int n);
bool commReductionsNeeded();   // more than one rank (or the RCCL self-test mode): sums must go through the all-reduce
void commAllreduceDevice(double *d_data, int n, hipStream_t s);
void comm_allreduce_max(double *data, int n);

}  // namespace quda
