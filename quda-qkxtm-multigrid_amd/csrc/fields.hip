// fields.hip — field allocation, host<->device reorder (+ precision / gamma-basis change), gauge and clover
// upload into the CDNA4 device layouts.  Reference behaviour restated: lib/color_spinor_field.cpp:129-216
// (geometry), lib/copy_color_spinor.cuh:49-91 (basis rotation), lib/cuda_color_spinor_field.cu:513-590
// (load/save), lib/clover_invert.cu:56-85 (twisted inverse).
#include "fields.h"
#include "p2p.h"

#include <cmath>
#include <cstring>
#include <vector>

#include "basis.h"
#include "device_io.h"
#include "halo.h"

namespace quda {

// ================================================================================================
// ColorSpinorField
// ================================================================================================
ColorSpinorParam::ColorSpinorParam(void *V, const QudaInvertParam &inv, const int *X, bool pc_solution) {
  location = QUDA_CPU_FIELD_LOCATION;
  nColor = 3; nSpin = 4; nDim = 4;
  for (int d = 0; d < 4; d++) x[d] = X[d];
  siteSubset = pc_solution ? QUDA_PARITY_SITE_SUBSET : QUDA_FULL_SITE_SUBSET;
  if (pc_solution) x[0] /= 2;
  precision = inv.cpu_prec;
  pad = 0;
  twistFlavor = inv.twist_flavor;
  siteOrder = QUDA_EVEN_ODD_SITE_ORDER;
  if (inv.dirac_order == QUDA_DIRAC_ORDER) fieldOrder = QUDA_SPACE_SPIN_COLOR_FIELD_ORDER;
  else if (inv.dirac_order == QUDA_QDP_DIRAC_ORDER) fieldOrder = QUDA_SPACE_COLOR_SPIN_FIELD_ORDER;
  else errorQuda("Dirac order %d not supported (QUDA_DIRAC_ORDER, QUDA_QDP_DIRAC_ORDER)", inv.dirac_order);
  gammaBasis = inv.gamma_basis;
  create = QUDA_REFERENCE_FIELD_CREATE;
  v = V;
}

static size_t alignUp(size_t n, size_t a) { return (n + a - 1) / a * a; }

// Plane padding (sites) of the device fields, QUDA_AMD_FIELD_PAD: the planes of a field are stride x 16 bytes apart, which is a multiple
// of 1 MiB on 32^4 and 48^3 x 96 — every plane of a site then falls into the same L2 set window (the reference pads for the same
// reason, "partition camping": sp_pad / ga_pad / cl_pad of QudaInvertParam / QudaGaugeParam).
int fieldPadSites() {
  static int v = [] { const char *e = getenv("QUDA_AMD_FIELD_PAD"); return e ? atoi(e) : 0; }();
  return v;
}
static int gaugePadSites() {
  static int v = [] { const char *e = getenv("QUDA_AMD_GAUGE_PAD"); return e ? atoi(e) : fieldPadSites(); }();
  return v;
}

ColorSpinorField::ColorSpinorField(const ColorSpinorParam &p)
    : location(p.location), nColor(p.nColor), nSpin(p.nSpin), nDim(p.nDim), pad(p.pad), precision(p.precision),
      siteSubset(p.siteSubset), siteOrder(p.siteOrder), fieldOrder(p.fieldOrder), gammaBasis(p.gammaBasis),
      twistFlavor(p.twistFlavor), v_(nullptr), norm_(nullptr), owns(false), even_(nullptr), odd_(nullptr) {
  volume = 1;
  for (int d = 0; d < 4; d++) { x[d] = p.x[d]; volume *= p.x[d]; }
  if (volume <= 0) errorQuda("empty field: x = %d %d %d %d", x[0], x[1], x[2], x[3]);
  const int nsub = siteSubset == QUDA_FULL_SITE_SUBSET ? 2 : 1;
  volumeCB = volume / nsub;
  if (location == QUDA_CPU_FIELD_LOCATION) pad = 0;
  else if (p.planePad && p.create != QUDA_REFERENCE_FIELD_CREATE && nSpin == 4) pad += fieldPadSites();
  stride = volumeCB + pad;
  if (location == QUDA_CUDA_FIELD_LOCATION) {
    fieldOrder = (precision == QUDA_DOUBLE_PRECISION || nSpin != 4) ? QUDA_FLOAT2_FIELD_ORDER : QUDA_FLOAT4_FIELD_ORDER;
    if (nSpin == 4) gammaBasis = QUDA_DEGRAND_ROSSI_GAMMA_BASIS;  // device-internal basis (see fields.h)
    if (nSpin != 4 && precision == QUDA_HALF_PRECISION) errorQuda("16-bit coarse fields are not supported");
  } else {
    if (precision == QUDA_HALF_PRECISION) errorQuda("16-bit host fields are not supported");
    if (fieldOrder == QUDA_INVALID_FIELD_ORDER) fieldOrder = QUDA_SPACE_SPIN_COLOR_FIELD_ORDER;
  }
  const size_t per_parity = (size_t)stride * nColor * nSpin * 2 * precision;
  // keep each parity half 1 KiB aligned like the reference (TEX_ALIGN_REQ, include/quda_internal.h:32-33)
  const size_t half = location == QUDA_CUDA_FIELD_LOCATION ? alignUp(per_parity, 1024) : per_parity;
  bytes = nsub * half;
  norm_bytes = precision == QUDA_HALF_PRECISION ? nsub * alignUp((size_t)stride * sizeof(float), 1024) : 0;

  if (p.create == QUDA_REFERENCE_FIELD_CREATE) {
    v_ = p.v; norm_ = p.norm;
    if (!v_) errorQuda("reference field without data pointer");
  } else {
    owns = true;
    if (location == QUDA_CUDA_FIELD_LOCATION) {
      v_ = poolDeviceMalloc(bytes);
      if (norm_bytes) norm_ = poolDeviceMalloc(norm_bytes);
    } else {
      v_ = malloc(bytes);
      if (!v_) errorQuda("host allocation of %zu bytes failed", bytes);
    }
    if (p.create == QUDA_ZERO_FIELD_CREATE || (pad > 0 && location == QUDA_CUDA_FIELD_LOCATION)) zero();   // the flat BLAS kernels run over the pad too
  }
}

ColorSpinorParam ColorSpinorField::param() const {
  ColorSpinorParam p;
  p.location = location; p.nColor = nColor; p.nSpin = nSpin; p.nDim = nDim;
  for (int d = 0; d < 4; d++) p.x[d] = x[d];
  p.precision = precision; p.pad = pad; p.planePad = false; p.twistFlavor = twistFlavor; p.siteSubset = siteSubset; p.siteOrder = siteOrder;
  p.fieldOrder = fieldOrder; p.gammaBasis = gammaBasis; p.create = QUDA_NULL_FIELD_CREATE;
  return p;
}

ColorSpinorField::ColorSpinorField(const ColorSpinorField &src) : ColorSpinorField([&] { ColorSpinorParam p = src.param(); return p; }()) {
  copyColorSpinor(*this, src);
}

ColorSpinorField::~ColorSpinorField() {
  delete even_;
  delete odd_;
  if (owns) {
    if (location == QUDA_CUDA_FIELD_LOCATION) {
      poolDeviceFree(v_, bytes);
      poolDeviceFree(norm_, norm_bytes);
    } else {
      free(v_);
    }
  }
}

void ColorSpinorField::zero() {
  if (location == QUDA_CUDA_FIELD_LOCATION) {
    HIP_CHECK(hipMemsetAsync(v_, 0, bytes, computeStream()));
    if (norm_bytes) HIP_CHECK(hipMemsetAsync(norm_, 0, norm_bytes, computeStream()));
  } else {
    memset(v_, 0, bytes);
  }
}

void ColorSpinorField::createViews() {
  if (siteSubset != QUDA_FULL_SITE_SUBSET) errorQuda("Even()/Odd() need a full field");
  ColorSpinorParam p = param();
  p.siteSubset = QUDA_PARITY_SITE_SUBSET;
  p.x[0] = x[0] / 2;
  p.create = QUDA_REFERENCE_FIELD_CREATE;
  p.v = v_; p.norm = norm_;
  even_ = new ColorSpinorField(p);
  p.v = (char *)v_ + bytes / 2;  // reference lib/cpu_color_spinor_field.cpp:165
  p.norm = norm_ ? (char *)norm_ + norm_bytes / 2 : nullptr;
  odd_ = new ColorSpinorField(p);
}
ColorSpinorField &ColorSpinorField::Even() { if (!even_) createViews(); return *even_; }
ColorSpinorField &ColorSpinorField::Odd() { if (!odd_) createViews(); return *odd_; }

ColorSpinorField &ColorSpinorField::operator=(const ColorSpinorField &src) {
  if (&src != this) copyColorSpinor(*this, src);
  return *this;
}

static ColorSpinorParam deviceParamFrom(const ColorSpinorParam &p) { ColorSpinorParam q = p; q.location = QUDA_CUDA_FIELD_LOCATION; return q; }
static ColorSpinorParam hostParamFrom(const ColorSpinorParam &p) { ColorSpinorParam q = p; q.location = QUDA_CPU_FIELD_LOCATION; return q; }
cudaColorSpinorField::cudaColorSpinorField(const ColorSpinorParam &p) : ColorSpinorField(deviceParamFrom(p)) {}
cudaColorSpinorField::cudaColorSpinorField(const ColorSpinorField &src, const ColorSpinorParam &p) : ColorSpinorField(deviceParamFrom(p)) {
  twistFlavor = src.twistFlavor;
  copyColorSpinor(*this, src);
}
cpuColorSpinorField::cpuColorSpinorField(const ColorSpinorParam &p) : ColorSpinorField(hostParamFrom(p)) {}

// ---- reorder kernels -----------------------------------------------------------------------------
// host site-major record of NR reals: index (site*NR + (s*Nc + c)*2 + z)   [SPACE_SPIN_COLOR]
//                                       or   (site*NR + (c*Ns + s)*2 + z)   [SPACE_COLOR_SPIN]
template <typename TDev, typename THost>
__global__ void spinor_h2d_kernel(void *dev, float *norm, int stride, const THost *host, int Vh, int color_spin, int change) {
  using real = typename Store<TDev>::real;
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= Vh) return;
  real r[24], q[24];
  const THost *h = host + (size_t)x * 24;
#pragma unroll
  for (int s = 0; s < 4; s++)
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const int hi = color_spin ? (c * 4 + s) * 2 : (s * 3 + c) * 2;
      r[(s * 3 + c) * 2] = (real)h[hi];
      r[(s * 3 + c) * 2 + 1] = (real)h[hi + 1];
    }
  if (change) { rotate_basis(q, r, change); Planar<TDev, 24>::store(q, dev, stride, x, norm, x); }
  else Planar<TDev, 24>::store(r, dev, stride, x, norm, x);
}

template <typename TDev, typename THost>
__global__ void spinor_d2h_kernel(THost *host, const void *dev, const float *norm, int stride, int Vh, int color_spin, int change) {
  using real = typename Store<TDev>::real;
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= Vh) return;
  real r[24], q[24];
  Planar<TDev, 24>::load(r, dev, stride, x, norm, x);
  if (change) rotate_basis(q, r, change);
  THost *h = host + (size_t)x * 24;
#pragma unroll
  for (int s = 0; s < 4; s++)
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const int hi = color_spin ? (c * 4 + s) * 2 : (s * 3 + c) * 2;
      h[hi] = (THost)(change ? q[(s * 3 + c) * 2] : r[(s * 3 + c) * 2]);
      h[hi + 1] = (THost)(change ? q[(s * 3 + c) * 2 + 1] : r[(s * 3 + c) * 2 + 1]);
    }
}

template <typename TOut, typename TIn>
__global__ void spinor_d2d_kernel(void *out, float *onorm, int ostride, const void *in, const float *inorm, int istride, int Vh) {
  using rin = typename Store<TIn>::real;
  using rout = typename Store<TOut>::real;
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= Vh) return;
  rin r[24];
  rout q[24];
  Planar<TIn, 24>::load(r, in, istride, x, inorm, x);
#pragma unroll
  for (int k = 0; k < 24; k++) q[k] = (rout)r[k];
  Planar<TOut, 24>::store(q, out, ostride, x, onorm, x);
}

// generic complex-plane fields (coarse grids: nSpin*nColor complex FLOAT2 planes), host site-major
template <typename TDev, typename THost>
__global__ void generic_h2d_kernel(TDev *dev, int stride, const THost *host, int Vh, int ncomplex) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= Vh) return;
  for (int k = 0; k < ncomplex; k++) {
    dev[((size_t)k * stride + x) * 2] = (TDev)host[((size_t)x * ncomplex + k) * 2];
    dev[((size_t)k * stride + x) * 2 + 1] = (TDev)host[((size_t)x * ncomplex + k) * 2 + 1];
  }
}
template <typename TDev, typename THost>
__global__ void generic_d2h_kernel(THost *host, const TDev *dev, int stride, int Vh, int ncomplex) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= Vh) return;
  for (int k = 0; k < ncomplex; k++) {
    host[((size_t)x * ncomplex + k) * 2] = (THost)dev[((size_t)k * stride + x) * 2];
    host[((size_t)x * ncomplex + k) * 2 + 1] = (THost)dev[((size_t)k * stride + x) * 2 + 1];
  }
}
template <typename TOut, typename TIn>
__global__ void generic_d2d_kernel(TOut *out, int ostride, const TIn *in, int istride, int Vh, int ncomplex) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= Vh) return;
  for (int k = 0; k < ncomplex; k++) {
    out[((size_t)k * ostride + x) * 2] = (TOut)in[((size_t)k * istride + x) * 2];
    out[((size_t)k * ostride + x) * 2 + 1] = (TOut)in[((size_t)k * istride + x) * 2 + 1];
  }
}

// staging buffer for host<->device transfers (grows on demand, lives until endQuda)
static void *g_stage = nullptr;
static size_t g_stage_bytes = 0;
void *stagingBuffer(size_t bytes) {
  if (bytes > g_stage_bytes) {
    if (g_stage) HIP_CHECK(hipFree(g_stage));
    HIP_CHECK(qaMalloc(&g_stage, bytes));
    g_stage_bytes = bytes;
  }
  return g_stage;
}
void freeStagingBuffer() {
  if (g_stage) (void)hipFree(g_stage);
  g_stage = nullptr;
  g_stage_bytes = 0;
}

static int basisChange(QudaGammaBasis from, QudaGammaBasis to) {
  auto norm = [](QudaGammaBasis b) { return b == QUDA_CHIRAL_GAMMA_BASIS ? QUDA_DEGRAND_ROSSI_GAMMA_BASIS : b; };
  from = norm(from); to = norm(to);
  if (from == to) return BASIS_NONE;
  if (from == QUDA_UKQCD_GAMMA_BASIS && to == QUDA_DEGRAND_ROSSI_GAMMA_BASIS) return BASIS_UKQCD_TO_DR;
  if (from == QUDA_DEGRAND_ROSSI_GAMMA_BASIS && to == QUDA_UKQCD_GAMMA_BASIS) return BASIS_DR_TO_UKQCD;
  errorQuda("unsupported basis change %d -> %d", from, to);
  return 0;
}

template <typename TDev, typename THost> static void h2dParity(ColorSpinorField &dst, const ColorSpinorField &src) {
  const int Vh = dst.VolumeCB(), bs = 256, nb = (Vh + bs - 1) / bs;
  const size_t hbytes = (size_t)Vh * src.nSpin * src.nColor * 2 * sizeof(THost);
  void *stage = stagingBuffer(hbytes);
  HIP_CHECK(hipMemcpyAsync(stage, src.V(), hbytes, hipMemcpyHostToDevice, computeStream()));
  if (dst.nSpin == 4 && dst.nColor == 3) {
    const int cs = src.fieldOrder == QUDA_SPACE_COLOR_SPIN_FIELD_ORDER;
    hipLaunchKernelGGL((spinor_h2d_kernel<TDev, THost>), dim3(nb), dim3(bs), 0, computeStream(), dst.V(), (float *)dst.Norm(), dst.Stride(),
                       (const THost *)stage, Vh, cs, basisChange(src.gammaBasis, dst.gammaBasis));
  } else {
    if (sizeof(TDev) == 2) errorQuda("16-bit coarse fields unsupported");
    using D = typename Store<TDev>::real;
    hipLaunchKernelGGL((generic_h2d_kernel<D, THost>), dim3(nb), dim3(bs), 0, computeStream(), (D *)dst.V(), dst.Stride(), (const THost *)stage, Vh,
                       dst.nSpin * dst.nColor);
  }
  HIP_CHECK(hipGetLastError());
  HIP_CHECK(hipStreamSynchronize(computeStream()));  // staging buffer is reused by the next transfer
}

template <typename TDev, typename THost> static void d2hParity(ColorSpinorField &dst, const ColorSpinorField &src) {
  const int Vh = src.VolumeCB(), bs = 256, nb = (Vh + bs - 1) / bs;
  const size_t hbytes = (size_t)Vh * src.nSpin * src.nColor * 2 * sizeof(THost);
  void *stage = stagingBuffer(hbytes);
  if (src.nSpin == 4 && src.nColor == 3) {
    const int cs = dst.fieldOrder == QUDA_SPACE_COLOR_SPIN_FIELD_ORDER;
    hipLaunchKernelGGL((spinor_d2h_kernel<TDev, THost>), dim3(nb), dim3(bs), 0, computeStream(), (THost *)stage, src.V(), (const float *)src.Norm(),
                       src.Stride(), Vh, cs, basisChange(src.gammaBasis, dst.gammaBasis));
  } else {
    using D = typename Store<TDev>::real;
    hipLaunchKernelGGL((generic_d2h_kernel<D, THost>), dim3(nb), dim3(bs), 0, computeStream(), (THost *)stage, (const D *)src.V(), src.Stride(), Vh,
                       src.nSpin * src.nColor);
  }
  HIP_CHECK(hipGetLastError());
  HIP_CHECK(hipMemcpyAsync(dst.V(), stage, hbytes, hipMemcpyDeviceToHost, computeStream()));
  HIP_CHECK(hipStreamSynchronize(computeStream()));
  p2pCheck("download of a result field");
}

template <typename TOut, typename TIn> static void d2dParity(ColorSpinorField &dst, const ColorSpinorField &src) {
  const int Vh = src.VolumeCB(), bs = 256, nb = (Vh + bs - 1) / bs;
  acct(src.nSpin == 4 && src.nColor == 3 ? "spinor_d2d_kernel" : "generic_d2d_kernel", (double)Vh * src.nSpin * src.nColor * 2 * (sizeof(TOut) + sizeof(TIn)), src.nSpin == 4 ? "level 0" : "coarse");
  if (src.nSpin == 4 && src.nColor == 3) {
    hipLaunchKernelGGL((spinor_d2d_kernel<TOut, TIn>), dim3(nb), dim3(bs), 0, computeStream(), dst.V(), (float *)dst.Norm(), dst.Stride(), src.V(),
                       (const float *)src.Norm(), src.Stride(), Vh);
  } else {
    using O = typename Store<TOut>::real;
    using I = typename Store<TIn>::real;
    hipLaunchKernelGGL((generic_d2d_kernel<O, I>), dim3(nb), dim3(bs), 0, computeStream(), (O *)dst.V(), dst.Stride(), (const I *)src.V(), src.Stride(),
                       Vh, src.nSpin * src.nColor);
  }
  HIP_CHECK(hipGetLastError());
}

#define QA_DISPATCH_DEV(prec, CALL)                      \
  switch (prec) {                                        \
    case QUDA_DOUBLE_PRECISION: { using TD = double; CALL; } break; \
    case QUDA_SINGLE_PRECISION: { using TD = float; CALL; } break;  \
    case QUDA_HALF_PRECISION: { using TD = short; CALL; } break;    \
    default: errorQuda("bad precision %d", prec);        \
  }

static void copyParity(ColorSpinorField &dst, const ColorSpinorField &src) {
  const bool dd = dst.Location() == QUDA_CUDA_FIELD_LOCATION, sd = src.Location() == QUDA_CUDA_FIELD_LOCATION;
  if (dd && !sd) {
    if (src.Precision() == QUDA_DOUBLE_PRECISION) { QA_DISPATCH_DEV(dst.Precision(), (h2dParity<TD, double>(dst, src))); }
    else { QA_DISPATCH_DEV(dst.Precision(), (h2dParity<TD, float>(dst, src))); }
  } else if (!dd && sd) {
    if (dst.Precision() == QUDA_DOUBLE_PRECISION) { QA_DISPATCH_DEV(src.Precision(), (d2hParity<TD, double>(dst, src))); }
    else { QA_DISPATCH_DEV(src.Precision(), (d2hParity<TD, float>(dst, src))); }
  } else if (dd && sd) {
    if (dst.Precision() == src.Precision() && dst.Stride() == src.Stride()) {
      const size_t n = (size_t)src.Stride() * src.nSpin * src.nColor * 2 * src.Precision();
      HIP_CHECK(hipMemcpyAsync(dst.V(), src.V(), n, hipMemcpyDeviceToDevice, computeStream()));
      if (src.Precision() == QUDA_HALF_PRECISION)
        HIP_CHECK(hipMemcpyAsync(dst.Norm(), src.Norm(), (size_t)src.Stride() * sizeof(float), hipMemcpyDeviceToDevice, computeStream()));
    } else {
      switch (src.Precision()) {
        case QUDA_DOUBLE_PRECISION: QA_DISPATCH_DEV(dst.Precision(), (d2dParity<TD, double>(dst, src))); break;
        case QUDA_SINGLE_PRECISION: QA_DISPATCH_DEV(dst.Precision(), (d2dParity<TD, float>(dst, src))); break;
        case QUDA_HALF_PRECISION: QA_DISPATCH_DEV(dst.Precision(), (d2dParity<TD, short>(dst, src))); break;
        default: errorQuda("bad precision");
      }
    }
  } else {
    if (dst.Precision() != src.Precision() || dst.fieldOrder != src.fieldOrder || dst.gammaBasis != src.gammaBasis)
      errorQuda("host-to-host copies with layout change are not supported");
    memcpy(dst.V(), src.V(), (size_t)src.VolumeCB() * src.nSpin * src.nColor * 2 * src.Precision());
  }
}

void copyColorSpinor(ColorSpinorField &dst, const ColorSpinorField &src) {
  if (dst.nSpin != src.nSpin || dst.nColor != src.nColor) errorQuda("spin/colour mismatch %d,%d vs %d,%d", dst.nSpin, dst.nColor, src.nSpin, src.nColor);
  if (dst.SiteSubset() != src.SiteSubset()) errorQuda("site subset mismatch");
  if (dst.VolumeCB() != src.VolumeCB()) errorQuda("volume mismatch %d vs %d", dst.VolumeCB(), src.VolumeCB());
  if (src.SiteSubset() == QUDA_FULL_SITE_SUBSET) {
    copyParity(dst.Even(), src.Even());
    copyParity(dst.Odd(), src.Odd());
  } else {
    copyParity(dst, src);
  }
  dst.twistFlavor = dst.twistFlavor == QUDA_TWIST_NO || dst.twistFlavor == QUDA_TWIST_INVALID ? src.twistFlavor : dst.twistFlavor;
}

// ================================================================================================
// GaugeField
// ================================================================================================
GaugeField::GaugeField(const LatticeGeom &g, QudaPrecision prec, QudaReconstructType recon, QudaTboundary tb, double aniso)
    : geom(g), precision(prec), reconstruct(recon), t_boundary(tb), anisotropy(aniso), stride(g.Vh + gaugePadSites()), data(nullptr), tbc_folded(true) {
  if (recon != QUDA_RECONSTRUCT_NO && recon != QUDA_RECONSTRUCT_12 && recon != QUDA_RECONSTRUCT_8) errorQuda("reconstruct %d not supported (18, 12, 8)", recon);
  if (prec == QUDA_HALF_PRECISION && aniso != 1.0) errorQuda("16-bit links need anisotropy 1 (fixed-point range)");
  link_bytes = alignUp((size_t)stride * (int)recon * (int)prec, 1024);
  bytes = 16 * link_bytes;
  HIP_CHECK(qaMalloc(&data, bytes));
}
GaugeField::~GaugeField() { if (data) (void)hipFree(data); }

// boundary slice x_d = L_d - 1 of U_d (both parities) -> contiguous [parity][face index][18] block for the +d neighbour
template <typename THost>
__global__ void gauge_face_pack_kernel(THost *out, const THost *h, LatticeGeom g, int d) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  const int nf = g.faceCB[d];
  if (tid >= 2 * nf) return;
  const int q = tid >= nf, f = tid - q * nf;
  int c[4], L[3], o[3], n = 0;
  for (int k = 0; k < 4; k++) if (k != d) { L[n] = g.X[k]; o[n] = k; n++; }
  int l = 2 * f;
  const int c0 = l % L[0]; l /= L[0];
  const int c1 = l % L[1]; const int c2 = l / L[1];
  c[d] = g.X[d] - 1;
  c[o[0]] = c0; c[o[1]] = c1; c[o[2]] = c2;
  c[o[0]] += (q + c[0] + c[1] + c[2] + c[3]) & 1;
  const int idx = (((c[3] * g.X[2] + c[2]) * g.X[1] + c[1]) * g.X[0] + c[0]) >> 1;
  const THost *src = h + ((size_t)q * g.Vh + idx) * 18;
  THost *dst = out + (size_t)tid * 18;
  for (int k = 0; k < 18; k++) dst[k] = src[k];
}

// one thread per (parity, site): builds the 8 matrices the stencil needs at that site.  ghost[d] != nullptr: the
// backward link of a site on the x_d = 0 face lives on the -d neighbour rank and is taken from its packed last slice.
template <typename THost> struct GhostLinks { const THost *g[4]; };

// store one link in the device's reconstruct type: 18 / 12 the leading reals as they are, 8 the packed form of device_io.h su3_pack8
template <typename TDev, int R, typename real> __device__ __forceinline__ void store_link(const real *U, void *blk, int stride, int idx) {
  if constexpr (R == 8) {
    real o[8];
    su3_pack8(o, U, (real)(1.0 / PhaseUnit<TDev>::value));
    Planar<TDev, 8>::store(o, blk, stride, idx, nullptr, 0);
  } else {
    Planar<TDev, R>::store(U, blk, stride, idx, nullptr, 0);
  }
}

template <typename TDev, int R, typename THost>
__global__ void gauge_load_kernel(char *data, size_t link_bytes, int stride, const THost *h0, const THost *h1, const THost *h2, const THost *h3,
                                  LatticeGeom g, GhostLinks<THost> ghost) {
  using real = typename Store<TDev>::real;
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= 2 * g.Vh) return;
  const int parity = gid >= g.Vh, idx = gid - parity * g.Vh;
  const uint32_t za = g.dXh.div((uint32_t)idx);
  const int xh = idx - (int)za * g.Xh;
  const uint32_t zb = g.dY.div(za);
  const int y = (int)za - (int)zb * g.X[1];
  const int t = (int)g.dZ.div(zb);
  const int z = (int)zb - t * g.X[2];
  const int xodd = (y + z + t + parity) & 1;
  const int xf = 2 * xh + xodd;
  const int Xh = g.Xh, sy = Xh, sz = Xh * g.X[1], st = Xh * g.X[1] * g.X[2];
  int nb[4], face[4];
  nb[0] = xodd ? idx : (xh == 0 ? idx + (Xh - 1) : idx - 1);
  nb[1] = y == 0 ? idx + (g.X[1] - 1) * sy : idx - sy;
  nb[2] = z == 0 ? idx + (g.X[2] - 1) * sz : idx - sz;
  nb[3] = t == 0 ? idx + (g.X[3] - 1) * st : idx - st;
  face[0] = ((t * g.X[2] + z) * g.X[1] + y) >> 1;
  face[1] = ((t * g.X[2] + z) * g.X[0] + xf) >> 1;
  face[2] = ((t * g.X[1] + y) * g.X[0] + xf) >> 1;
  face[3] = ((z * g.X[1] + y) * g.X[0] + xf) >> 1;
  const int coord[4] = {xf, y, z, t};
  const THost *h[4] = {h0, h1, h2, h3};
  char *base = data + (size_t)parity * 8 * link_bytes;
#pragma unroll
  for (int mu = 0; mu < 4; mu++) {
    real U[18];
    const THost *f = h[mu] + ((size_t)parity * g.Vh + idx) * 18;
#pragma unroll
    for (int k = 0; k < 18; k++) U[k] = (real)f[k];
    store_link<TDev, R>(U, base + (size_t)(2 * mu) * link_bytes, stride, idx);
    const THost *bk = (ghost.g[mu] && coord[mu] == 0) ? ghost.g[mu] + ((size_t)(1 - parity) * g.faceCB[mu] + face[mu]) * 18
                                                       : h[mu] + ((size_t)(1 - parity) * g.Vh + nb[mu]) * 18;
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int c = 0; c < 3; c++) {  // dagger
        U[r * 6 + c * 2] = (real)bk[c * 6 + r * 2];
        U[r * 6 + c * 2 + 1] = -(real)bk[c * 6 + r * 2 + 1];
      }
    store_link<TDev, R>(U, base + (size_t)(2 * mu + 1) * link_bytes, stride, idx);
  }
}

template <typename TDev, int R, typename THost> static void gaugeLoad(GaugeField &U, void *const h_gauge[4]) {
  const LatticeGeom &g = U.geom;
  const size_t n = (size_t)g.V * 18 * sizeof(THost);
  size_t ghost_bytes = 0;
  // links of the -d neighbour are needed wherever dimension d is split over ranks; QUDA_AMD_FORCE_GAUGE_HALO=1 also sends a
  // self-partitioned dimension (qudaAmdSetPartitionMask) through the same pack / exchange / ghost-index path, for testing
  static const bool force = getenv("QUDA_AMD_FORCE_GAUGE_HALO") != nullptr;
  auto split = [&](int d) { return commGrid().dims[d] > 1 || (force && commGrid().forced[d]); };
  for (int d = 0; d < 4; d++) if (split(d)) ghost_bytes += 2 * (size_t)2 * g.faceCB[d] * 18 * sizeof(THost);
  char *stage = (char *)stagingBuffer(4 * n + ghost_bytes);
  hipStream_t s = computeStream();
  for (int d = 0; d < 4; d++) HIP_CHECK(hipMemcpyAsync(stage + d * n, h_gauge[d], n, hipMemcpyHostToDevice, s));
  GhostLinks<THost> ghost;
  std::vector<HaloMsg> msgs;
  char *p = stage + 4 * n;
  for (int d = 0; d < 4; d++) {
    ghost.g[d] = nullptr;
    if (!split(d)) continue;
    const size_t fb = (size_t)2 * g.faceCB[d] * 18 * sizeof(THost);
    THost *sendb = (THost *)p; p += fb;
    THost *recvb = (THost *)p; p += fb;
    const int nt = 2 * g.faceCB[d];
    hipLaunchKernelGGL((gauge_face_pack_kernel<THost>), dim3((nt + 255) / 256), dim3(256), 0, s, sendb, (const THost *)(stage + d * n), g, d);
    HIP_CHECK(hipGetLastError());
    msgs.push_back({d, +1, sendb, recvb, fb});  // my last slice goes forward; I receive my -d neighbour's last slice
    ghost.g[d] = recvb;
  }
  if (!msgs.empty()) commExchange(msgs, s);
  const int bs = 256, nb = (2 * g.Vh + bs - 1) / bs;
  hipLaunchKernelGGL((gauge_load_kernel<TDev, R, THost>), dim3(nb), dim3(bs), 0, s, (char *)U.data, U.link_bytes, U.stride, (const THost *)stage,
                     (const THost *)(stage + n), (const THost *)(stage + 2 * n), (const THost *)(stage + 3 * n), g, ghost);
  HIP_CHECK(hipGetLastError());
  HIP_CHECK(hipStreamSynchronize(s));
}

void GaugeField::loadQDP(void *const h_gauge[4], QudaPrecision cpu_prec) {
#define QA_GL(TD, RR)                                                     \
  if (cpu_prec == QUDA_DOUBLE_PRECISION) gaugeLoad<TD, RR, double>(*this, h_gauge); \
  else gaugeLoad<TD, RR, float>(*this, h_gauge);
  if (reconstruct == QUDA_RECONSTRUCT_NO) { QA_DISPATCH_DEV(precision, QA_GL(TD, 18)); }
  else if (reconstruct == QUDA_RECONSTRUCT_8) { QA_DISPATCH_DEV(precision, QA_GL(TD, 8)); }
  else { QA_DISPATCH_DEV(precision, QA_GL(TD, 12)); }
#undef QA_GL
}

// device-to-device copy of all 16 link blocks with precision change (same reconstruct): the 16-bit links of the multigrid's
// half-precision smoother are made from the resident fp32 field, the host copy is not kept
template <typename TOut, typename TIn, int R>
__global__ void gauge_convert_kernel(char *out, size_t out_link_bytes, const char *in, size_t in_link_bytes, int stride, int Vh) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= 16 * Vh) return;
  const int blk = gid / Vh, idx = gid - blk * Vh;
  typename Store<TIn>::real u[R];
  Planar<TIn, R>::load(u, in + (size_t)blk * in_link_bytes, stride, idx, nullptr, 0);
  typename Store<TOut>::real v[R];
#pragma unroll
  for (int k = 0; k < R; k++) v[k] = (typename Store<TOut>::real)u[k];
  if (R == 8) {   // the two phases are stored in the unit of their precision (16-bit: phase / pi)
    const typename Store<TOut>::real f = (typename Store<TOut>::real)(PhaseUnit<TIn>::value / PhaseUnit<TOut>::value);
    v[0] *= f; v[1] *= f;
  }
  Planar<TOut, R>::store(v, out + (size_t)blk * out_link_bytes, stride, idx, nullptr, 0);
}
void GaugeField::copyFrom(const GaugeField &src) {
  if (!(src.geom == geom) || src.reconstruct != reconstruct) errorQuda("gauge copy: geometry / reconstruct mismatch");
  if (src.precision != QUDA_SINGLE_PRECISION || precision != QUDA_HALF_PRECISION) errorQuda("gauge copy: fp32 -> 16-bit only");
  const int bs = 256, nb = (16 * geom.Vh + bs - 1) / bs;
  if (reconstruct == QUDA_RECONSTRUCT_12)
    hipLaunchKernelGGL((gauge_convert_kernel<short, float, 12>), dim3(nb), dim3(bs), 0, computeStream(), (char *)data, link_bytes, (const char *)src.data, src.link_bytes, stride, geom.Vh);
  else if (reconstruct == QUDA_RECONSTRUCT_8)
    hipLaunchKernelGGL((gauge_convert_kernel<short, float, 8>), dim3(nb), dim3(bs), 0, computeStream(), (char *)data, link_bytes, (const char *)src.data, src.link_bytes, stride, geom.Vh);
  else
    hipLaunchKernelGGL((gauge_convert_kernel<short, float, 18>), dim3(nb), dim3(bs), 0, computeStream(), (char *)data, link_bytes, (const char *)src.data, src.link_bytes, stride, geom.Vh);
  HIP_CHECK(hipGetLastError());
}

// ================================================================================================
// CloverField
// ================================================================================================
CloverField::CloverField(const LatticeGeom &g, QudaPrecision prec)
    : geom(g), precision(prec), stride(g.Vh + gaugePadSites()), clover(nullptr), cloverInv(nullptr), norm(nullptr), invNorm(nullptr), twisted(false), mu2(0) {
  parity_bytes = alignUp((size_t)stride * 72 * (int)prec, 1024);
  bytes = 2 * parity_bytes;
  HIP_CHECK(qaMalloc(&clover, bytes));
  HIP_CHECK(qaMalloc(&cloverInv, bytes));
  parity_norm_bytes = 0;
  if (prec == QUDA_HALF_PRECISION) {
    parity_norm_bytes = (size_t)2 * stride * sizeof(float);
    HIP_CHECK(qaMalloc((void **)&norm, 2 * parity_norm_bytes));
    HIP_CHECK(qaMalloc((void **)&invNorm, 2 * parity_norm_bytes));
  }
  trlog[0] = trlog[1] = 0;
}
CloverField::~CloverField() {
  if (clover) (void)hipFree(clover);
  if (cloverInv) (void)hipFree(cloverInv);
  if (norm) (void)hipFree(norm);
  if (invNorm) (void)hipFree(invNorm);
}

template <typename TDev, typename THost>
__global__ void clover_load_kernel(char *dev, float *norm, size_t parity_bytes, int stride, const THost *host, int Vh) {
  using real = typename Store<TDev>::real;
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= 2 * Vh) return;
  const int parity = gid >= Vh, idx = gid - parity * Vh;
#pragma unroll
  for (int chi = 0; chi < 2; chi++) {
    real C[36];
    const THost *h = host + (((size_t)parity * Vh + idx) * 2 + chi) * 36;
#pragma unroll
    for (int k = 0; k < 36; k++) C[k] = (real)h[k];
    Planar<TDev, 36>::store(C, dev + (size_t)parity * parity_bytes + (size_t)chi * 36 * sizeof(TDev) * stride, stride, idx,
                            norm ? norm + (size_t)parity * 2 * stride : nullptr, chi * stride + idx);
  }
}

template <typename TDev, typename THost>
__global__ void clover_save_kernel(THost *host, const char *dev, const float *norm, size_t parity_bytes, int stride, int Vh) {
  using real = typename Store<TDev>::real;
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= 2 * Vh) return;
  const int parity = gid >= Vh, idx = gid - parity * Vh;
#pragma unroll
  for (int chi = 0; chi < 2; chi++) {
    real C[36];
    Planar<TDev, 36>::load(C, dev + (size_t)parity * parity_bytes + (size_t)chi * 36 * sizeof(TDev) * stride, stride, idx,
                           norm ? norm + (size_t)parity * 2 * stride : nullptr, chi * stride + idx);
    THost *h = host + (((size_t)parity * Vh + idx) * 2 + chi) * 36;
#pragma unroll
    for (int k = 0; k < 36; k++) h[k] = (THost)C[k];
  }
}

// (A^2 + mu2)^-1 (mu2 == 0: A^-1) per chiral block, computed in fp64 registers by Gauss-Jordan on the
// Hermitian 6x6 (small, setup-time only; reference lib/clover_invert.cu:56-85 uses Cholesky).
template <typename TDev>
__global__ void clover_invert_kernel(char *inv, float *invNorm, const char *A, const float *Anorm, size_t parity_bytes, int stride, int Vh,
                                     double mu2, double *trlog) {
  using real = typename Store<TDev>::real;
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= 2 * Vh) return;
  const int parity = gid >= Vh, idx = gid - parity * Vh;
  double tl = 0.0;
  for (int chi = 0; chi < 2; chi++) {
    real C[36];
    Planar<TDev, 36>::load(C, A + (size_t)parity * parity_bytes + (size_t)chi * 36 * sizeof(TDev) * stride, stride, idx,
                           Anorm ? Anorm + (size_t)parity * 2 * stride : nullptr, chi * stride + idx);
    double M[6][6][2], S[6][12][2];
    for (int r = 0; r < 6; r++)
      for (int c = 0; c < 6; c++) {
        if (r == c) { M[r][c][0] = C[r]; M[r][c][1] = 0; }
        else if (r > c) { const int k = 15 - (6 - c) * (5 - c) / 2 + r - c - 1; M[r][c][0] = C[6 + 2 * k]; M[r][c][1] = C[6 + 2 * k + 1]; }
        else { const int k = 15 - (6 - r) * (5 - r) / 2 + c - r - 1; M[r][c][0] = C[6 + 2 * k]; M[r][c][1] = -C[6 + 2 * k + 1]; }
      }
    for (int r = 0; r < 6; r++)
      for (int c = 0; c < 6; c++) {
        double re, im;
        if (mu2 != 0.0) {
          re = 0; im = 0;
          for (int k = 0; k < 6; k++) {
            re += M[r][k][0] * M[k][c][0] - M[r][k][1] * M[k][c][1];
            im += M[r][k][0] * M[k][c][1] + M[r][k][1] * M[k][c][0];
          }
          if (r == c) re += mu2;
        } else { re = M[r][c][0]; im = M[r][c][1]; }
        S[r][c][0] = re; S[r][c][1] = im;
        S[r][c + 6][0] = r == c; S[r][c + 6][1] = 0;
      }
    for (int p = 0; p < 6; p++) {
      int best = p; double bm = 0;
      for (int r = p; r < 6; r++) { const double m = S[r][p][0] * S[r][p][0] + S[r][p][1] * S[r][p][1]; if (m > bm) { bm = m; best = r; } }
      if (best != p)
        for (int c = 0; c < 12; c++)
          for (int zz = 0; zz < 2; zz++) { const double tt = S[p][c][zz]; S[p][c][zz] = S[best][c][zz]; S[best][c][zz] = tt; }
      const double pr = S[p][p][0], pi = S[p][p][1], den = pr * pr + pi * pi;
      tl += 0.5 * log(den);
      const double ir = pr / den, ii = -pi / den;
      for (int c = 0; c < 12; c++) {
        const double re = S[p][c][0] * ir - S[p][c][1] * ii, im = S[p][c][0] * ii + S[p][c][1] * ir;
        S[p][c][0] = re; S[p][c][1] = im;
      }
      for (int r = 0; r < 6; r++) {
        if (r == p) continue;
        const double fr = S[r][p][0], fi = S[r][p][1];
        for (int c = 0; c < 12; c++) {
          S[r][c][0] -= fr * S[p][c][0] - fi * S[p][c][1];
          S[r][c][1] -= fr * S[p][c][1] + fi * S[p][c][0];
        }
      }
    }
    real O[36];
    for (int r = 0; r < 6; r++) O[r] = (real)S[r][r + 6][0];
    for (int c = 0; c < 6; c++)
      for (int r = c + 1; r < 6; r++) {
        const int k = 15 - (6 - c) * (5 - c) / 2 + r - c - 1;
        O[6 + 2 * k] = (real)(0.5 * (S[r][c + 6][0] + S[c][r + 6][0]));
        O[6 + 2 * k + 1] = (real)(0.5 * (S[r][c + 6][1] - S[c][r + 6][1]));
      }
    Planar<TDev, 36>::store(O, inv + (size_t)parity * parity_bytes + (size_t)chi * 36 * sizeof(TDev) * stride, stride, idx,
                            invNorm ? invNorm + (size_t)parity * 2 * stride : nullptr, chi * stride + idx);
  }
  if (trlog) atomicAdd(&trlog[parity], tl);
}

// ---- clover term from the gauge field (reference computeFmunu, lib/field_strength_tensor.cu:30-192, and computeClover,
// lib/clover_quda.cu:41-139): one thread per site builds the six clover-leaf field strengths F_mu_nu = (Q - Q^dag)/8 and folds
// them straight into the four 3x3 colour blocks of the two chiral blocks, B1[ch] = i c (F0 -/+ F5), B2[ch] = c (F1 +/- F4 -
// i (F2 -/+ F3)); each chiral block is [[1 - B1, B2^dag], [B2, 1 + B1]].  Arithmetic in fp64 whatever the link precision;
// output = the TRUE clover matrix in packed order (the reference's native order stores half of it). ----
struct M3 { double re[9], im[9]; };
__device__ __forceinline__ void m3_mul(M3 &c, const M3 &a, const M3 &b) {
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) {
      double r = 0, m = 0;
#pragma unroll
      for (int k = 0; k < 3; k++) { r += a.re[i * 3 + k] * b.re[k * 3 + j] - a.im[i * 3 + k] * b.im[k * 3 + j]; m += a.re[i * 3 + k] * b.im[k * 3 + j] + a.im[i * 3 + k] * b.re[k * 3 + j]; }
      c.re[i * 3 + j] = r; c.im[i * 3 + j] = m;
    }
}
__device__ __forceinline__ void m3_dag(M3 &c, const M3 &a) {
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) { c.re[i * 3 + j] = a.re[j * 3 + i]; c.im[i * 3 + j] = -a.im[j * 3 + i]; }
}
struct GaugeView { const char *data; size_t link_bytes; int stride; int X[4]; int tsign, tsign_bwd; };
// U_mu at the (possibly out-of-range, wrapped) coordinates x: the forward link stored at its own site
template <typename T, int R> __device__ __forceinline__ void load_link(M3 &U, const GaugeView &g, int mu, const int *xin) {
  using real = typename Store<T>::real;
  int x[4];
#pragma unroll
  for (int d = 0; d < 4; d++) { x[d] = xin[d]; if (x[d] < 0) x[d] += g.X[d]; if (x[d] >= g.X[d]) x[d] -= g.X[d]; }
  const int par = (x[0] + x[1] + x[2] + x[3]) & 1;
  const int idx = (((x[3] * g.X[2] + x[2]) * g.X[1] + x[1]) * g.X[0] + x[0]) >> 1;
  real u[18];
  // the host links carry the anti-periodic sign on the last time slice; recon-12 stores rows 0,1 as given and has to put the
  // sign back on the reconstructed third row (in the plaquette the two boundary links of a leaf then cancel their signs)
  const real sign = (R != 18 && mu == 3 && x[3] == g.X[3] - 1) ? (real)g.tsign : (real)1;   // (recon-8: u0 of the reconstruction)
  Link<T, R>::load(u, g.data + ((size_t)par * 8 + 2 * mu) * g.link_bytes, g.stride, idx, sign);
#pragma unroll
  for (int k = 0; k < 9; k++) { U.re[k] = u[2 * k]; U.im[k] = u[2 * k + 1]; }
}
// leaf = A^(dag) B^(dag) C^(dag) D^(dag) of the links (mu_k at x_k), accumulated into Q
template <typename T, int R>
__device__ __forceinline__ void add_leaf(M3 &Q, const GaugeView &g, int m0, const int *x0, bool d0, int m1, const int *x1, bool d1, int m2, const int *x2, bool d2,
                                         int m3, const int *x3, bool d3) {
  M3 a, b, t;
  load_link<T, R>(a, g, m0, x0); if (d0) { m3_dag(t, a); a = t; }
  load_link<T, R>(b, g, m1, x1); if (d1) { m3_dag(t, b); b = t; }
  m3_mul(t, a, b);
  load_link<T, R>(b, g, m2, x2); if (d2) { m3_dag(a, b); b = a; }
  m3_mul(a, t, b);
  load_link<T, R>(b, g, m3, x3); if (d3) { m3_dag(t, b); b = t; }
  m3_mul(t, a, b);
#pragma unroll
  for (int k = 0; k < 9; k++) { Q.re[k] += t.re[k]; Q.im[k] += t.im[k]; }
}

template <typename T, int R>
__global__ void __launch_bounds__(128) clover_from_gauge_kernel(double *packed, GaugeView g, double coeff, int Vh) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= 2 * Vh) return;
  const int parity = gid >= Vh, idx = gid - parity * Vh;
  int x[4];
  {
    const int Xh = g.X[0] >> 1;
    int l = idx;
    const int xh = l % Xh; l /= Xh;
    x[1] = l % g.X[1]; l /= g.X[1];
    x[2] = l % g.X[2]; x[3] = l / g.X[2];
    x[0] = 2 * xh + ((x[1] + x[2] + x[3] + parity) & 1);
  }
  M3 b1[2], b2[2];
#pragma unroll
  for (int c = 0; c < 2; c++)
#pragma unroll
    for (int k = 0; k < 9; k++) { b1[c].re[k] = b1[c].im[k] = b2[c].re[k] = b2[c].im[k] = 0; }
  for (int mu = 1; mu < 4; mu++)
    for (int nu = 0; nu < mu; nu++) {
      M3 Q;
#pragma unroll
      for (int k = 0; k < 9; k++) Q.re[k] = Q.im[k] = 0;
      int xpm[4], xpn[4], xmm[4], xmn[4], xpn_mm[4], xpm_mn[4], xmm_mn[4];
#pragma unroll
      for (int d = 0; d < 4; d++) xpm[d] = xpn[d] = xmm[d] = xmn[d] = xpn_mm[d] = xpm_mn[d] = xmm_mn[d] = x[d];
      xpm[mu]++; xpn[nu]++; xmm[mu]--; xmn[nu]--; xpn_mm[nu]++; xpn_mm[mu]--; xpm_mn[mu]++; xpm_mn[nu]--; xmm_mn[mu]--; xmm_mn[nu]--;
      add_leaf<T, R>(Q, g, mu, x, false, nu, xpm, false, mu, xpn, true, nu, x, true);            // U(x,mu) U(x+mu,nu) U^(x+nu,mu) U^(x,nu)
      add_leaf<T, R>(Q, g, nu, x, false, mu, xpn_mm, true, nu, xmm, true, mu, xmm, false);       // U(x,nu) U^(x+nu-mu,mu) U^(x-mu,nu) U(x-mu,mu)
      add_leaf<T, R>(Q, g, nu, xmn, true, mu, xmn, false, nu, xpm_mn, false, mu, x, true);       // U^(x-nu,nu) U(x-nu,mu) U(x+mu-nu,nu) U^(x,mu)
      add_leaf<T, R>(Q, g, mu, xmm, true, nu, xmm_mn, true, mu, xmm_mn, false, nu, xmn, false);  // U^(x-mu,mu) U^(x-mu-nu,nu) U(x-mu-nu,mu) U(x-nu,nu)
      // F = (Q - Q^dag) / 8, scattered into the blocks (F index = mu(mu-1)/2 + nu)
      const int fi = mu * (mu - 1) / 2 + nu;
#pragma unroll
      for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
          const double fr = 0.125 * (Q.re[i * 3 + j] - Q.re[j * 3 + i]), fm = 0.125 * (Q.im[i * 3 + j] + Q.im[j * 3 + i]);
          const int k = i * 3 + j;
          // i c F = c (-fm + i fr);   c F = c (fr + i fm);   -i c F = c (fm - i fr)
          switch (fi) {
            case 0: b1[0].re[k] += -coeff * fm; b1[0].im[k] += coeff * fr; b1[1].re[k] += -coeff * fm; b1[1].im[k] += coeff * fr; break;
            case 5: b1[0].re[k] -= -coeff * fm; b1[0].im[k] -= coeff * fr; b1[1].re[k] += -coeff * fm; b1[1].im[k] += coeff * fr; break;
            case 1: b2[0].re[k] += coeff * fr; b2[0].im[k] += coeff * fm; b2[1].re[k] += coeff * fr; b2[1].im[k] += coeff * fm; break;
            case 4: b2[0].re[k] += coeff * fr; b2[0].im[k] += coeff * fm; b2[1].re[k] -= coeff * fr; b2[1].im[k] -= coeff * fm; break;
            case 2: b2[0].re[k] += coeff * fm; b2[0].im[k] -= coeff * fr; b2[1].re[k] += coeff * fm; b2[1].im[k] -= coeff * fr; break;
            default: b2[0].re[k] -= coeff * fm; b2[0].im[k] += coeff * fr; b2[1].re[k] += coeff * fm; b2[1].im[k] -= coeff * fr; break;  // fi == 3
          }
        }
    }
  // packed order: 6 diagonal reals, then the 15 strictly-lower entries column by column (tests/clover_reference.cpp:45-53)
  for (int ch = 0; ch < 2; ch++) {
    double *A = packed + (((size_t)parity * Vh + idx) * 2 + ch) * 36;
    for (int i = 0; i < 3; i++) { A[i] = 1.0 - b1[ch].re[i * 3 + i]; A[i + 3] = 1.0 + b1[ch].re[i * 3 + i]; }
    int k = 0;
    for (int col = 0; col < 6; col++)
      for (int row = col + 1; row < 6; row++, k++) {
        double re, im;
        const int rs = row / 3, rc = row % 3, cs = col / 3, cc = col % 3;
        if (rs == 0) { re = -b1[ch].re[rc * 3 + cc]; im = -b1[ch].im[rc * 3 + cc]; }          // spin 0 x spin 0: -B1
        else if (cs == 1) { re = b1[ch].re[rc * 3 + cc]; im = b1[ch].im[rc * 3 + cc]; }       // spin 1 x spin 1: +B1
        else { re = b2[ch].re[rc * 3 + cc]; im = b2[ch].im[rc * 3 + cc]; }                    // spin 1 x spin 0: B2
        A[6 + 2 * k] = re; A[6 + 2 * k + 1] = im;
      }
  }
}

// ---- the same construction on a grid-decomposed lattice.  A leaf that starts at x - mu, x - nu or x - mu - nu is the plaquette
// P_mu_nu of that site carried to x along the links in between (U^dag P U), so instead of gathering links from up to three
// neighbouring ranks (corners included) the field P is built once from forward-shifted links and then transported with
// ghost-aware nearest-neighbour shifts (dslash.h applyShift): Q = P + T_mu P + T_nu P + T_nu T_mu P,
// (T_mu F)(x) = U_mu(x - mu)^dag F(x - mu) U_mu(x - mu) — and U_mu(x - mu)^dag is stored at x (bidirectional links). ----
struct MatField { double *p; int Vh; __host__ __device__ double *par(int q) const { return p + (size_t)q * 24 * Vh; } };
__device__ __forceinline__ void mf_load(M3 &m, const MatField &f, int par, int idx) {
  double v[24];
  Planar<double, 24>::load(v, f.par(par), f.Vh, idx, nullptr, idx);
#pragma unroll
  for (int k = 0; k < 9; k++) { m.re[k] = v[2 * k]; m.im[k] = v[2 * k + 1]; }
}
__device__ __forceinline__ void mf_store(const M3 &m, const MatField &f, int par, int idx) {
  double v[24];
#pragma unroll
  for (int k = 0; k < 9; k++) { v[2 * k] = m.re[k]; v[2 * k + 1] = m.im[k]; }
#pragma unroll
  for (int k = 18; k < 24; k++) v[k] = 0;
  Planar<double, 24>::store(v, f.par(par), f.Vh, idx, nullptr, idx);
}
// stored matrix `slot` (2 mu: U_mu(x); 2 mu + 1: U_mu(x - mu)^dag) of site (par, idx) at time coordinate t
template <typename T, int R> __device__ __forceinline__ void load_w(M3 &U, const GaugeView &g, int par, int idx, int slot, int t) {
  using real = typename Store<T>::real;
  real sign = 1;
  if (R != 18 && (slot >> 1) == 3) sign = (slot & 1) ? (t == 0 ? (real)g.tsign_bwd : (real)1) : (t == g.X[3] - 1 ? (real)g.tsign : (real)1);
  real u[18];
  Link<T, R>::load(u, g.data + ((size_t)par * 8 + slot) * g.link_bytes, g.stride, idx, sign);
#pragma unroll
  for (int k = 0; k < 9; k++) { U.re[k] = u[2 * k]; U.im[k] = u[2 * k + 1]; }
}
__device__ __forceinline__ int time_coord(const GaugeView &g, int idx) { return idx / ((g.X[0] >> 1) * g.X[1] * g.X[2]); }

template <typename T, int R> __global__ void cl_extract_kernel(MatField out, GaugeView g, int mu, int Vh) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= 2 * Vh) return;
  const int par = gid >= Vh, idx = gid - par * Vh;
  M3 u;
  load_w<T, R>(u, g, par, idx, 2 * mu, time_coord(g, idx));
  mf_store(u, out, par, idx);
}
// P = U_mu(x) Gnu(x) Gmu(x)^dag U_nu(x)^dag with Gnu = U_nu(x + mu), Gmu = U_mu(x + nu)
template <typename T, int R> __global__ void cl_plaq_kernel(MatField P, MatField Gnu, MatField Gmu, GaugeView g, int mu, int nu, int Vh) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= 2 * Vh) return;
  const int par = gid >= Vh, idx = gid - par * Vh, t = time_coord(g, idx);
  M3 a, b, c, d;
  load_w<T, R>(a, g, par, idx, 2 * mu, t);
  mf_load(b, Gnu, par, idx);
  m3_mul(c, a, b);
  mf_load(b, Gmu, par, idx); m3_dag(a, b);
  m3_mul(d, c, a);
  load_w<T, R>(b, g, par, idx, 2 * nu, t); m3_dag(a, b);
  m3_mul(c, d, a);
  mf_store(c, P, par, idx);
}
// out = W S W^dag with W = U_mu(x - mu)^dag (slot 2 mu + 1) and S = F(x - mu) already shifted to x
template <typename T, int R> __global__ void cl_transport_kernel(MatField out, MatField S, GaugeView g, int mu, int Vh) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= 2 * Vh) return;
  const int par = gid >= Vh, idx = gid - par * Vh;
  M3 w, s, a, b;
  load_w<T, R>(w, g, par, idx, 2 * mu + 1, time_coord(g, idx));
  mf_load(s, S, par, idx);
  m3_mul(a, w, s);
  m3_dag(b, w);
  m3_mul(s, a, b);
  mf_store(s, out, par, idx);
}
// F = (Q - Q^dag)/8 with Q = P + A + B + C, scattered into the four colour blocks (same table as clover_from_gauge_kernel)
__global__ void cl_accum_kernel(MatField b1_0, MatField b1_1, MatField b2_0, MatField b2_1, MatField P, MatField A, MatField B, MatField C, int fi, double coeff,
                                int first, int Vh) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= 2 * Vh) return;
  const int par = gid >= Vh, idx = gid - par * Vh;
  M3 Q, m, b1[2], b2[2];
  mf_load(Q, P, par, idx);
  mf_load(m, A, par, idx);
#pragma unroll
  for (int k = 0; k < 9; k++) { Q.re[k] += m.re[k]; Q.im[k] += m.im[k]; }
  mf_load(m, B, par, idx);
#pragma unroll
  for (int k = 0; k < 9; k++) { Q.re[k] += m.re[k]; Q.im[k] += m.im[k]; }
  mf_load(m, C, par, idx);
#pragma unroll
  for (int k = 0; k < 9; k++) { Q.re[k] += m.re[k]; Q.im[k] += m.im[k]; }
  if (first) {
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
      for (int k = 0; k < 9; k++) { b1[c].re[k] = b1[c].im[k] = b2[c].re[k] = b2[c].im[k] = 0; }
  } else {
    mf_load(b1[0], b1_0, par, idx); mf_load(b1[1], b1_1, par, idx); mf_load(b2[0], b2_0, par, idx); mf_load(b2[1], b2_1, par, idx);
  }
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) {
      const double fr = 0.125 * (Q.re[i * 3 + j] - Q.re[j * 3 + i]), fm = 0.125 * (Q.im[i * 3 + j] + Q.im[j * 3 + i]);
      const int k = i * 3 + j;
      switch (fi) {
        case 0: b1[0].re[k] += -coeff * fm; b1[0].im[k] += coeff * fr; b1[1].re[k] += -coeff * fm; b1[1].im[k] += coeff * fr; break;
        case 5: b1[0].re[k] -= -coeff * fm; b1[0].im[k] -= coeff * fr; b1[1].re[k] += -coeff * fm; b1[1].im[k] += coeff * fr; break;
        case 1: b2[0].re[k] += coeff * fr; b2[0].im[k] += coeff * fm; b2[1].re[k] += coeff * fr; b2[1].im[k] += coeff * fm; break;
        case 4: b2[0].re[k] += coeff * fr; b2[0].im[k] += coeff * fm; b2[1].re[k] -= coeff * fr; b2[1].im[k] -= coeff * fm; break;
        case 2: b2[0].re[k] += coeff * fm; b2[0].im[k] -= coeff * fr; b2[1].re[k] += coeff * fm; b2[1].im[k] -= coeff * fr; break;
        default: b2[0].re[k] -= coeff * fm; b2[0].im[k] += coeff * fr; b2[1].re[k] += coeff * fm; b2[1].im[k] -= coeff * fr; break;
      }
    }
  mf_store(b1[0], b1_0, par, idx); mf_store(b1[1], b1_1, par, idx); mf_store(b2[0], b2_0, par, idx); mf_store(b2[1], b2_1, par, idx);
}
__global__ void cl_pack_kernel(double *packed, MatField b1_0, MatField b1_1, MatField b2_0, MatField b2_1, int Vh) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= 2 * Vh) return;
  const int par = gid >= Vh, idx = gid - par * Vh;
  for (int ch = 0; ch < 2; ch++) {
    M3 b1, b2;
    mf_load(b1, ch ? b1_1 : b1_0, par, idx);
    mf_load(b2, ch ? b2_1 : b2_0, par, idx);
    double *A = packed + (((size_t)par * Vh + idx) * 2 + ch) * 36;
    for (int i = 0; i < 3; i++) { A[i] = 1.0 - b1.re[i * 3 + i]; A[i + 3] = 1.0 + b1.re[i * 3 + i]; }
    int k = 0;
    for (int col = 0; col < 6; col++)
      for (int row = col + 1; row < 6; row++, k++) {
        const int rs = row / 3, rc = row % 3, cs = col / 3, cc = col % 3;
        double re, im;
        if (rs == 0) { re = -b1.re[rc * 3 + cc]; im = -b1.im[rc * 3 + cc]; }
        else if (cs == 1) { re = b1.re[rc * 3 + cc]; im = b1.im[rc * 3 + cc]; }
        else { re = b2.re[rc * 3 + cc]; im = b2.im[rc * 3 + cc]; }
        A[6 + 2 * k] = re; A[6 + 2 * k + 1] = im;
      }
  }
}

void applyShift(double *out, const double *in, const LatticeGeom &g, int stride, int parity, int dir);  // dslash.hip

template <typename T, int R> static void cloverFromGaugeDecomposed(double *stage, const GaugeField &U, const GaugeView &g, double coeff) {
  const LatticeGeom &geom = U.geom;
  const int Vh = geom.Vh, bs = 128, nb = (2 * Vh + bs - 1) / bs;
  const size_t fieldDoubles = (size_t)2 * 24 * Vh;
  double *pool = nullptr;
  HIP_CHECK(qaMalloc((void **)&pool, 9 * fieldDoubles * sizeof(double)));
  MatField W[5], b1[2], b2[2];
  for (int i = 0; i < 5; i++) W[i] = {pool + i * fieldDoubles, Vh};
  b1[0] = {pool + 5 * fieldDoubles, Vh}; b1[1] = {pool + 6 * fieldDoubles, Vh};
  b2[0] = {pool + 7 * fieldDoubles, Vh}; b2[1] = {pool + 8 * fieldDoubles, Vh};
  hipStream_t s = computeStream();
  auto shift = [&](MatField out, MatField in, int dir) {   // out(x) = in(x + dhat(dir)), both parities
    for (int par = 0; par < 2; par++) applyShift(out.par(par), in.par(1 - par), geom, Vh, par, dir);
  };
  bool first = true;
  for (int mu = 1; mu < 4; mu++)
    for (int nu = 0; nu < mu; nu++) {
      hipLaunchKernelGGL((cl_extract_kernel<T, R>), dim3(nb), dim3(bs), 0, s, W[0], g, nu, Vh);
      hipLaunchKernelGGL((cl_extract_kernel<T, R>), dim3(nb), dim3(bs), 0, s, W[1], g, mu, Vh);
      shift(W[2], W[0], 2 * mu);   // U_nu(x + mu)
      shift(W[3], W[1], 2 * nu);   // U_mu(x + nu)
      hipLaunchKernelGGL((cl_plaq_kernel<T, R>), dim3(nb), dim3(bs), 0, s, W[4], W[2], W[3], g, mu, nu, Vh);            // P
      shift(W[0], W[4], 2 * mu + 1);
      hipLaunchKernelGGL((cl_transport_kernel<T, R>), dim3(nb), dim3(bs), 0, s, W[1], W[0], g, mu, Vh);                  // A = T_mu P
      shift(W[0], W[4], 2 * nu + 1);
      hipLaunchKernelGGL((cl_transport_kernel<T, R>), dim3(nb), dim3(bs), 0, s, W[2], W[0], g, nu, Vh);                  // B = T_nu P
      shift(W[0], W[1], 2 * nu + 1);
      hipLaunchKernelGGL((cl_transport_kernel<T, R>), dim3(nb), dim3(bs), 0, s, W[3], W[0], g, nu, Vh);                  // C = T_nu T_mu P
      hipLaunchKernelGGL(cl_accum_kernel, dim3(nb), dim3(bs), 0, s, b1[0], b1[1], b2[0], b2[1], W[4], W[1], W[2], W[3], mu * (mu - 1) / 2 + nu, coeff,
                         first ? 1 : 0, Vh);
      HIP_CHECK(hipGetLastError());
      first = false;
    }
  hipLaunchKernelGGL(cl_pack_kernel, dim3(nb), dim3(bs), 0, s, stage, b1[0], b1[1], b2[0], b2[1], Vh);
  HIP_CHECK(hipGetLastError());
  HIP_CHECK(hipStreamSynchronize(s));
  HIP_CHECK(hipFree(pool));
}

void CloverField::computeFromGauge(const GaugeField &U, double coeff) {
  if (!(U.geom == geom)) errorQuda("gauge and clover geometry differ");
  if (U.anisotropy != 1.0) errorQuda("cannot compute anisotropic clover field");
  bool decomposed = false;
  for (int d = 0; d < 4; d++) decomposed = decomposed || commGrid().partitioned(d);
  { const char *e = getenv("QUDA_AMD_CLOVER_TRANSPORT"); if (e && atoi(e)) decomposed = true; }  // force the transport formulation (tests)
  const size_t n = (size_t)geom.V * 72 * sizeof(double);
  double *stage = (double *)stagingBuffer(n);
  GaugeView g;
  g.data = (const char *)U.data; g.link_bytes = U.link_bytes; g.stride = U.stride;
  for (int d = 0; d < 4; d++) g.X[d] = geom.X[d];
  // anti-periodic sign of the reconstructed third row (recon-12): on the last / first rank in t only
  const bool first_t = commGrid().coords[3] == 0, last_t = commGrid().coords[3] == commGrid().dims[3] - 1;
  g.tsign = (U.t_boundary == QUDA_ANTI_PERIODIC_T && last_t) ? -1 : 1;
  g.tsign_bwd = (U.t_boundary == QUDA_ANTI_PERIODIC_T && first_t) ? -1 : 1;
  const int bs = 128, nb = (2 * geom.Vh + bs - 1) / bs;
  const bool r12 = U.reconstruct == QUDA_RECONSTRUCT_12, r8 = U.reconstruct == QUDA_RECONSTRUCT_8;
  if (decomposed) {
#define QA_CD(T) { if (r12) cloverFromGaugeDecomposed<T, 12>(stage, U, g, coeff); else if (r8) cloverFromGaugeDecomposed<T, 8>(stage, U, g, coeff); else cloverFromGaugeDecomposed<T, 18>(stage, U, g, coeff); }
    switch (U.precision) {
      case QUDA_DOUBLE_PRECISION: QA_CD(double) break;
      case QUDA_SINGLE_PRECISION: QA_CD(float) break;
      case QUDA_HALF_PRECISION: QA_CD(short) break;
      default: errorQuda("bad gauge precision %d", U.precision);
    }
#undef QA_CD
  } else {
#define QA_CG(T) \
  if (r12) hipLaunchKernelGGL((clover_from_gauge_kernel<T, 12>), dim3(nb), dim3(bs), 0, computeStream(), stage, g, coeff, geom.Vh); \
  else if (r8) hipLaunchKernelGGL((clover_from_gauge_kernel<T, 8>), dim3(nb), dim3(bs), 0, computeStream(), stage, g, coeff, geom.Vh); \
  else hipLaunchKernelGGL((clover_from_gauge_kernel<T, 18>), dim3(nb), dim3(bs), 0, computeStream(), stage, g, coeff, geom.Vh);
  switch (U.precision) {
    case QUDA_DOUBLE_PRECISION: QA_CG(double) break;
    case QUDA_SINGLE_PRECISION: QA_CG(float) break;
    case QUDA_HALF_PRECISION: QA_CG(short) break;
    default: errorQuda("bad gauge precision %d", U.precision);
  }
#undef QA_CG
  }
  HIP_CHECK(hipGetLastError());
  const int bl = 256, nl = (2 * geom.Vh + bl - 1) / bl;
#define QA_CL2(TD) hipLaunchKernelGGL((clover_load_kernel<TD, double>), dim3(nl), dim3(bl), 0, computeStream(), (char *)clover, norm, parity_bytes, stride, (const double *)stage, geom.Vh)
  QA_DISPATCH_DEV(precision, QA_CL2(TD));
#undef QA_CL2
  HIP_CHECK(hipGetLastError());
  HIP_CHECK(hipStreamSynchronize(computeStream()));
}

template <typename TDev, typename THost> static void cloverLoad(CloverField &c, void *dev, float *nrm, const void *host) {
  const size_t n = (size_t)c.geom.V * 72 * sizeof(THost);
  void *stage = stagingBuffer(n);
  HIP_CHECK(hipMemcpyAsync(stage, host, n, hipMemcpyHostToDevice, computeStream()));
  const int bs = 256, nb = (2 * c.geom.Vh + bs - 1) / bs;
  hipLaunchKernelGGL((clover_load_kernel<TDev, THost>), dim3(nb), dim3(bs), 0, computeStream(), (char *)dev, nrm, c.parity_bytes, c.stride,
                     (const THost *)stage, c.geom.Vh);
  HIP_CHECK(hipGetLastError());
  HIP_CHECK(hipStreamSynchronize(computeStream()));
}

void CloverField::loadPacked(const void *h_clover, const void *h_inv, QudaPrecision cpu_prec) {
#define QA_CL(TD, DEV, NRM, HOST)                                                   \
  if (cpu_prec == QUDA_DOUBLE_PRECISION) cloverLoad<TD, double>(*this, DEV, NRM, HOST); \
  else cloverLoad<TD, float>(*this, DEV, NRM, HOST);
  if (h_clover) { QA_DISPATCH_DEV(precision, QA_CL(TD, clover, norm, h_clover)); }
  if (h_inv) { QA_DISPATCH_DEV(precision, QA_CL(TD, cloverInv, invNorm, h_inv)); }
#undef QA_CL
}

template <typename TDev> static void launchCloverInvert(CloverField &c, int nb, int bs, double mu2, double *d_trlog) {
  hipLaunchKernelGGL((clover_invert_kernel<TDev>), dim3(nb), dim3(bs), 0, computeStream(), (char *)c.cloverInv, c.invNorm, (const char *)c.clover,
                     c.norm, c.parity_bytes, c.stride, c.geom.Vh, mu2, d_trlog);
}

void CloverField::computeInverse(double mu2_) {
  mu2 = mu2_;
  twisted = mu2_ != 0.0;
  double *d_trlog = nullptr;
  HIP_CHECK(qaMalloc((void **)&d_trlog, 2 * sizeof(double)));
  HIP_CHECK(hipMemsetAsync(d_trlog, 0, 2 * sizeof(double), computeStream()));
  const int bs = 128, nb = (2 * geom.Vh + bs - 1) / bs;
  QA_DISPATCH_DEV(precision, (launchCloverInvert<TD>(*this, nb, bs, mu2_, d_trlog)));
  HIP_CHECK(hipGetLastError());
  HIP_CHECK(hipMemcpyAsync(trlog, d_trlog, 2 * sizeof(double), hipMemcpyDeviceToHost, computeStream()));
  HIP_CHECK(hipStreamSynchronize(computeStream()));
  HIP_CHECK(hipFree(d_trlog));
}

void CloverField::savePackedInverse(void *h_inv, QudaPrecision cpu_prec) const {
  const size_t n = (size_t)geom.V * 72 * (int)cpu_prec;
  void *stage = stagingBuffer(n);
  const int bs = 256, nb = (2 * geom.Vh + bs - 1) / bs;
#define QA_CS(TD)                                                                                                              \
  if (cpu_prec == QUDA_DOUBLE_PRECISION)                                                                                       \
    hipLaunchKernelGGL((clover_save_kernel<TD, double>), dim3(nb), dim3(bs), 0, computeStream(), (double *)stage, (const char *)cloverInv, invNorm, \
                       parity_bytes, stride, geom.Vh);                                                                         \
  else                                                                                                                         \
    hipLaunchKernelGGL((clover_save_kernel<TD, float>), dim3(nb), dim3(bs), 0, computeStream(), (float *)stage, (const char *)cloverInv, invNorm,  \
                       parity_bytes, stride, geom.Vh);
  QA_DISPATCH_DEV(precision, QA_CS(TD));
#undef QA_CS
  HIP_CHECK(hipGetLastError());
  HIP_CHECK(hipMemcpyAsync(h_inv, stage, n, hipMemcpyDeviceToHost, computeStream()));
  HIP_CHECK(hipStreamSynchronize(computeStream()));
}

void CloverField::savePacked(void *h_inv, QudaPrecision cpu_prec) const {
  const size_t n = (size_t)geom.V * 72 * (int)cpu_prec;
  void *stage = stagingBuffer(n);
  const int bs = 256, nb = (2 * geom.Vh + bs - 1) / bs;
#define QA_CS(TD)                                                                                                              \
  if (cpu_prec == QUDA_DOUBLE_PRECISION)                                                                                       \
    hipLaunchKernelGGL((clover_save_kernel<TD, double>), dim3(nb), dim3(bs), 0, computeStream(), (double *)stage, (const char *)clover, norm, \
                       parity_bytes, stride, geom.Vh);                                                                         \
  else                                                                                                                         \
    hipLaunchKernelGGL((clover_save_kernel<TD, float>), dim3(nb), dim3(bs), 0, computeStream(), (float *)stage, (const char *)clover, norm,  \
                       parity_bytes, stride, geom.Vh);
  QA_DISPATCH_DEV(precision, QA_CS(TD));
#undef QA_CS
  HIP_CHECK(hipGetLastError());
  HIP_CHECK(hipMemcpyAsync(h_inv, stage, n, hipMemcpyDeviceToHost, computeStream()));
  HIP_CHECK(hipStreamSynchronize(computeStream()));
}


void comm_allreduce(double *data, int n);   // comm.cpp

// ================================================================================================================
// APE smearing of the spatial links and the plaquette (reference lib/gauge_ape.cu:44-156,
// include/su3_project.cuh:23-124, lib/interface_quda.cpp:5565-5640, lib/gauge_plaq.cu:38-152).  Same transport formulation as
// the decomposed clover construction above: forward links as 3x3 matrix fields, neighbours through the ghost-aware
// nearest-neighbour shift (applyShift), so a grid-decomposed lattice needs no extended gauge halo and no corner exchange
// (the reference: copyExtendedGauge + exchangeExtendedGhost every step).  fp64 throughout.
//   upper staple of (x, nu) in the mu-nu plane:  U_mu(x) U_nu(x+mu) U_mu(x+nu)^dag
//   lower staple:                                [U_mu^dag U_nu T_nu(U_mu)](x - mu)      (built at x - mu, then shifted to x)
__device__ __forceinline__ void m3_det(double &dr, double &di, const M3 &a) {
  dr = 0; di = 0;
#pragma unroll
  for (int c = 0; c < 3; c++) {
    const int c1 = (c + 1) % 3, c2 = (c + 2) % 3;
    const double mr = (a.re[3 + c1] * a.re[6 + c2] - a.im[3 + c1] * a.im[6 + c2]) - (a.re[3 + c2] * a.re[6 + c1] - a.im[3 + c2] * a.im[6 + c1]);
    const double mi = (a.re[3 + c1] * a.im[6 + c2] + a.im[3 + c1] * a.re[6 + c2]) - (a.re[3 + c2] * a.im[6 + c1] + a.im[3 + c2] * a.re[6 + c1]);
    dr += a.re[c] * mr - a.im[c] * mi;
    di += a.re[c] * mi + a.im[c] * mr;
  }
}
__device__ __forceinline__ void m3_inverse(M3 &o, const M3 &a) {   // adjugate / determinant
  double dr, di;
  m3_det(dr, di, a);
  const double n = dr * dr + di * di, ir = dr / n, ii = -di / n;
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) {
      const int r1 = (j + 1) % 3, r2 = (j + 2) % 3, c1 = (i + 1) % 3, c2 = (i + 2) % 3;
      const double cr = (a.re[3 * r1 + c1] * a.re[3 * r2 + c2] - a.im[3 * r1 + c1] * a.im[3 * r2 + c2]) - (a.re[3 * r1 + c2] * a.re[3 * r2 + c1] - a.im[3 * r1 + c2] * a.im[3 * r2 + c1]);
      const double ci = (a.re[3 * r1 + c1] * a.im[3 * r2 + c2] + a.im[3 * r1 + c1] * a.re[3 * r2 + c2]) - (a.re[3 * r1 + c2] * a.im[3 * r2 + c1] + a.im[3 * r1 + c2] * a.re[3 * r2 + c1]);
      o.re[3 * i + j] = cr * ir - ci * ii;
      o.im[3 * i + j] = cr * ii + ci * ir;
    }
}
// su3_project.cuh polarSu3: Newton iteration X <- (X + X^-dag)/2 until X is unitary to tol (elementwise, X vs (X^-1)^dag),
// then the phase of the determinant is divided out.  Capped at 100 sweeps (the reference loops until the test passes).
__device__ __forceinline__ void polar_su3(M3 &m, double tol) {
  M3 out = m, inv;
  m3_inverse(inv, out);
  for (int sweep = 0; sweep < 100; sweep++) {
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
      for (int j = 0; j < 3; j++) {
        out.re[3 * i + j] = 0.5 * (out.re[3 * i + j] + inv.re[3 * j + i]);
        out.im[3 * i + j] = 0.5 * (out.im[3 * i + j] - inv.im[3 * j + i]);
      }
    m3_inverse(inv, out);
    bool bad = false;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
      for (int j = 0; j < 3; j++) bad = bad || fabs(out.re[3 * i + j] - inv.re[3 * j + i]) > tol || fabs(out.im[3 * i + j] + inv.im[3 * j + i]) > tol;
    if (!bad) break;
  }
  double dr, di;
  m3_det(dr, di, out);
  const double mod = pow(dr * dr + di * di, 1.0 / 6.0), angle = atan2(di, dr) / -3.0;
  const double cr = cos(angle) / mod, ci = sin(angle) / mod;
#pragma unroll
  for (int k = 0; k < 9; k++) { m.re[k] = out.re[k] * cr - out.im[k] * ci; m.im[k] = out.re[k] * ci + out.im[k] * cr; }
}

// out (+)= op(A) op(B), op = identity or Hermitian conjugate
__global__ void mf_mul_kernel(MatField out, MatField A, MatField B, int dagA, int dagB, int accumulate, int Vh) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= 2 * Vh) return;
  const int par = gid >= Vh, idx = gid - par * Vh;
  M3 a, b, t, c;
  mf_load(a, A, par, idx); if (dagA) { m3_dag(t, a); a = t; }
  mf_load(b, B, par, idx); if (dagB) { m3_dag(t, b); b = t; }
  m3_mul(c, a, b);
  if (accumulate) {
    mf_load(t, out, par, idx);
#pragma unroll
    for (int k = 0; k < 9; k++) { c.re[k] += t.re[k]; c.im[k] += t.im[k]; }
  }
  mf_store(c, out, par, idx);
}
__global__ void mf_add_kernel(MatField out, MatField in, int Vh) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= 2 * Vh) return;
  const int par = gid >= Vh, idx = gid - par * Vh;
  M3 a, b;
  mf_load(a, out, par, idx); mf_load(b, in, par, idx);
#pragma unroll
  for (int k = 0; k < 9; k++) { a.re[k] += b.re[k]; a.im[k] += b.im[k]; }
  mf_store(a, out, par, idx);
}
// computeAPEStep: TestU = (1 - alpha) + alpha/4 S U^dag, projected; U' = TestU U
__global__ void ape_project_kernel(MatField Unew, MatField S, MatField U, double alpha, double tol, int Vh) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= 2 * Vh) return;
  const int par = gid >= Vh, idx = gid - par * Vh;
  M3 s, u, ud, t;
  mf_load(s, S, par, idx); mf_load(u, U, par, idx);
  const double f = alpha / 4.0;
#pragma unroll
  for (int k = 0; k < 9; k++) { s.re[k] *= f; s.im[k] *= f; }
  m3_dag(ud, u);
  m3_mul(t, s, ud);
  t.re[0] += 1.0 - alpha; t.re[4] += 1.0 - alpha; t.re[8] += 1.0 - alpha;
  polar_su3(t, tol);
  m3_mul(s, t, u);
  mf_store(s, Unew, par, idx);
}
// sum over sites of Re tr [A B C^dag D^dag] into acc[slot] (block tree + one atomic per block)
__global__ void __launch_bounds__(256) plaq_trace_kernel(double *acc, int slot, MatField A, MatField B, MatField Cm, MatField D, int Vh) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  double tr = 0;
  if (gid < 2 * Vh) {
    const int par = gid >= Vh, idx = gid - par * Vh;
    M3 a, b, t, t2, d;
    mf_load(a, A, par, idx); mf_load(b, B, par, idx);
    m3_mul(t, a, b);
    mf_load(a, Cm, par, idx); m3_dag(d, a); m3_mul(t2, t, d);
    mf_load(a, D, par, idx); m3_dag(d, a); m3_mul(t, t2, d);
    tr = t.re[0] + t.re[4] + t.re[8];
  }
  __shared__ double lds[4];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) tr += __shfl_down(tr, off, 64);
  if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = tr;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(acc + slot, lds[0] + lds[1] + lds[2] + lds[3]);
}
// forward links of one direction -> host QDP order (even sites then odd, 18 reals per site)
__global__ void mf_to_qdp_kernel(double *qdp, MatField F, int Vh) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= 2 * Vh) return;
  const int par = gid >= Vh, idx = gid - par * Vh;
  M3 u;
  mf_load(u, F, par, idx);
  double *o = qdp + (size_t)gid * 18;
#pragma unroll
  for (int k = 0; k < 9; k++) { o[2 * k] = u.re[k]; o[2 * k + 1] = u.im[k]; }
}

static GaugeView viewOf(const GaugeField &U) {
  GaugeView g;
  g.data = (const char *)U.data; g.link_bytes = U.link_bytes; g.stride = U.stride;
  for (int d = 0; d < 4; d++) g.X[d] = U.geom.X[d];
  const bool first_t = commGrid().coords[3] == 0, last_t = commGrid().coords[3] == commGrid().dims[3] - 1;
  g.tsign = (U.t_boundary == QUDA_ANTI_PERIODIC_T && last_t) ? -1 : 1;
  g.tsign_bwd = (U.t_boundary == QUDA_ANTI_PERIODIC_T && first_t) ? -1 : 1;
  return g;
}
static void extractForwardLinks(MatField F, const GaugeField &U, int mu) {
  const GaugeView g = viewOf(U);
  const int Vh = U.geom.Vh, bs = 128, nb = (2 * Vh + bs - 1) / bs;
  const bool r12 = U.reconstruct == QUDA_RECONSTRUCT_12, r8 = U.reconstruct == QUDA_RECONSTRUCT_8;
#define QA_EX(T) { if (r12) hipLaunchKernelGGL((cl_extract_kernel<T, 12>), dim3(nb), dim3(bs), 0, computeStream(), F, g, mu, Vh); \
                   else if (r8) hipLaunchKernelGGL((cl_extract_kernel<T, 8>), dim3(nb), dim3(bs), 0, computeStream(), F, g, mu, Vh); \
                   else hipLaunchKernelGGL((cl_extract_kernel<T, 18>), dim3(nb), dim3(bs), 0, computeStream(), F, g, mu, Vh); }
  switch (U.precision) {
    case QUDA_DOUBLE_PRECISION: QA_EX(double) break;
    case QUDA_SINGLE_PRECISION: QA_EX(float) break;
    case QUDA_HALF_PRECISION: QA_EX(short) break;
    default: errorQuda("bad gauge precision %d", U.precision);
  }
#undef QA_EX
  HIP_CHECK(hipGetLastError());
}

GaugeField *apeSmear(const GaugeField &U, unsigned nSteps, double alpha) {
  const LatticeGeom &geom = U.geom;
  const int Vh = geom.Vh, bs = 128, nb = (2 * Vh + bs - 1) / bs;
  const size_t fieldDoubles = (size_t)2 * 24 * Vh;
  double *pool = nullptr;
  HIP_CHECK(qaMalloc((void **)&pool, 13 * fieldDoubles * sizeof(double)));
  MatField F[4], G[3], S, W1, W2, T1, T2;
  for (int i = 0; i < 4; i++) F[i] = {pool + i * fieldDoubles, Vh};
  for (int i = 0; i < 3; i++) G[i] = {pool + (4 + i) * fieldDoubles, Vh};
  S = {pool + 7 * fieldDoubles, Vh}; W1 = {pool + 8 * fieldDoubles, Vh}; W2 = {pool + 9 * fieldDoubles, Vh};
  T1 = {pool + 10 * fieldDoubles, Vh}; T2 = {pool + 11 * fieldDoubles, Vh};
  hipStream_t s = computeStream();
  auto shift = [&](MatField out, MatField in, int dir) {   // out(x) = in(x + dhat(dir)), both parities
    for (int par = 0; par < 2; par++) applyShift(out.par(par), in.par(1 - par), geom, Vh, par, dir);
  };
  for (int mu = 0; mu < 4; mu++) extractForwardLinks(F[mu], U, mu);
  const double tol = 1e-15;   // DOUBLE_TOL, lib/gauge_ape.cu:9
  for (unsigned step = 0; step < nSteps; step++) {
    for (int nu = 0; nu < 3; nu++) {
      HIP_CHECK(hipMemsetAsync(S.p, 0, fieldDoubles * sizeof(double), s));
      for (int mu = 0; mu < 3; mu++) {
        if (mu == nu) continue;
        shift(W1, F[nu], 2 * mu);   // U_nu(x + mu)
        shift(W2, F[mu], 2 * nu);   // U_mu(x + nu)
        hipLaunchKernelGGL(mf_mul_kernel, dim3(nb), dim3(bs), 0, s, T1, F[mu], W1, 0, 0, 0, Vh);   // U_mu(x) U_nu(x+mu)
        hipLaunchKernelGGL(mf_mul_kernel, dim3(nb), dim3(bs), 0, s, S, T1, W2, 0, 1, 1, Vh);       // S += ... U_mu(x+nu)^dag
        hipLaunchKernelGGL(mf_mul_kernel, dim3(nb), dim3(bs), 0, s, T1, F[mu], F[nu], 1, 0, 0, Vh); // U_mu(x)^dag U_nu(x)
        hipLaunchKernelGGL(mf_mul_kernel, dim3(nb), dim3(bs), 0, s, T2, T1, W2, 0, 0, 0, Vh);      // ... U_mu(x+nu)
        shift(T1, T2, 2 * mu + 1);  // carried from x - mu to x
        hipLaunchKernelGGL(mf_add_kernel, dim3(nb), dim3(bs), 0, s, S, T1, Vh);
        HIP_CHECK(hipGetLastError());
      }
      hipLaunchKernelGGL(ape_project_kernel, dim3(nb), dim3(bs), 0, s, G[nu], S, F[nu], alpha, tol, Vh);
      HIP_CHECK(hipGetLastError());
    }
    // every direction of a step is smeared from the links of the previous step (the reference reads a copy, :5621-5627)
    for (int nu = 0; nu < 3; nu++) std::swap(F[nu], G[nu]);
  }
  // back through the loader: it builds the bidirectional layout and fetches the backward links that live on the neighbour ranks
  std::vector<std::vector<double>> host(4, std::vector<double>((size_t)geom.V * 18));
  void *ptr[4];
  double *stage = (double *)stagingBuffer((size_t)geom.V * 18 * sizeof(double));
  for (int mu = 0; mu < 4; mu++) {
    hipLaunchKernelGGL(mf_to_qdp_kernel, dim3(nb), dim3(bs), 0, s, stage, F[mu], Vh);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipMemcpyAsync(host[mu].data(), stage, (size_t)geom.V * 18 * sizeof(double), hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    ptr[mu] = host[mu].data();
  }
  HIP_CHECK(hipFree(pool));
  GaugeField *out = new GaugeField(geom, QUDA_DOUBLE_PRECISION, QUDA_RECONSTRUCT_NO, U.t_boundary, U.anisotropy);
  out->loadQDP(ptr, QUDA_DOUBLE_PRECISION);
  return out;
}

void saveGaugeQDP(const GaugeField &U, void *const h_gauge[4], QudaPrecision cpu_prec) {
  if (cpu_prec != QUDA_DOUBLE_PRECISION) errorQuda("saving links: fp64 host fields only");
  const LatticeGeom &geom = U.geom;
  const int Vh = geom.Vh, bs = 128, nb = (2 * Vh + bs - 1) / bs;
  double *pool = nullptr;
  HIP_CHECK(qaMalloc((void **)&pool, (size_t)2 * 24 * Vh * sizeof(double)));
  MatField F = {pool, Vh};
  double *stage = (double *)stagingBuffer((size_t)geom.V * 18 * sizeof(double));
  for (int mu = 0; mu < 4; mu++) {
    extractForwardLinks(F, U, mu);
    hipLaunchKernelGGL(mf_to_qdp_kernel, dim3(nb), dim3(bs), 0, computeStream(), stage, F, Vh);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipMemcpyAsync(h_gauge[mu], stage, (size_t)geom.V * 18 * sizeof(double), hipMemcpyDeviceToHost, computeStream()));
    HIP_CHECK(hipStreamSynchronize(computeStream()));
  }
  HIP_CHECK(hipFree(pool));
}

// plq[0] = mean of the spatial and temporal averages, plq[1] = spatial, plq[2] = temporal (lib/gauge_plaq.cu:129-153)
void plaquette(const GaugeField &U, double plq[3]) {
  const LatticeGeom &geom = U.geom;
  const int Vh = geom.Vh, nb = (2 * Vh + 255) / 256;
  const size_t fieldDoubles = (size_t)2 * 24 * Vh;
  double *pool = nullptr, *d_acc = nullptr;
  HIP_CHECK(qaMalloc((void **)&pool, 6 * fieldDoubles * sizeof(double)));
  HIP_CHECK(qaMalloc((void **)&d_acc, 2 * sizeof(double)));
  HIP_CHECK(hipMemsetAsync(d_acc, 0, 2 * sizeof(double), computeStream()));
  MatField F[4], W1, W2;
  for (int i = 0; i < 4; i++) F[i] = {pool + i * fieldDoubles, Vh};
  W1 = {pool + 4 * fieldDoubles, Vh}; W2 = {pool + 5 * fieldDoubles, Vh};
  auto shift = [&](MatField out, MatField in, int dir) {
    for (int par = 0; par < 2; par++) applyShift(out.par(par), in.par(1 - par), geom, Vh, par, dir);
  };
  for (int mu = 0; mu < 4; mu++) extractForwardLinks(F[mu], U, mu);
  for (int mu = 0; mu < 3; mu++)
    for (int nu = mu + 1; nu < 4; nu++) {
      shift(W1, F[nu], 2 * mu);
      shift(W2, F[mu], 2 * nu);
      hipLaunchKernelGGL(plaq_trace_kernel, dim3(nb), dim3(256), 0, computeStream(), d_acc, nu < 3 ? 0 : 1, F[mu], W1, W2, F[nu], Vh);
      HIP_CHECK(hipGetLastError());
    }
  double h[2];
  HIP_CHECK(hipMemcpyAsync(h, d_acc, 2 * sizeof(double), hipMemcpyDeviceToHost, computeStream()));
  HIP_CHECK(hipStreamSynchronize(computeStream()));
  HIP_CHECK(hipFree(pool)); HIP_CHECK(hipFree(d_acc));
  comm_allreduce(h, 2);
  const double norm = 9.0 * (double)geom.V * commGrid().size;
  plq[1] = h[0] / norm; plq[2] = h[1] / norm; plq[0] = 0.5 * (plq[1] + plq[2]);
}

}  // namespace quda
