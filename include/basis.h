// basis.h — spin-basis rotation between the host UKQCD (non-relativistic) basis and the device DeGrand-Rossi basis.
#pragma once

#include <hip/hip_runtime.h>

namespace quda {

enum BasisChange { BASIS_NONE = 0, BASIS_UKQCD_TO_DR = 1, BASIS_DR_TO_UKQCD = 2 };

template <typename real> __device__ __forceinline__ void rotate_basis(real *o, const real *in, int change) {
  // DR -> UKQCD is the reference's NonRelBasis, UKQCD -> DR its RelBasis (lib/copy_color_spinor.cuh:49-91)
  const real k = (real)0.70710678118654752440;
#pragma unroll
  for (int c = 0; c < 6; c++) {
    const real i0 = in[c], i1 = in[6 + c], i2 = in[12 + c], i3 = in[18 + c];
    if (change == BASIS_UKQCD_TO_DR) {
      o[c] = -k * (i1 + i3); o[6 + c] = k * (i0 + i2); o[12 + c] = k * (i3 - i1); o[18 + c] = k * (i0 - i2);
    } else {
      o[c] = k * (i1 + i3); o[6 + c] = -k * (i0 + i2); o[12 + c] = k * (i1 - i3); o[18 + c] = k * (i2 - i0);
    }
  }
}

}  // namespace quda
