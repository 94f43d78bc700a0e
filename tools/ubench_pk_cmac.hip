// Semantics check of the packed complex multiply-add used by the fp32 / 16-bit stencil (dslash.hip su3_mv_pk):
//   acc += u * h   (complex, (re, im) in an aligned register pair)  =  two v_pk_fma_f32 with op_sel / neg_lo
//   hipcc --offload-arch=gfx950 -O3 -o ubench_pk_cmac tools/ubench_pk_cmac.hip && ./ubench_pk_cmac
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 cmac(f2 acc, f2 u, f2 h) {
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(u), "v"(h));
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "+v"(acc) : "v"(u), "v"(h));
  return acc;
}
// acc += conj(u) * h
__device__ __forceinline__ f2 cmac_conj(f2 acc, f2 u, f2 h) {
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(u), "v"(h));
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_hi:[1,0,0]" : "+v"(acc) : "v"(u), "v"(h));
  return acc;
}
__global__ void k(const float *in, float *out) {
  const int t = threadIdx.x;
  f2 a = {in[6 * t], in[6 * t + 1]}, u = {in[6 * t + 2], in[6 * t + 3]}, h = {in[6 * t + 4], in[6 * t + 5]};
  const f2 r = cmac(a, u, h), c = cmac_conj(a, u, h);
  out[4 * t] = r.x; out[4 * t + 1] = r.y; out[4 * t + 2] = c.x; out[4 * t + 3] = c.y;
}
int main() {
  float h_in[64 * 6], h_out[64 * 4], *d_in, *d_out;
  for (int i = 0; i < 64 * 6; i++) h_in[i] = sinf(0.37f * i) * 3.f;
  hipMalloc(&d_in, sizeof(h_in)); hipMalloc(&d_out, sizeof(h_out));
  hipMemcpy(d_in, h_in, sizeof(h_in), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_in, d_out);
  hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
  double worst = 0;
  for (int t = 0; t < 64; t++) {
    const float *p = h_in + 6 * t;
    const double re = p[0] + (double)p[2] * p[4] - (double)p[3] * p[5], im = p[1] + (double)p[2] * p[5] + (double)p[3] * p[4];
    const double cre = p[0] + (double)p[2] * p[4] + (double)p[3] * p[5], cim = p[1] + (double)p[2] * p[5] - (double)p[3] * p[4];
    worst = fmax(worst, fmax(fmax(fabs(h_out[4 * t] - re), fabs(h_out[4 * t + 1] - im)), fmax(fabs(h_out[4 * t + 2] - cre), fabs(h_out[4 * t + 3] - cim))));
  }
  printf("packed complex multiply-add: worst deviation %.3e (%s)\n", worst, worst < 1e-5 ? "ok" : "WRONG");
  return worst < 1e-5 ? 0 : 1;
}
