// Micro-benchmark behind DESIGN §3 (peer-store halo): what does a pack wave's burst of 16-byte flag-in-data stores cost, by cache
// policy of the store (aux bits: 1 sc0, 2 nt, 16 sc1), kind of destination memory (fine-grained window as p2p.hip allocates it, or
// ordinary device memory) and number of work-groups sharing the same total (4.7 MB = the six fp64 faces of 32 x 16 x 16 x 16)?
// Per variant: kernel time from HIP events, and from wall_clock64 inside the kernel the time a wave needs to ISSUE its stores and
// the time until they are all acknowledged (s_waitcnt vmcnt(0)).
//   hipcc --offload-arch=gfx950 -O3 -o ubench_ll_store tools/ubench_ll_store.hip && ./ubench_ll_store
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int AUX, int NV>
__global__ void __launch_bounds__(256) store_kernel(char *dst, int nsites, unsigned flag, unsigned long long *stamps) {
  // grid <= nsites / 256: a thread stores `per` sites one after the other; grid > nsites / 256: a site's NV planes are split over
  // `split` threads.  Site f of plane v at (v * nsites + f) * 16, as ghost_ll_store
  const int full = nsites / 256;
  const int split = (int)gridDim.x > full ? (int)gridDim.x / full : 1;
  const int per = (int)gridDim.x < full ? full / (int)gridDim.x : 1;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(dst, 0, nsites * NV * 16, 0x00020000);
  const unsigned long long t0 = wall_clock64();
  for (int r = 0; r < per; r++) {
    const int item = (blockIdx.x * per + r) * 256 + threadIdx.x;
    const int f = item % nsites, part = item / nsites;
    if (part < split) {
#pragma unroll
      for (int v0 = 0; v0 < NV; v0++) {
        if (v0 < NV / split) {
          const int v = part * (NV / split) + v0;
          u32x4 q; q.x = f + v; q.y = flag; q.z = f - v; q.w = flag;
          __builtin_amdgcn_raw_buffer_store_b128(q, rs, f * 16 + v * nsites * 16, 0, AUX);
        }
      }
    }
  }
  const unsigned long long t1 = wall_clock64();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t2 = wall_clock64();
  if (stamps && (threadIdx.x & 63) == 0) {
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    stamps[2 * w] = t1 - t0; stamps[2 * w + 1] = t2 - t0;
  }
}

__global__ void empty_kernel() {}

template <int AUX> static void run(const char *memname, char *dst, int nsites, int blocks, unsigned long long *stamps, unsigned long long *hstamps) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  constexpr int NV = 12;
  for (int i = 0; i < 20; i++) hipLaunchKernelGGL((store_kernel<AUX, NV>), dim3(blocks), dim3(256), 0, 0, dst, nsites, 1u + i, nullptr);
  CK(hipDeviceSynchronize());
  const int reps = 200;
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < reps; i++) hipLaunchKernelGGL((store_kernel<AUX, NV>), dim3(blocks), dim3(256), 0, 0, dst, nsites, 100u + i, nullptr);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  hipLaunchKernelGGL((store_kernel<AUX, NV>), dim3(blocks), dim3(256), 0, 0, dst, nsites, 7u, stamps);
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(hstamps, stamps, blocks * 4 * 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  double si = 0, sa = 0, mi = 0, ma = 0;
  for (int w = 0; w < blocks * 4; w++) { const double a = hstamps[2 * w] * 0.01, b = hstamps[2 * w + 1] * 0.01; si += a; sa += b; mi = std::max(mi, a); ma = std::max(ma, b); }
  printf("%-12s aux %2d  blocks %4d: kernel %6.2f us   issue mean %5.2f max %5.2f us   acknowledged mean %5.2f max %5.2f us   (%.2f TB/s)\n", memname, AUX, blocks,
         1e3 * ms / reps, si / (blocks * 4), mi, sa / (blocks * 4), ma, (double)nsites * NV * 16 / (1e-3 * ms / reps) * 1e-12);
  CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
}

int main() {
  const int nsites = 24576;   // six faces of 4096 sites
  const size_t bytes = (size_t)nsites * 12 * 16;
  char *fine, *coarse;
  CK(hipExtMallocWithFlags((void **)&fine, bytes, hipDeviceMallocFinegrained));
  CK(hipMalloc((void **)&coarse, bytes));
  unsigned long long *stamps, *hstamps = (unsigned long long *)malloc(1024 * 4 * 2 * sizeof(unsigned long long));
  CK(hipMalloc((void **)&stamps, 1024 * 4 * 2 * sizeof(unsigned long long)));
  {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 20; i++) hipLaunchKernelGGL(empty_kernel, dim3(96), dim3(256), 0, 0);
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < 200; i++) hipLaunchKernelGGL(empty_kernel, dim3(96), dim3(256), 0, 0);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("empty kernel: %.2f us per launch\n", 1e3 * ms / 200);
  }
  for (int pass = 0; pass < 2; pass++) {
    char *dst = pass ? coarse : fine;
    const char *nm = pass ? "device" : "fine-grained";
    for (int blocks : {24, 48, 96, 192, 384}) {
      if (blocks * 256 > nsites && blocks != 96) { }
      run<0>(nm, dst, nsites, blocks, stamps, hstamps);
      run<2>(nm, dst, nsites, blocks, stamps, hstamps);
      run<16>(nm, dst, nsites, blocks, stamps, hstamps);
      run<17>(nm, dst, nsites, blocks, stamps, hstamps);
      run<19>(nm, dst, nsites, blocks, stamps, hstamps);
    }
  }
  return 0;
}
