// Device-wide barrier variants for the persistent coarse-cycle kernel (csrc/coarse_cycle.hip), timed in isolation:
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_grid_barrier.hip -o /tmp/ubench_grid_barrier && /tmp/ubench_grid_barrier
// A  one counter, release / acquire fences at agent scope around it          B  the same without the fences (s_waitcnt only)
// C  arrival slots + a master work-group that gathers them and raises a release word (fences)     D  the same without fences
// every variant with a short and a long s_sleep in the polling loop; grid sizes 64 ... 512 work-groups of 256 threads, 2000 barriers.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int VAR, int SLEEP> __global__ void __launch_bounds__(256) bar_kernel(unsigned *ctr, unsigned *slots, unsigned *release, float *data, int iters) {
  unsigned epoch = 0;
  for (int it = 0; it < iters; it++) {
    // a little work whose result the next phase of another work-group reads (so the fences have something to do)
    data[(size_t)blockIdx.x * 256 + threadIdx.x] = data[(size_t)((blockIdx.x + 1) % gridDim.x) * 256 + threadIdx.x] + 1.0f;
    __syncthreads();
    epoch++;
    if (VAR == 0 || VAR == 1) {
      if (threadIdx.x == 0) {
        if (VAR == 0) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned target = epoch * gridDim.x;
        while ((int)(__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) __builtin_amdgcn_s_sleep(SLEEP);
        if (VAR == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      }
    } else {
      if (threadIdx.x == 0) {
        if (VAR == 2) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(slots + blockIdx.x * 16, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // one 64-byte line per work-group
      }
      if (blockIdx.x == 0) {
        for (int w = threadIdx.x; w < (int)gridDim.x; w += 256)
          while ((int)(__hip_atomic_load(slots + w * 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - epoch) < 0) __builtin_amdgcn_s_sleep(SLEEP);
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(release, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else if (threadIdx.x == 0) {
        while ((int)(__hip_atomic_load(release, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - epoch) < 0) __builtin_amdgcn_s_sleep(SLEEP);
      }
      if (threadIdx.x == 0 && VAR == 2) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
  }
}

template <int VAR, int SLEEP> static void run(const char *name, int grid, unsigned *ctr, unsigned *slots, unsigned *release, float *data) {
  const int iters = 2000;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 2; rep++) {
    CK(hipMemset(ctr, 0, 64)); CK(hipMemset(slots, 0, 1024 * 64)); CK(hipMemset(release, 0, 64));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((bar_kernel<VAR, SLEEP>), dim3(grid), dim3(256), 0, 0, ctr, slots, release, data, iters);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep) printf("%-44s sleep %2d  grid %4d  %7.2f us per barrier\n", name, SLEEP, grid, 1e3 * ms / iters);
  }
}

int main() {
  unsigned *ctr, *slots, *release; float *data;
  CK(hipMalloc(&ctr, 64)); CK(hipMalloc(&slots, 1024 * 64)); CK(hipMalloc(&release, 64)); CK(hipMalloc(&data, 1024 * 256 * 4));
  CK(hipMemset(data, 0, 1024 * 256 * 4));
  for (int grid : {64, 128, 256, 512}) {
    run<0, 1>("A counter + agent fences", grid, ctr, slots, release, data);
    run<0, 8>("A counter + agent fences", grid, ctr, slots, release, data);
    run<1, 1>("B counter, no fences", grid, ctr, slots, release, data);
    run<1, 8>("B counter, no fences", grid, ctr, slots, release, data);
    run<2, 1>("C slots + master + release word, fences", grid, ctr, slots, release, data);
    run<2, 8>("C slots + master + release word, fences", grid, ctr, slots, release, data);
    run<3, 1>("D slots + master + release word, no fences", grid, ctr, slots, release, data);
    run<3, 8>("D slots + master + release word, no fences", grid, ctr, slots, release, data);
  }
  return 0;
}
