// any_order.hip — does a kernel launched with hipExtAnyOrderLaunch start before the previous kernel
// of the SAME stream has finished on gfx950?  Kernel A spins ~20 us; kernel B records its start.
// Build: hipcc -O3 --offload-arch=gfx950 tools/any_order.hip -o tools/bin/any_order
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>

__global__ void spin(unsigned long long *t, unsigned long long ticks) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(4);
  if (threadIdx.x == 0 && blockIdx.x == 0) { t[0] = t0; t[1] = __builtin_amdgcn_s_memrealtime(); }
}
__global__ void stamp(unsigned long long *t) {
  if (threadIdx.x == 0 && blockIdx.x == 0) t[2] = __builtin_amdgcn_s_memrealtime();
}

int main() {
  hipStream_t st; (void)hipStreamCreate(&st);
  unsigned long long *t; (void)hipMalloc(&t, 64);
  unsigned long long h[3];
  for (int flags : {0, (int)hipExtAnyOrderLaunch}) {
    for (int blocks : {64, 1024, 8192}) {
      for (int rep = 0; rep < 3; ++rep) {
        hipExtLaunchKernelGGL(spin, dim3(blocks), dim3(64), 0, st, nullptr, nullptr, 0, t, 2000ull);
        hipExtLaunchKernelGGL(stamp, dim3(64), dim3(64), 0, st, nullptr, nullptr, flags, t);
        (void)hipStreamSynchronize(st);
      }
      (void)hipMemcpy(h, t, 24, hipMemcpyDeviceToHost);
      printf("flags %d, A = %5d blocks: A ran %.2f us; B started %.2f us after A started (%s A ended)\n", flags, blocks,
             (h[1] - h[0]) / 100.0, ((long long)h[2] - (long long)h[0]) / 100.0, h[2] < h[1] ? "BEFORE" : "after");
    }
  }
  return 0;
}
