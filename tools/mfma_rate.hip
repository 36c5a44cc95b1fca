// Matrix-instruction issue rate on this GPU (v_mfma_f32_32x32x16_bf16), alone and with the streamed block's fragment
// reads beside it: how many SIMD cycles per instruction at 1 and 2 waves per SIMD; also s_memtime ticks per second.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/mfma_rate tools/mfma_rate.hip && tools/bin/mfma_rate
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>  // 0: one accumulator chain; 1: seven accumulators round robin; 2: chain + two 16-byte LDS reads per three
__global__ __launch_bounds__(512, 2) void rate_kernel(int iters, float *out, unsigned long long *ticks) {
  extern __shared__ uint4 lds[];
  const int lane = threadIdx.x & 63;
  for (int q = threadIdx.x; q < 28672 / 16; q += blockDim.x) lds[q] = make_uint4(q, q + 1, q + 2, q + 3);
  __syncthreads();
  f32x16 acc[7];
  for (int u = 0; u < 7; ++u)
    for (int r = 0; r < 16; ++r) acc[u][r] = 0.0f;
  uint4 a = lds[lane], b = lds[64 + lane];
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 14; ++k) {
      if (MODE == 2) {
        a = lds[(k * 2) * 64 + lane];
        b = lds[(k * 2 + 1) * 64 + lane];
      }
      const bf16x8 wa = *reinterpret_cast<const bf16x8 *>(&a), wb = *reinterpret_cast<const bf16x8 *>(&b);
      const int u = MODE == 1 ? k % 7 : 0;
      acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa, wb, acc[u], 0, 0, 0);
      acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wb, wa, acc[u], 0, 0, 0);
      acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa, wa, acc[u], 0, 0, 0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.0f;
  for (int u = 0; u < 7; ++u)
    for (int r = 0; r < 16; ++r) s += acc[u][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}

template <int MODE>
void run(const char *name, int waves_per_block, int blocks_per_cu, int cus) {
  const int iters = 200, mfma_per_wave = iters * 14 * 3;
  float *out;
  unsigned long long *ticks, h = 0;
  hipMalloc(&out, (size_t)cus * blocks_per_cu * waves_per_block * 64 * 4);
  hipMalloc(&ticks, 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const size_t shm = blocks_per_cu == 1 ? 100 * 1024 : 32 * 1024;  // (forces the occupancy asked for)
  hipFuncSetAttribute((const void *)rate_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(rate_kernel<MODE>, dim3(cus * blocks_per_cu), dim3(waves_per_block * 64), shm, 0, iters, out, ticks);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
  }
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  hipMemcpy(&h, ticks, 8, hipMemcpyDeviceToHost);
  const double waves_per_simd = waves_per_block * blocks_per_cu / 4.0;
  const double per_simd = mfma_per_wave * waves_per_simd;
  printf("%-46s %d waves/SIMD: %8.1f us, %6.1f ns per instruction and SIMD (= %5.1f cycles at 2.4 GHz); wave 0: %llu ticks = %.2f ticks/ns\n", name,
         (int)waves_per_simd, ms * 1e3, ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4, h, h / (ms * 1e6));
  hipFree(out);
  hipFree(ticks);
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  printf("%s, %d CUs, clock %d MHz\n", p.name, cus, p.clockRate / 1000);
  run<0>("one accumulator chain", 4, 1, cus);
  run<0>("one accumulator chain", 8, 1, cus);
  run<1>("seven accumulators round robin", 4, 1, cus);
  run<1>("seven accumulators round robin", 8, 1, cus);
  run<2>("chain + 2 x 16-byte LDS reads per 3 instructions", 4, 1, cus);
  run<2>("chain + 2 x 16-byte LDS reads per 3 instructions", 8, 1, cus);
  return 0;
}
