// launch_rate.hip — how fast does gfx950 start waves?  Times kernels that do almost nothing per
// wave, over grid size, workgroup size, LDS allocation and the number of kernarg dwords each wave
// reads, plus a fixed amount of dependent VALU work.  Build: hipcc -O3 --offload-arch=gfx950
// tools/launch_rate.hip -o tools/bin/launch_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

struct Args { float v[64]; float *out; };

template <int LDS, int NARG, int WORK>
__global__ void probe(Args a) {
  extern __shared__ float dyn[];
  __shared__ float lds[LDS > 0 ? LDS / 4 : 1];
  float x = (float)threadIdx.x;
#pragma unroll
  for (int i = 0; i < NARG; ++i) x += a.v[i];
#pragma unroll 1
  for (int i = 0; i < WORK; ++i) x = x * 1.0001f + 0.5f;
  if (LDS > 0) { lds[threadIdx.x % (LDS / 4)] = x; x += lds[(threadIdx.x + 1) % (LDS / 4)]; }
  if (x == 12345.678f) a.out[0] = x;  // never
}

template <int LDS, int NARG, int WORK>
static void run(const char *name, int wg, hipStream_t st, float *out) {
  Args a{}; a.out = out;
  for (int i = 0; i < 64; ++i) a.v[i] = 0.001f * i;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int waves : {256, 1024, 2048, 4096, 8192, 16384, 65536}) {
    const int blocks = waves * 64 / wg;
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((probe<LDS, NARG, WORK>), dim3(blocks), dim3(wg), 0, st, a);
    hipStreamSynchronize(st);
    const int reps = 200;
    hipEventRecord(e0, st);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((probe<LDS, NARG, WORK>), dim3(blocks), dim3(wg), 0, st, a);
    hipEventRecord(e1, st);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s wg %4d waves %6d  %7.2f us/launch  %6.3f ns/wave\n", name, wg, waves, ms * 1e3 / reps, ms * 1e6 / reps / waves);
  }
}

int main() {
  hipStream_t st; hipStreamCreate(&st);
  float *out; hipMalloc(&out, 4);
  run<0, 0, 0>("empty", 64, st, out);
  run<0, 0, 0>("empty", 256, st, out);
  run<0, 0, 0>("empty", 1024, st, out);
  run<3072, 0, 0>("lds3k", 64, st, out);
  run<0, 16, 0>("16 kernarg dwords", 64, st, out);
  run<0, 64, 0>("64 kernarg dwords", 64, st, out);
  run<0, 0, 500>("500 dependent fma", 64, st, out);
  run<0, 0, 500>("500 dependent fma", 256, st, out);
  run<3072, 16, 500>("lds+16args+500fma", 64, st, out);
  run<3072, 16, 500>("lds+16args+500fma", 256, st, out);
  return 0;
}
