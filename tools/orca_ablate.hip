// Diagnostic: where does the ORCA group kernel spend its time?  Ablation by MODE on synthetic
// crossing-like tiles (not part of the product; built and run by hand on the GPU box).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I eb-cadrl_amd/csrc tools/orca_ablate.hip -o /tmp/orca_ablate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include "ebc_orca_group.h"

struct Tile { float *px, *py, *vx, *vy, *rad, *mx, *prx, *pry; float *out; int E, N; };

template <int GS, int MODE>
__global__ __launch_bounds__(64) void k(EbcParams p, Tile t) {
  constexpr int HPW = 64 / GS;
  __shared__ __align__(16) float dist_lds[64];
  __shared__ float4 lines_lds[64];
  __shared__ float4 proj_lds[64];
  const int lane = threadIdx.x, group = lane / GS, j = lane - group * GS;
  const long h = (long)blockIdx.x * HPW + group;
  const int N = t.N;
  const bool h_ok = h < (long)t.E * N;
  const int e = h_ok ? (int)(h / N) : 0, i = h_ok ? (int)(h - (long)e * N) : 0;
  const size_t base = (size_t)e * N, ks = base + i;
  const int oj = j < i ? j : j + 1;
  const size_t ko = base + (oj < N ? oj : N - 1);
  float posx = t.px[ks], posy = t.py[ks], velx = t.vx[ks], vely = t.vy[ks], radius = t.rad[ks];
  float maxSpeed = t.mx[ks], prefx = t.prx[ks], prefy = t.pry[ks];
  float opx = t.px[ko], opy = t.py[ko], ovx = t.vx[ko], ovy = t.vy[ko], orad = t.rad[ko];
  const bool valid = h_ok && j < N - 1;
  float ox = prefx, oy = prefy;
  if (MODE == 0) {
    ebc::orca_group<GS>(p, j, group, valid, posx, posy, velx, vely, radius, maxSpeed, prefx, prefy, opx, opy,
                        ovx, ovy, orad, dist_lds + group * GS, lines_lds + group * GS, proj_lds + group * GS, N - 1, ox, oy);
  } else if (MODE == 3) {
    ox = posx + opx + velx + vely + radius + maxSpeed + opy + ovx + ovy + orad;
  }
  if (h_ok && j == 0) { t.out[h * 2] = ox; t.out[h * 2 + 1] = oy; }
}

template <int GS, int MODE>
float run(EbcParams p, Tile t, int blocks, int iters) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int q = 0; q < 20; ++q) hipLaunchKernelGGL((k<GS, MODE>), dim3(blocks), dim3(64), 0, 0, p, t);
  hipEventRecord(a, 0);
  for (int q = 0; q < iters; ++q) hipLaunchKernelGGL((k<GS, MODE>), dim3(blocks), dim3(64), 0, 0, p, t);
  hipEventRecord(b, 0); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms / iters * 1e3f;
}

int main(int argc, char **argv) {
  int E = argc > 1 ? atoi(argv[1]) : 4096, N = 10;
  EbcParams p{}; p.time_step = 0.25; p.orca_neighbor_dist = 10; p.orca_time_horizon = 5; p.orca_max_neighbors = 10;
  size_t n = (size_t)E * N;
  std::vector<float> h[8];
  srand(1);
  auto rnd = [] { return rand() / (float)RAND_MAX; };
  for (auto &v : h) v.resize(n);
  for (size_t q = 0; q < n; ++q) {
    float dense = (q / N) % 2 ? 3.0f : 9.0f;  // half the scenes crowded
    h[0][q] = (rnd() - 0.5f) * dense; h[1][q] = (rnd() - 0.5f) * dense;
    h[2][q] = (rnd() - 0.5f); h[3][q] = (rnd() - 0.5f);
    h[4][q] = 0.21f + 0.3f * rnd(); h[5][q] = 0.4f + 0.6f * rnd();
    float a = rnd() * 6.28f; h[6][q] = cosf(a); h[7][q] = sinf(a);
  }
  Tile t; t.E = E; t.N = N;
  float **dst[8] = {&t.px, &t.py, &t.vx, &t.vy, &t.rad, &t.mx, &t.prx, &t.pry};
  for (int c = 0; c < 8; ++c) { hipMalloc(dst[c], n * 4); hipMemcpy(*dst[c], h[c].data(), n * 4, hipMemcpyHostToDevice); }
  hipMalloc(&t.out, n * 8);
  int blocks16 = (int)((n + 3) / 4);
  printf("E %d N %d waves %d\n", E, N, blocks16);
  printf("GS16 full   %8.2f us\n", run<16, 0>(p, t, blocks16, 200));
  printf("GS16 loads  %8.2f us\n", run<16, 3>(p, t, blocks16, 200));
  printf("GS16 empty  %8.2f us\n", run<16, 1>(p, t, blocks16, 200));
  return 0;
}
