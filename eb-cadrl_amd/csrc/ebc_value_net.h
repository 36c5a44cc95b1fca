// ebc_value_net.h — dense layers of the SARL value network (rl/policy/sarl.py:9-82) on the bf16 matrix
// cores at f32 accuracy.
//
// Why not f32 MFMA: v_mfma_f32_32x32x2_f32 runs at 1/16 of the bf16 rate.  Every f32 operand is split
// instead into two bf16 numbers, x = hi + lo (hi = bf16(x), lo = bf16(x - hi)), and a product keeps
// its three leading terms, hi*hi + hi*lo + lo*hi, accumulated in f32 by the MFMA: 16/3 the f32
// matrix rate.  On the reference's decision runs the values move by at most 1.5e-5 (bar: 2e-4;
// tools/split_bf16_accuracy.py); plain bf16 moves them by 7e-3 and changes half the decisions.
//
// Layout: layers are computed TRANSPOSED, H^T = W * X^T: the 32 samples of a tile sit on the lanes
// (column = lane & 31), units on the rows of the 32x32 accumulator tile (row = (reg & 3) + 8 (reg >> 2)
// + 4 (lane >> 5)).  A finished tile is then the next layer's B operand as it stands — registers
// 8s .. 8s+7 are the fragment of k-step s, fragment element j being row 16s + 8 (j >> 2) + 4h + (j & 3)
// (cdna_hip_programming.md, "An accumulator tile as the next MFMA's operand") — so activations never
// leave the registers between layers; the host permutes the weights' columns to that k order when it
// packs them into A fragments (pack_layer in ebcsim.hip).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ebc {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct Frag2 {  // one operand fragment, split
  bf16x8 hi, lo;
};

__device__ __forceinline__ Frag2 split8(const float (&v)[8]) {
  Frag2 f;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    f.hi[j] = (__bf16)v[j];
    f.lo[j] = (__bf16)(v[j] - (float)f.hi[j]);
  }
  return f;
}

// acc += W * X for one k-step: the three leading terms of (Wh + Wl)(Xh + Xl), small ones first
__device__ __forceinline__ f32x16 mfma_split(const Frag2 &w, const Frag2 &x, f32x16 acc) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.lo, x.hi, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.hi, x.lo, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.hi, x.hi, acc, 0, 0, 0);
  return acc;
}

// Packed layer: A fragments [out tile][in tile][k-step 0/1][hi, lo][lane][8 bf16] and the bias in
// accumulator order [out tile][lane half][16].
struct PackedLayer {
  const uint4 *frag;
  const float *bias;
  int in_tiles, out_tiles;
};

__device__ __forceinline__ Frag2 load_weight(const PackedLayer &L, int t, int u, int s, int lane) {
  const uint4 *p = L.frag + ((((size_t)t * L.in_tiles + u) * 2 + s) * 2) * 64 + lane;
  const uint4 h = p[0], l = p[64];
  Frag2 f;
  f.hi = *reinterpret_cast<const bf16x8 *>(&h);
  f.lo = *reinterpret_cast<const bf16x8 *>(&l);
  return f;
}

__device__ __forceinline__ f32x16 bias_tile(const PackedLayer &L, int t, int lane) {
  const float4 *b = reinterpret_cast<const float4 *>(L.bias + ((size_t)t * 2 + (lane >> 5)) * 16);
  f32x16 acc;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float4 v = b[q];
    acc[4 * q] = v.x; acc[4 * q + 1] = v.y; acc[4 * q + 2] = v.z; acc[4 * q + 3] = v.w;
  }
  return acc;
}

// the two B fragments (k-steps) a finished 32x32 tile offers to the next layer
__device__ __forceinline__ void tile_frags(const f32x16 &acc, bool relu, Frag2 (&out)[2]) {
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = relu ? fmaxf(acc[8 * s + j], 0.0f) : acc[8 * s + j];
    out[s] = split8(v);
  }
}

// Y = act(W2 * relu(W1 * X + b1) + b2), one wave per tile of 32 rows of X [M][K0] (K0 <= 32 TI).
// TI / TO: input / output tiles (compile time: they index registers); the hidden tiles stream.
template <int TI, int TO>
__global__ __launch_bounds__(64) void mlp2_split_kernel(const float *X, int M, int K0, PackedLayer L1, PackedLayer L2,
                                                       int relu_out, float *Y, int O) {
  const int lane = threadIdx.x, col = lane & 31, half = lane >> 5;
  const int m = blockIdx.x * 32 + col;
  // the input tile as B fragments, natural k order: element j of k-step s is k = 16 s + 8 half + j
  Frag2 x[TI][2];
#pragma unroll
  for (int u = 0; u < TI; ++u)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k = u * 32 + 16 * s + 8 * half + j;
        v[j] = (m < M && k < K0) ? X[(size_t)m * K0 + k] : 0.0f;
      }
      x[u][s] = split8(v);
    }
  f32x16 out[TO];
#pragma unroll
  for (int t = 0; t < TO; ++t) out[t] = bias_tile(L2, t, lane);
  for (int u = 0; u < L1.out_tiles; ++u) {  // hidden tile u: made, turned into fragments, consumed
    f32x16 hid = bias_tile(L1, u, lane);
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int s = 0; s < 2; ++s) hid = mfma_split(load_weight(L1, u, i, s, lane), x[i][s], hid);
    Frag2 hf[2];
    tile_frags(hid, true, hf);
#pragma unroll
    for (int t = 0; t < TO; ++t)
#pragma unroll
      for (int s = 0; s < 2; ++s) out[t] = mfma_split(load_weight(L2, t, u, s, lane), hf[s], out[t]);
  }
  if (m < M) {
#pragma unroll
    for (int t = 0; t < TO; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int unit = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (unit < O) Y[(size_t)m * O + unit] = relu_out ? fmaxf(out[t][r], 0.0f) : out[t][r];
      }
  }
}

}  // namespace ebc
