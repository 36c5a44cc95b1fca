// ebc_vn_common.h — types and fragment helpers shared by the value-network kernels (ebc_value_net.h: the general
// two-layer block; ebc_vn_stream.h: the streamed attention block).  Conventions: the header of ebc_value_net.h.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ebc {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float vn_f32x4 __attribute__((ext_vector_type(4)));
typedef vn_f32x4 __attribute__((address_space(3))) *LdsF4;  // explicit LDS pointers: no flat-address casts in the loops
#define EBC_VN_XROW 144  // bytes per row of a wave's input transposition tile: 128 + 16 (conflict-free 16-byte reads)
#define EBC_VN_GROUPS 4         // group-term rows a wave parks in LDS (its 32 rows span at most that many groups)
#define EBC_VN_GROUP_PITCH 912  // bytes per parked row: 224 floats + 16
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

// Two things the attention stack of the value network needs beyond a plain two-layer block.
//   row_bias [M / group_rows][H]: added to the hidden pre-activation of every row of a group (the
//     pair's mean-state term of attention layer 0, cat([h1, g]) without the concatenation);
//   final_w [O], final_b: a third layer with one output, y [M] = final_w . relu(out) + final_b,
//     worked out from the accumulators instead of storing out [M][O].
//   partial [ceil(M / 32)][3][O]: per 32-row tile, the weighted sums of its rows over each of the (at most three,
//     seg_rows >= 16) consecutive row groups of seg_rows rows it touches — sum_r row_weight[r] * y[r][:] (row_weight
//     NULL: 1) — so that a per-group mean (sarl.py:56-58) or attention-weighted sum (sarl.py:73-76) is the sum of
//     two or three of these (pair_combine_kernel) instead of a second pass over y; store_y == 0: y itself is not
//     written at all.  The sums are kept in float64 (products of two float32 are exact there, 18 of them add up
//     with 2^-53 relative error): whatever rows of a pair share a tile, whichever lanes hold them, the float32 the
//     combine step rounds to is the same — identical pairs at different places of the batch get identical results,
//     as with a serial row loop.
struct MlpExtra {
  const float *row_bias;
  int group_rows, H;
  const float *final_w;
  float final_b;
  const float *row_weight;
  double *partial;
  int seg_rows, store_y;
  // Round 3: activations handed from block to block ALREADY SPLIT and in fragment order.  frag_out: the output tiles
  // leave as the B fragments the next block's first layer multiplies — tile_frags() of the finished accumulators,
  // [row tile][output tile][k-step][hi, lo][lane] 16-byte pieces, the same 4 bytes per element as float32 rows —;
  // frag_in: the input IS such a tensor (X is not read): no row loads through an LDS transposition tile and no
  // per-lane split8 in the consumer's input phase.  A consumer of fragments has its first layer packed in accumulator
  // k order (ebc_mlp2_create_ex, EBC_MLP_IN_FRAGMENTS).
  const uint4 *frag_in;
  uint4 *frag_out;
};
#define EBC_VN_SEGS 3

// Packed layer: A fragments [out tile][in tile][k-step 0/1][hi, lo][lane][8 bf16] and the bias in
// accumulator order [out tile][lane half][16].
struct PackedLayer {
  const uint4 *frag;
  const float *bias;
  int in_tiles, out_tiles;
};

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

// max(v, 0) for a finite v as ONE integer instruction on the bits (a negative float is a negative integer; -0 -> +0):
// fmaxf costs two, the first a canonicalisation of its operand that IEEE mode demands for signalling NaNs.  (A NaN with
// the sign bit clear stays a NaN where fmaxf gives 0: the blocks' values are finite or the result is garbage anyway.)
__device__ __forceinline__ float relu_bits(float v) {
  const int b = __float_as_int(v);
  return __int_as_float(b > 0 ? b : 0);
}

// the two B fragments (k-steps) a finished 32x32 tile offers to the next layer
__device__ __forceinline__ void tile_frags(const f32x16 &acc, bool relu, Frag2 (&out)[2]) {
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = relu ? relu_bits(acc[8 * s + j]) : acc[8 * s + j];
    out[s] = split8(v);
  }
}

template <int I>
struct IntC {  // a compile-time index as an argument (macro-free unrolling with constant register indices)
  static constexpr int value = I;
};

}  // namespace ebc
