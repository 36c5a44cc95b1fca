// ebc_vn_stream.h — the wide blocks of the SARL value network (/root/reference/rl/policy/sarl.py:38-82: `mlp1` on the
// observation rows, the attention stack mlp(cat([h1, g]), [.., .., 1]), `mlp2`), streamed.
//
// The general block (mlp2_split_wg_kernel, ebc_value_net.h) keeps a wave's INPUT fragments in registers across all
// hidden tiles.  At 7 input + 7 output tiles (the attention stack) that is 104 + 112 registers before anything else: the
// compiler spilled 21 of them, and every reload inside the loop is an `s_waitcnt vmcnt(0)` that also waits for the
// weight staging in flight behind it (one in-order counter) — the matrix pipe sat at 26 % busy.
//
// Here the loops are turned inside out so that nothing wide stays resident:
//   phase A, one PERIOD per input tile i: hacc[u] += W1[u][i] . x[i] for ALL hidden tiles u (TH accumulators; the
//     input fragments of one tile are loaded, used for 3 TH instructions and dropped);
//   phase B: bias (in the accumulators from the start) [+ the pair's group term: GROUP], ReLU, split: the TH hidden
//     tiles become the 2 TH fragments of the second layer, in place;
//   phase C, one period per output tile t: out = W2[t][.] . hid over all hidden tiles (ONE accumulator, not TO of them),
//     and the tile is finished on the spot: FINAL — the one-output third layer's share of it (the attention scores);
//     else the general block's tile epilogue — the tile as the next block's fragments (MlpExtra.frag_out) and / or its
//     weighted pair sums (MlpExtra.partial), its rows never written (`mlp1`, `mlp2`).
// Every period multiplies one SLAB of weights (TH fragments x 2 k-steps x hi/lo: 28 KB at 7 hidden tiles), the same
// for all waves of the workgroup: slabs arrive by LDS-DMA two periods ahead into a ring of three; input fragments come
// from the hand-off tensor (MlpExtra.frag_in: already split, already in fragment order) by plain 16-byte loads hidden in
// asm, two periods ahead into a ring of three register sets — or, ROWS, from float32 rows split by the lanes in the
// head (one input tile).  ONE barrier per period, in front of it ONE counted wait that leaves the newest period's loads
// in flight.  With GROUP and FINAL the products and sums per output are those of the general block in the same order:
// bit-equal scores; the other two forms add their hidden tiles up in one chain where the general block alternates two
// accumulators: equal to rounding (tests/test_value_net.py).  Measurements: DESIGN.md section 3 item 18.
#pragma once

#include "ebc_vn_common.h"

namespace ebc {

template <int... I>
struct IntSeq {};
template <int N, int... I>
struct MakeIntSeq : MakeIntSeq<N - 1, N - 1, I...> {};
template <int... I>
struct MakeIntSeq<0, I...> {
  using type = IntSeq<I...>;
};
template <class F, int... I>
__device__ __forceinline__ void for_each_int(F &&f, IntSeq<I...>) {
  (f(IntC<I>{}), ...);
}

typedef unsigned vn_u32x4 __attribute__((ext_vector_type(4)));

template <int N>
__device__ __forceinline__ void vm_wait() {  // at most N vector-memory operations of this wave still in flight
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// TI / TH / TO: input / hidden / output tiles; KIN / KH: the input (hidden) width ends in the first half of its last
// tile: that tile's second k-step multiplies zeros and is left out.
// GROUP: the pair's group term is added to the hidden pre-activation (the attention stack).  FINAL: a one-output third
// layer finishes every output tile on the spot (y [M]); else the tile goes through the pair-sum epilogue of the general
// block (MlpExtra.partial / row_weight / seg_rows: `mlp2`, whose rows are never written), one tile per period.
// ROWS: the input is float32 rows X [M][K0] (one input tile: `mlp1` on the 17-float observation rows), split by the lanes
// in the head, natural k order; else the fragment tensor MlpExtra.frag_in.  Not FINAL: the tile can also leave as the
// next block's fragments (MlpExtra.frag_out).
template <int TI, int TH, int TO, int NW, int KIN, int KH, bool GROUP, bool FINAL, bool ROWS = false>
__global__ __launch_bounds__(64 * NW, 2) void mlp2_stream_kernel(int M, PackedLayer L1, PackedLayer L2, float *Y, int O, MlpExtra ex, int relu_out,
                                                                 const float *X, int K0) {
  static_assert(!ROWS || (TI == 1 && !KIN), "row input: one input tile");
  extern __shared__ uint4 sbuf[];  // slab ring [3][TH][2][2][64] | hidden bias | output bias | third layer | group terms | output tiles
  constexpr int PART = 64, SLAB = TH * 4 * PART, PIECES = TH * 4;
  constexpr int NST = (PIECES + NW - 1) / NW;  // staging instructions per wave and slab (the last waves repeat the last piece)
  constexpr int P = TI + TO;                   // periods
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, col = lane & 31, half = lane >> 5;
  const int tile = blockIdx.x * NW + wave, m0w = tile * 32, m = m0w + col;
  float *hbias = reinterpret_cast<float *>(sbuf + 3 * SLAB), *obias = hbias + TH * 32, *fwl = obias + TO * 32;
  const LdsF4 gt = (LdsF4)(reinterpret_cast<unsigned char *>(fwl + TO * 32) + (size_t)wave * EBC_VN_GROUPS * EBC_VN_GROUP_PITCH);
  // the wave's parked output tile of the pair-sum epilogue: 32 rows x 144 bytes (128 of the tile + the row's weight pair)
  constexpr size_t GT_BYTES = GROUP ? (size_t)NW * EBC_VN_GROUPS * EBC_VN_GROUP_PITCH : 0;
  const LdsF4 yt = (LdsF4)(reinterpret_cast<unsigned char *>(fwl + TO * 32) + GT_BYTES + (size_t)wave * 32 * EBC_VN_XROW);
  typedef float __attribute__((address_space(3))) *LdsF;
  typedef double __attribute__((address_space(3))) *LdsD;
  const LdsF ytf = (LdsF)yt;
  const LdsD ytd = (LdsD)yt;
#ifdef EBC_VNS_TRACE  // measurement build (tools/vn_stream_timeline.py): per wave and period, the clock before the
                      // counted wait, after it, after the barrier
  unsigned long long *trl = reinterpret_cast<unsigned long long *>(reinterpret_cast<unsigned char *>(fwl + TO * 32) +
                                                                   GT_BYTES + (FINAL ? 0 : (size_t)NW * 32 * EBC_VN_XROW)) + (size_t)wave * P * 4;
  auto stamp = [&](int p, int k) {
    const unsigned long long t = __builtin_amdgcn_s_memtime();
    if (lane == 0) trl[p * 4 + k] = t;
  };
#else
  auto stamp = [](int, int) {};
#endif

  auto stage = [&](int p) {  // slab p -> ring slot p % 3
    uint4 *dst = sbuf + (p % 3) * SLAB;
#pragma unroll
    for (int j = 0; j < NST; ++j) {
      int q = wave + NW * j;
      if (q > PIECES - 1) q = PIECES - 1;
      // phase A slab i: pieces (u, s, hi/lo) of L1 [u][i][s][hi/lo]; phase C slab t: L2 [t][u][s][hi/lo], contiguous
      const uint4 *src = p < TI ? L1.frag + ((size_t)(q >> 2) * TI + p) * 4 * PART + (size_t)(q & 3) * PART
                                : L2.frag + (size_t)(p - TI) * SLAB + (size_t)q * PART;
      unsigned keep;
      const uint4 *gsrc = src + lane;
      const unsigned lds_dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(dst + (size_t)q * PART));
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
    }
  };
  // the wave's input fragments: [row tile][input tile][k-step][hi, lo][lane] (a tile past M reads tile 0: the tensor has
  // ceil(M / 32) tiles)
  const uint4 *xsrc = ROWS ? nullptr : ex.frag_in + (m0w < M ? (size_t)tile * TI * 4 * PART : 0) + lane;
  vn_u32x4 xr[3][4];  // (a register vector type: asm operands)
  auto xload = [&](auto ic) {
    constexpr int p = decltype(ic)::value;
    const uint4 *src = xsrc + (size_t)p * 4 * PART;
    vn_u32x4 &r0 = xr[p % 3][0], &r1 = xr[p % 3][1], &r2 = xr[p % 3][2], &r3 = xr[p % 3][3];
    if (KIN && p == TI - 1)
      asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %2, off offset:1024" : "=&v"(r0), "=&v"(r1) : "v"(src) : "memory");
    else
      asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %4, off offset:1024\n\t"
                   "global_load_dwordx4 %2, %4, off offset:2048\n\tglobal_load_dwordx4 %3, %4, off offset:3072"
                   : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(src) : "memory");
  };

  // ---- what the loop must not load from memory (an ordinary load's wait would also wait for the staging): the biases,
  // the third layer in accumulator order, the group terms of the wave's rows -> LDS
  for (int q = threadIdx.x; q < TH * 32; q += 64 * NW) hbias[q] = L1.bias[q];
  for (int q = threadIdx.x; q < TO * 32; q += 64 * NW) {
    obias[q] = L2.bias[q];
    if (FINAL) {
      const int t = q >> 5, hh = (q >> 4) & 1, r = q & 15, unit = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
      fwl[q] = unit < O ? ex.final_w[unit] : 0.0f;
    }
  }
  const int Hp = (ex.H + 3) & ~3;  // floats per parked row
  int seg_lo = 0;
  if (!FINAL) {  // the rows' weights, as (w, 0) / (0, w) by the group a row belongs to (the general block's epilogue)
    if (lane < 32) {
      const int row = m0w + lane, hrow0 = m0w + (lane & 16);
      const int hnext = (hrow0 / ex.seg_rows + 1) * ex.seg_rows - hrow0;
      const double w = row < M ? (ex.row_weight ? (double)ex.row_weight[row] : 1.0) : 0.0;
      const bool first = (lane & 15) < hnext;
      ytd[(lane * EBC_VN_XROW + 128) / 8] = first ? w : 0.0;
      ytd[(lane * EBC_VN_XROW + 136) / 8] = first ? 0.0 : w;
    }
    seg_lo = (m0w + 16 * half) / ex.seg_rows - m0w / ex.seg_rows;
  }
  if (GROUP) {
    const int first = m0w / ex.group_rows;
    const int groups_total = (M + ex.group_rows - 1) / ex.group_rows;
    for (int q = lane; q < EBC_VN_GROUPS * (Hp / 4); q += 64) {
      const int gq = q / (Hp / 4), c4 = q - gq * (Hp / 4);
      vn_f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
      if (first + gq < groups_total) v = *reinterpret_cast<const vn_f32x4 *>(ex.row_bias + (size_t)(first + gq) * ex.H + 4 * c4);
      gt[(gq * EBC_VN_GROUP_PITCH) / 16 + c4] = v;
    }
  }
  const int g_local = (GROUP && m < M) ? m / ex.group_rows - m0w / ex.group_rows : -1;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  // ---- prologue: the loads of periods 0 and 1
  if constexpr (ROWS) {  // this lane's row, the 8 columns of its half per k-step: element j of k-step s is k = 16 s + 8 half + j
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int kk = 16 * s2 + 8 * half + j;
        v[j] = (m < M && kk < K0) ? X[(size_t)m * K0 + kk] : 0.0f;
      }
      const Frag2 f = split8(v);
      xr[0][2 * s2] = *reinterpret_cast<const vn_u32x4 *>(&f.hi);
      xr[0][2 * s2 + 1] = *reinterpret_cast<const vn_u32x4 *>(&f.lo);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    xload(IntC<0>{});
  }
  stage(0);
  if constexpr (TI > 1) xload(IntC<1>{});
  stage(1);

  f32x16 hacc[TH];
  // ---- phase A
  auto period_a = [&](auto ic) {
    constexpr int p = decltype(ic)::value;
    // the loads of period p have landed when at most those of period p + 1 are in flight (issue order = period order)
    constexpr int later = NST + (p + 1 < TI ? ((KIN && p + 1 == TI - 1) ? 2 : 4) : 0);
    stamp(p, 0);
    {
      vn_u32x4 &r0 = xr[p % 3][0], &r1 = xr[p % 3][1], &r2 = xr[p % 3][2], &r3 = xr[p % 3][3];
      if (KIN && p == TI - 1) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(r0), "+v"(r1) : "n"(later) : "memory");
      else asm volatile("s_waitcnt vmcnt(%4)" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "n"(later) : "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    stamp(p, 1);
    __builtin_amdgcn_s_barrier();  // every wave's share of slab p is in; every wave is done reading slab p - 1
    asm volatile("" ::: "memory");
    stamp(p, 2);
    if constexpr (p + 2 < TI) xload(IntC<p + 2>{});
    if constexpr (p + 2 < P) stage(p + 2);
    if constexpr (p == 0) {
#pragma unroll
      for (int u = 0; u < TH; ++u) {
        const float4 *hb = reinterpret_cast<const float4 *>(hbias + (u * 2 + half) * 16);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 v = hb[q];
          hacc[u][4 * q] = v.x; hacc[u][4 * q + 1] = v.y; hacc[u][4 * q + 2] = v.z; hacc[u][4 * q + 3] = v.w;
        }
      }
    }
    // the slab's fragments are read TWO steps (a step = one fragment pair, three instructions) ahead of their use, in
    // program order pinned by the scheduling barriers: left alone the compiler reads a pair, waits, multiplies, and the
    // LDS latency of every pair lies open
    const uint4 *w = sbuf + (p % 3) * SLAB + lane;
    constexpr int STEPS = (KIN && p == TI - 1) ? TH : 2 * TH;  // step k: k-step s = k / TH of hidden tile u = k % TH
    uint4 fh[3], fl[3];
    auto fread = [&](int k) {
      const int s = k / TH, u = k % TH;
      fh[k % 3] = w[((u * 2 + s) * 2) * PART];
      fl[k % 3] = w[((u * 2 + s) * 2 + 1) * PART];
    };
    fread(0);
    fread(1);
#pragma unroll
    for (int k = 0; k < STEPS; ++k) {
      if (k + 2 < STEPS) fread(k + 2);
      __builtin_amdgcn_sched_barrier(0);
      const int s = k / TH, u = k % TH;
      const bf16x8 xh = *reinterpret_cast<const bf16x8 *>(&xr[p % 3][2 * s]), xl = *reinterpret_cast<const bf16x8 *>(&xr[p % 3][2 * s + 1]);
      const bf16x8 wh = *reinterpret_cast<const bf16x8 *>(&fh[k % 3]), wl = *reinterpret_cast<const bf16x8 *>(&fl[k % 3]);
      hacc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, xh, hacc[u], 0, 0, 0);
      hacc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xl, hacc[u], 0, 0, 0);
      hacc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xh, hacc[u], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  for_each_int(period_a, typename MakeIntSeq<TI>::type{});

  // ---- phase B: + the group term of this lane's row, ReLU, split
  Frag2 hf[TH][2];
#pragma unroll
  for (int u = 0; u < TH; ++u) {
    f32x16 hid = hacc[u];
#pragma unroll
    for (int g = 0; g < (GROUP ? 4 : 0); ++g) {
      const int unit = u * 32 + 8 * g + 4 * half;
      const bool on = g_local >= 0 && unit + 3 < Hp;  // (read always, from a parked row; selected afterwards: no branch per piece)
      const vn_f32x4 t = gt[((on ? g_local : 0) * EBC_VN_GROUP_PITCH) / 16 + (unit + 3 < Hp ? unit / 4 : 0)];
      hid[4 * g] += on ? t.x : 0.0f; hid[4 * g + 1] += on ? t.y : 0.0f; hid[4 * g + 2] += on ? t.z : 0.0f; hid[4 * g + 3] += on ? t.w : 0.0f;
    }
    tile_frags(hid, true, hf[u]);
  }

  // ---- phase C
  float acc = 0.0f;
  for (int t = 0; t < TO; ++t) {
    const int p = TI + t;
    stamp(p, 0);
    if (t + 1 < TO) vm_wait<NST>();
    else vm_wait<0>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    stamp(p, 1);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    stamp(p, 2);
    if (t + 2 < TO) stage(p + 2);
    f32x16 out;
    {
      const float4 *ob = reinterpret_cast<const float4 *>(obias + (t * 2 + half) * 16);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 v = ob[q];
        out[4 * q] = v.x; out[4 * q + 1] = v.y; out[4 * q + 2] = v.z; out[4 * q + 3] = v.w;
      }
    }
    const uint4 *w = sbuf + (p % 3) * SLAB + lane;
    constexpr int STEPS = KH ? 2 * TH - 1 : 2 * TH;  // step k: k-step s = k % 2 of hidden tile u = k / 2 (the general block's order)
    uint4 fh[3], fl[3];
    auto fread = [&](int k) {
      const int s = k & 1, u = k >> 1;
      fh[k % 3] = w[((u * 2 + s) * 2) * PART];
      fl[k % 3] = w[((u * 2 + s) * 2 + 1) * PART];
    };
    fread(0);
    fread(1);
#pragma unroll
    for (int k = 0; k < STEPS; ++k) {
      if (k + 2 < STEPS) fread(k + 2);
      __builtin_amdgcn_sched_barrier(0);
      const int s = k & 1, u = k >> 1;
      const bf16x8 wh = *reinterpret_cast<const bf16x8 *>(&fh[k % 3]), wl = *reinterpret_cast<const bf16x8 *>(&fl[k % 3]);
      out = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, hf[u][s].hi, out, 0, 0, 0);
      out = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, hf[u][s].lo, out, 0, 0, 0);
      out = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, hf[u][s].hi, out, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    // the third layer's share of this tile: units in register order, like the general block.  Its weights are read as
    // four 16-byte pieces and every product is added — the padding units' weights are parked as 0 and their outputs
    // are finite, so they add +0 — : a test per unit was a branch and a 4-byte LDS read apiece, 16 latencies in a row.
    if constexpr (FINAL) {
      const float4 *fw = reinterpret_cast<const float4 *>(fwl + (t * 2 + half) * 16);
      float fwv[16];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 v = fw[q];
        fwv[4 * q] = v.x; fwv[4 * q + 1] = v.y; fwv[4 * q + 2] = v.z; fwv[4 * q + 3] = v.w;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) acc += fwv[r] * relu_bits(out[r]);
    } else {
      // the general block's pair-sum epilogue on this tile (ebc_value_net.h): parked in the wave's LDS tile, lane
      // (unit, half) walks its column over 16 rows with the rows' weight pairs — the same float64 sums
      const int relu_lo = relu_out ? 0 : (int)0x80000000;
      if (ex.frag_out && m0w < M) {  // the tile as the next block's B fragments: 4 x 16 bytes per lane, 1 KB per store instruction
        uint4 *fout = ex.frag_out + ((size_t)(m0w >> 5) * TO + t) * 4 * PART + lane;
        Frag2 of[2];
        tile_frags(out, relu_out != 0, of);
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          fout[(s2 * 2) * PART] = *reinterpret_cast<const uint4 *>(&of[s2].hi);
          fout[(s2 * 2 + 1) * PART] = *reinterpret_cast<const uint4 *>(&of[s2].lo);
        }
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        vn_f32x4 v;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int b = __float_as_int(out[4 * g + c]);
          v[c] = __int_as_float(b > relu_lo ? b : relu_lo);
        }
        yt[(col * EBC_VN_XROW + (8 * g + 4 * half) * 4) / 16] = v;
      }
      __builtin_amdgcn_wave_barrier();
      double a0 = 0.0, a1 = 0.0;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int r = 16 * half + i;
        const double y = (double)ytf[(r * EBC_VN_XROW) / 4 + col];
        a0 = __builtin_fma(ytd[(r * EBC_VN_XROW + 128) / 8], y, a0);
        a1 = __builtin_fma(ytd[(r * EBC_VN_XROW + 136) / 8], y, a1);
      }
      const int unit = t * 32 + col;
#pragma unroll
      for (int sg = 0; sg < EBC_VN_SEGS; ++sg) {
        double c = (seg_lo == sg ? a0 : 0.0) + (seg_lo + 1 == sg ? a1 : 0.0);
        c += __shfl_xor(c, 32, 64);
        if (half == 0 && unit < O && m0w < M) ex.partial[((size_t)(m0w >> 5) * EBC_VN_SEGS + sg) * O + unit] = c;
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
  if constexpr (FINAL) {
    acc += __shfl_xor(acc, 32, 64);
    if (m < M && half == 0) Y[m] = acc + ex.final_b;
  }
#ifdef EBC_VNS_TRACE
  {
    const unsigned long long t = __builtin_amdgcn_s_memtime();
    if (lane == 0) trl[3] = t;  // (slot 3 of period 0: the wave's end)
    __builtin_amdgcn_wave_barrier();
    unsigned long long *dbg = reinterpret_cast<unsigned long long *>(const_cast<float *>(ex.row_weight));  // (unused by this block)
    if (dbg && lane < P * 4) dbg[(size_t)tile * P * 4 + lane] = trl[lane];
  }
#endif
}

}  // namespace ebc
