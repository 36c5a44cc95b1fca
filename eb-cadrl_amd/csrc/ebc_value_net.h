// ebc_value_net.h — dense layers of the SARL value network (rl/policy/sarl.py:9-82) on the bf16 matrix
// cores at f32 accuracy.
//
// Why not f32 MFMA: v_mfma_f32_32x32x2_f32 runs at 1/16 of the bf16 rate.  Every f32 operand is split
// instead into two bf16 numbers, x = hi + lo (hi = bf16(x), lo = bf16(x - hi)), and a product keeps
// its three leading terms, hi*hi + hi*lo + lo*hi, accumulated in f32 by the MFMA: 16/3 the f32
// matrix rate.  On the reference's decision runs the values move by at most 1.5e-5 (bar: 2e-4;
// tests/split_bf16_accuracy.py); plain bf16 moves them by 7e-3 and changes half the decisions.
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

#include "ebc_vn_common.h"

namespace ebc {

// Y = act(W2 * relu(W1 * X + b1) + b2) for X [M][K0] (K0 <= 32 TI), a workgroup of NW waves, one
// 32-row tile per wave.  TI / TO: input / output tiles (compile time: they index registers); the
// hidden tiles stream: each is made, turned into fragments and consumed.  The weights of a hidden tile
// are staged once per workgroup in LDS — by global_load_lds, 1 KB per wave instruction, no registers in
// between — and double-buffered: while the waves work on hidden tile u from one buffer, tile u + 1
// arrives in the other.  (With the weights coming from L2 per wave and per k-step a wave spent nine
// tenths of its time waiting for them.)
// NW waves per workgroup: 8 (two per SIMD sharing one copy of the weights) when the tiles fit 256
// registers and the double buffer is too big for two workgroups per CU; else 4 (two workgroups per CU when
// LDS allows, one wave per SIMD with up to 512 registers when the tiles need them).
// LEAN: the L2 halves share ONE LDS slot (staged at the start of the hidden phase of their own tile) instead of
// two: 2 A + B instead of 2 A + 2 B bytes, which lets TWO independent 4-wave workgroups live on a CU where one
// lock-stepped 8-wave workgroup did (the 17 -> 300 -> 200 shape, whose full layout already fits twice, is the most
// efficient of the three for that reason: its two workgroups drift apart and fill each other's waits).
// LEAN 2: ONE slot for the L1 halves too (A | B): every half is staged one phase ahead into the slot the barrier just
// passed has freed; for the shape whose lean layout is still too big to fit a CU twice (7 + 7 tiles).
// KIN: the input is at most 32 TI - 16 wide, i.e. the second k-step of its last tile multiplies zeros: left out at
// compile time (one layer-1 MFMA triple in 2 TI, and the eight registers of that fragment).
template <int TI, int TO, int NW, bool GROUP, int LEAN = 0, int KIN = 0>
__global__ __launch_bounds__(64 * NW, ((LEAN || (TI + TO) * 16 + 64 <= 256) ? 2 : 1)) void mlp2_split_wg_kernel(const float *X, int M, int K0, PackedLayer L1,
                                                           PackedLayer L2, int relu_out, float *Y, int O, MlpExtra ex) {
  extern __shared__ uint4 wbuf[];  // weights: A0 | B0 | A1 | B1 (LEAN: A0 | B | A1), then the hidden layer's biases
  constexpr int PART = 64;                          // uint4 per fragment half (hi or lo) = 1 KB
  constexpr int A_SIZE = TI * 2 * 2 * PART, B_SIZE = TO * 2 * 2 * PART;  // uint4 per L1 / L2 half
  constexpr int PER_U = LEAN == 2 ? (A_SIZE + B_SIZE + 1) / 2 : LEAN ? (2 * A_SIZE + B_SIZE + 1) / 2 : A_SIZE + B_SIZE;  // half the weight region, in uint4
  // uint4 free for the input tiles: from A1 on; LEAN 2 has no idle slot: the whole region, weights staged afterwards
  constexpr int XCAP = LEAN == 2 ? A_SIZE + B_SIZE : LEAN ? A_SIZE : A_SIZE + B_SIZE;
  auto off_a = [](int buf) { return (LEAN != 2 && buf) ? A_SIZE + B_SIZE : 0; };
  auto off_b = [](int buf) { return LEAN ? A_SIZE : (buf ? 2 * A_SIZE + B_SIZE : A_SIZE); };
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, col = lane & 31, half = lane >> 5;
  const int m = (blockIdx.x * NW + wave) * 32 + col;
  const int hidden_tiles = L1.out_tiles;
  // Weight staging, global -> LDS without registers (global_load_lds, 1 KB per wave instruction).  A hidden tile's
  // weights are two HALVES with different deadlines: its L1 row block (TI fragments, read by the hidden phase) and
  // its L2 column block (TO fragments, read by the output phase).  which = 0: L1 block, 1: L2 block.
  auto stage_half = [&](int u, int which, int buf) {
    const int first = which ? TI * 4 : 0, count = (which ? TO : TI) * 4;
    for (int q = wave; q < count; q += NW) {
      const int pp = first + q;
      const int blk = pp >> 2, s = (pp >> 1) & 1, part = pp & 1;
      const uint4 *src = blk < TI ? L1.frag + ((((size_t)u * TI + blk) * 2 + s) * 2 + part) * PART
                                  : L2.frag + ((((size_t)(blk - TI) * hidden_tiles + u) * 2 + s) * 2 + part) * PART;
      uint4 *dst = wbuf + (which ? off_b(buf) + (size_t)q * PART : off_a(buf) + (size_t)q * PART);
      // As an asm statement: the builtin makes the compiler drain every outstanding load before the
      // next LDS read (it cannot tell the two buffers apart), which is the overlap this kernel is
      // built for.  M0 = the wave-uniform LDS byte address; each lane supplies its source address.
      unsigned keep;
      const uint4 *gsrc = src + lane;
      const unsigned lds_dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)dst);  // wave-uniform: a scalar register
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
    }
  };
  // Pipelining across barriers: a phase ends with a COUNTED wait that leaves the stage issued at its start in
  // flight (every wave issued at least N_L1 / N_L2 instructions of it; older ones have then landed: one in-order
  // counter) and a raw s_barrier — __syncthreads() would drain vmcnt to 0, i.e. wait for the newest stage at the
  // end of the very phase that issued it (the ~36 %-of-peak ceiling of the two-barrier K-step,
  // cdna_hip_programming.md).  What a phase reads was staged a whole iteration earlier.
  constexpr int N_L1 = (TI * 4) / NW, N_L2 = (TO * 4) / NW;
  auto phase_end = [&](bool newest_in_flight, int n_keep) {
    if (!newest_in_flight) n_keep = 0;
    switch (n_keep) {  // the count is an immediate
      case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
      case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
      case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
      case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
      case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
      case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
      case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  // No ordinary global load inside the loop: while a global_load_lds is in flight the compiler waits
  // for ALL outstanding loads at the next use of a load result, and the overlap would be gone.  The
  // hidden biases therefore wait in LDS.
  float *hbias = reinterpret_cast<float *>(wbuf + 2 * PER_U);
  for (int q = threadIdx.x; q < hidden_tiles * 32; q += 64 * NW) hbias[q] = L1.bias[q];
  if (LEAN != 2) {
    stage_half(0, 0, 0);
    stage_half(0, 1, 0);
  }
  // the input tile as B fragments, natural k order: element j of k-step s is k = 16 s + 8 half + j
  Frag2 x[TI][2][1];
  if (ex.frag_in) {
    // (a workgroup's last waves can lie past the last row tile: the tensor has ceil(M / 32) tiles, they read tile 0)
    const bool tile_ok = (blockIdx.x * NW + wave) * 32 < M;
    const uint4 *src = ex.frag_in + (tile_ok ? (size_t)(blockIdx.x * NW + wave) * TI * 4 * PART : 0) + lane;
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        if (KIN != 0 && i == TI - 1 && s2 == 1) continue;
        const uint4 h = src[((i * 2 + s2) * 2) * PART], l = src[((i * 2 + s2) * 2 + 1) * PART];
        x[i][s2][0].hi = *reinterpret_cast<const bf16x8 *>(&h);
        x[i][s2][0].lo = *reinterpret_cast<const bf16x8 *>(&l);
      }
  } else if (TI >= 2 && (K0 & 3) == 0 && (size_t)NW * 32 * EBC_VN_XROW <= (size_t)XCAP * 16) {
    // Wide inputs (the 200-float h1 rows of mlp2 / attention) through LDS.  A lane needs ITS row, 8 consecutive
    // floats per k-step: read straight from memory that is 28 16-byte loads per lane at a row stride, every
    // 128-byte line touched by eight different wave instructions with 200+ KB per CU in flight — the lines
    // do not survive in the 32 KB L1 and come from L2 eight times (measured: the input phase cost 0.25-0.3 ms of
    // the 0.49 / 0.68 ms launches on 0.5 M rows).  Instead, per 32-float column block: 8 consecutive lanes read
    // one row's 128 bytes (each line exactly once), the wave parks the block in its own LDS tile (row pitch
    // 144 B: the 16-byte reads of 16 lanes then fall in 16 different bank groups) and reads it back row per lane.
    // the tiles live in weight buffer 1: nothing is staged into it before the barrier in front of the loop
    const LdsF4 xt = (LdsF4)(reinterpret_cast<unsigned char *>(wbuf + (LEAN == 2 ? 0 : A_SIZE + B_SIZE)) + (size_t)wave * 32 * EBC_VN_XROW);
    const int m0 = (blockIdx.x * NW + wave) * 32;
    const int piece = lane & 7, rsub = lane >> 3;
#pragma unroll
    for (int i = 0; i < TI; ++i) {
      vn_f32x4 v4[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int r = rsub + 8 * q, kk = i * 32 + 4 * piece;
        v4[q] = vn_f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        if (m0 + r < M && kk + 3 < K0) v4[q] = *reinterpret_cast<const vn_f32x4 *>(X + (size_t)(m0 + r) * K0 + kk);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) xt[((rsub + 8 * q) * EBC_VN_XROW + piece * 16) / 16] = v4[q];
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        if (KIN != 0 && i == TI - 1 && s2 == 1) continue;
        const vn_f32x4 a = xt[(col * EBC_VN_XROW + s2 * 64 + half * 32) / 16], b = xt[(col * EBC_VN_XROW + s2 * 64 + half * 32 + 16) / 16];
        const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        x[i][s2][0] = split8(v);
      }
      __builtin_amdgcn_wave_barrier();  // the tile is rewritten by the next column block
    }
  } else {
#pragma unroll
    for (int u = 0; u < TI; ++u)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        if (KIN != 0 && u == TI - 1 && s == 1) continue;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int kk = u * 32 + 16 * s + 8 * half + j;
          v[j] = (m < M && kk < K0) ? X[(size_t)m * K0 + kk] : 0.0f;
        }
        x[u][s][0] = split8(v);
      }
  }
  f32x16 out[TO][1];
#pragma unroll
  for (int t = 0; t < TO; ++t) out[t][0] = bias_tile(L2, t, lane);
  // the group term of hidden tile u for this lane's row: units 8 g + 4 half + 0..3 of the tile for g = 0..3
  const float *gb = (GROUP && m < M) ? ex.row_bias + (size_t)(m / ex.group_rows) * ex.H : nullptr;
  // The group terms of the (at most EBC_VN_GROUPS) groups a wave's 32 rows belong to wait in LDS: read from memory
  // inside the loop (four 16-byte loads per lane and hidden tile) their waits also waited for the weight staging
  // issued behind them (one in-order counter), and the loop ran at 0.42 instead of ~0.25 ms per 0.5 M rows.
  const int m0w = (blockIdx.x * NW + wave) * 32;
  const bool g_lds = GROUP && (ex.H & 3) == 0 && ex.group_rows > 0 && (31 / ex.group_rows + 2) <= EBC_VN_GROUPS;
  const int Hp = (ex.H + 3) & ~3;  // floats per parked row
  const LdsF4 gt = (LdsF4)(reinterpret_cast<unsigned char *>(wbuf) + 2 * (size_t)PER_U * 16 + (size_t)hidden_tiles * 32 * 4 +
                           (size_t)wave * EBC_VN_GROUPS * EBC_VN_GROUP_PITCH);
  if (g_lds) {
    const int first = m0w / ex.group_rows;
    const int groups_total = (M + ex.group_rows - 1) / ex.group_rows;
    for (int q = lane; q < EBC_VN_GROUPS * (Hp / 4); q += 64) {
      const int gq = q / (Hp / 4), c4 = q - gq * (Hp / 4);
      vn_f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
      if (first + gq < groups_total) v = *reinterpret_cast<const vn_f32x4 *>(ex.row_bias + (size_t)(first + gq) * ex.H + 4 * c4);
      gt[(gq * EBC_VN_GROUP_PITCH) / 16 + c4] = v;
    }
  }
  const int g_local = (GROUP && m < M) ? m / ex.group_rows - m0w / ex.group_rows : 0;
  auto group_bias = [&](int u, float4 (&v)[4]) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int unit = u * 32 + 8 * g + 4 * half;
      v[g] = make_float4(0, 0, 0, 0);
      if (g_lds) {
        if (gb && unit + 3 < Hp) {
          const vn_f32x4 t = gt[(g_local * EBC_VN_GROUP_PITCH) / 16 + unit / 4];
          v[g] = make_float4(t.x, t.y, t.z, t.w);
        }
        continue;
      }
      if (gb && unit + 3 < ex.H) v[g] = *reinterpret_cast<const float4 *>(gb + unit);
      else if (gb) {
        if (unit < ex.H) v[g].x = gb[unit];
        if (unit + 1 < ex.H) v[g].y = gb[unit + 1];
        if (unit + 2 < ex.H) v[g].z = gb[unit + 2];
      }
    }
  };
  // the output biases (ordinary loads) are used HERE: left to the compiler, their wait lands at the accumulators'
  // first use inside the loop — an `s_waitcnt vmcnt(0)` in every iteration's output phase, which drains the staging
#pragma unroll
  for (int t = 0; t < TO; ++t) asm volatile("" : "+v"(out[t][0]));
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the compiler does not count the asm loads
  __syncthreads();                                   // tile 0's weights are in; the input tiles (weight buffer 1) are done with
  if (LEAN == 2) {  // the input tiles had the weight region: tile 0's weights now
    stage_half(0, 0, 0);
    stage_half(0, 1, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  } else if (hidden_tiles > 1) {
    stage_half(1, 0, 1);                             // tile 1's L1 block: read a whole iteration from now
  }
  // Shapes whose input fragments + output accumulators leave few of the 256 registers two waves per SIMD get
  // (7 + 7 tiles: 224): one hidden accumulator instead of two, output tiles two at a time, the group term
  // read after the hidden MFMAs instead of held across them — the partner wave covers the dependent issue.
  constexpr bool TIGHT = (TI + TO) * 16 >= 224;
  // Dimensions are padded to tiles of 32, the MFMA's k-step is 16: when the real input (hidden) width ends in the
  // first half of its last tile, that tile's second k-step multiplies zeros — skipped (200 = 6 tiles + 8: one
  // k-step in 14 of either layer; wave-uniform conditions).
  constexpr bool skip_in = KIN != 0;
#ifdef EBC_VN_HSKIP  // experiment: the last hidden tile's empty second k-step skipped at run time
  const int last_full = (ex.H <= 32 * hidden_tiles - 16) ? hidden_tiles - 1 : hidden_tiles;
#else
  const int last_full = hidden_tiles;
#endif
  for (int u = 0; u < hidden_tiles; ++u) {
    const int buf = u & 1;
    // this tile's group term first (its wait, after the first batch of MFMAs below, then also waits for
    // the staging issued behind it — by then that has had those MFMAs' time), then the next tile's weights
    float4 gcur[4];
    if (GROUP && !TIGHT) group_bias(u, gcur);
    const bool more1 = u + 1 < hidden_tiles, more2 = u + 2 < hidden_tiles;
    if (LEAN) {
      if (u > 0) stage_half(u, 1, buf);        // THIS tile's L2 block into the one slot, free since the last barrier
    } else if (more1) {
      stage_half(u + 1, 1, buf ^ 1);           // next tile's L2 block: its buffer was read last in the output phase of u - 1
    }
    const uint4 *wa = wbuf + off_a(buf) + lane, *wb = wbuf + off_b(buf) + lane;
    auto frag = [&](int blk, int s) {
      const uint4 *w = blk < TI ? wa + ((blk * 2 + s) * 2) * PART : wb + (((blk - TI) * 2 + s) * 2) * PART;
      const uint4 h = w[0], l = w[PART];
      Frag2 f;
      f.hi = *reinterpret_cast<const bf16x8 *>(&h);
      f.lo = *reinterpret_cast<const bf16x8 *>(&l);
      return f;
    };
    // A dependent MFMA waits for the one before it (64 cycles against a 32-cycle issue interval), so
    // consecutive MFMAs go to different accumulators: the hidden tile alternates between two (summed at
    // the end), the output tiles take turns.
    f32x16 hacc[2] = {{}, {}};
    {
      const float4 *hb = reinterpret_cast<const float4 *>(hbias + (u * 2 + half) * 16);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 v = hb[q];
        hacc[0][4 * q] = v.x; hacc[0][4 * q + 1] = v.y; hacc[0][4 * q + 2] = v.z; hacc[0][4 * q + 3] = v.w;
      }
    }
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        if (i == TI - 1 && s == 1 && skip_in) continue;
        const Frag2 wf = frag(i, s);
        constexpr int first = 1;  // three MFMAs per step: 1 0 1 | 0 1 0 | ... never the same accumulator twice in a row
        const int p = TIGHT ? 0 : (((i * 2 + s) & 1) ? 1 - first : first);
        hacc[p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf.lo, x[i][s][0].hi, hacc[p], 0, 0, 0);
        hacc[TIGHT ? 0 : 1 - p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf.hi, x[i][s][0].lo, hacc[TIGHT ? 0 : 1 - p], 0, 0, 0);
        hacc[p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf.hi, x[i][s][0].hi, hacc[p], 0, 0, 0);
      }
    // ---- between the phases: this tile's L2 block (staged during the hidden phase of u - 1) has landed; every
    // wave is done reading this tile's L1 block, so its buffer takes the L1 block of tile u + 2
    phase_end(!LEAN && more1, N_L2);  // LEAN: this tile's L2 block is the newest stage: wait for everything
    if (LEAN == 2) {
      if (more1) stage_half(u + 1, 0, 0);  // the one L1 slot is free: next tile's L1 block, due after the output phase
    } else if (more2) {
      stage_half(u + 2, 0, buf);
    }
    f32x16 hid;
#pragma unroll
    for (int r = 0; r < 16; ++r) hid[r] = TIGHT ? hacc[0][r] : hacc[0][r] + hacc[1][r];
    if (GROUP && TIGHT) group_bias(u, gcur);
    if (GROUP) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        hid[4 * q] += gcur[q].x; hid[4 * q + 1] += gcur[q].y; hid[4 * q + 2] += gcur[q].z; hid[4 * q + 3] += gcur[q].w;
      }
    }
    Frag2 hf[2];
    if (!TIGHT) tile_frags(hid, true, hf);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      if (s == 1 && u >= last_full) continue;
      if (TIGHT) {  // one k-step's fragment at a time: eight registers fewer live across the output MFMAs
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = relu_bits(hid[8 * s + j]);
        hf[s] = split8(v);
      }
#pragma unroll
      for (int t0 = 0; t0 < TO; t0 += (TIGHT ? 2 : 4)) {  // four output tiles at a time: enough independent MFMAs, few registers
        constexpr int C = TIGHT ? 2 : 4;
        Frag2 wf[C];
#pragma unroll
        for (int c = 0; c < C; ++c)
          if (t0 + c < TO) wf[c] = frag(TI + t0 + c, s);
#pragma unroll
        for (int c = 0; c < C; ++c)
          if (t0 + c < TO) out[t0 + c][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[c].lo, hf[s].hi, out[t0 + c][0], 0, 0, 0);
#pragma unroll
        for (int c = 0; c < C; ++c)
          if (t0 + c < TO) out[t0 + c][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[c].hi, hf[s].lo, out[t0 + c][0], 0, 0, 0);
#pragma unroll
        for (int c = 0; c < C; ++c)
          if (t0 + c < TO) out[t0 + c][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[c].hi, hf[s].hi, out[t0 + c][0], 0, 0, 0);
      }
    }
    phase_end(LEAN != 2 && more2, N_L1);  // tile u + 1's L1 block (staged during the output phase of u - 1) has landed
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (ex.final_w) {  // third layer with one output: a dot product over the units this lane holds, then the other half's
    float acc = 0.0f;
#pragma unroll
    for (int t = 0; t < TO; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int unit = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (unit < O) acc += ex.final_w[unit] * relu_bits(out[t][0][r]);
      }
    acc += __shfl_xor(acc, 32, 64);
    if (m < M && half == 0) Y[m] = acc + ex.final_b;
    return;
  }
  if ((O & 3) == 0 && (size_t)NW * 32 * EBC_VN_XROW <= (size_t)PER_U * 2 * 16) {
    // Rows leave coalesced: an output tile (32 rows x 32 units) is parked in the wave's LDS tile — the weight
    // buffers are free behind the loop's last barrier — and read back with 8 consecutive lanes per row, so
    // every 128-byte line of Y is written by one instruction instead of by eight.
    const LdsF4 yt = (LdsF4)(reinterpret_cast<unsigned char *>(wbuf) + (size_t)wave * 32 * EBC_VN_XROW);
    const int m0 = (blockIdx.x * NW + wave) * 32;
    const int piece = lane & 7, rsub = lane >> 3;
    // The row-group sums (MlpExtra.partial).  Lane (col, half) owns unit `col` of the parked tile and walks rows
    // 16 half .. 16 half + 15 down its column: 16 rows meet at most two groups (seg_rows >= 16), so two float64
    // accumulators and the index of the first row of the second group do; the rows' weights wait, as float64, in the
    // 16 spare bytes each 144-byte tile row has.  One exchange between the halves per tile and group finishes it.
    const bool reduce = ex.partial != nullptr;
    typedef float __attribute__((address_space(3))) *LdsF;
    typedef double __attribute__((address_space(3))) *LdsD;
    const LdsF ytf = (LdsF)yt;
    const LdsD ytd = (LdsD)yt;
    int seg_lo = 0;
    if (reduce) {
      // Row r's weight is parked TWICE in the spare bytes, as the pair (w, 0) if the row belongs to the first group its
      // 16-row half meets and (0, w) if to the second: the column walk below is then two fused multiply-adds per row
      // (the product of two float32 is exact in float64, so fma(w, y, a) IS a + w * y) instead of a multiply, two adds
      // and four 32-bit selects.
      if (lane < 32) {
        const int row = m0 + lane, hrow0 = m0 + (lane & 16);
        const int hnext = (hrow0 / ex.seg_rows + 1) * ex.seg_rows - hrow0;  // rows of this row's half before the next group starts
        const double w = row < M ? (ex.row_weight ? (double)ex.row_weight[row] : 1.0) : 0.0;
        const bool first = (lane & 15) < hnext;
        ytd[(lane * EBC_VN_XROW + 128) / 8] = first ? w : 0.0;
        ytd[(lane * EBC_VN_XROW + 136) / 8] = first ? 0.0 : w;
      }
      const int row0 = m0 + 16 * half, s0 = row0 / ex.seg_rows;
      seg_lo = s0 - m0 / ex.seg_rows;
      __builtin_amdgcn_wave_barrier();
    }
    // relu_out as the lower bound of ONE integer max on the bits (0: ReLU; INT_MIN: the value as it is)
    const int relu_lo = relu_out ? 0 : (int)0x80000000;
    auto act = [&](float v) {
      const int b = __float_as_int(v);
      return __int_as_float(b > relu_lo ? b : relu_lo);
    };
    const bool store = ex.store_y != 0 || !reduce;
    // a workgroup's last waves can lie past the last row tile: they own no tile of the outputs that are kept per tile
    uint4 *fout = (ex.frag_out && m0 < M) ? ex.frag_out + (size_t)(m0 >> 5) * TO * 4 * PART + lane : nullptr;
#pragma unroll
    for (int t = 0; t < TO; ++t) {
      if (fout) {  // the tile as the next block's B fragments: 4 x 16 bytes per lane, 1 KB per store instruction
        Frag2 of[2];
        tile_frags(out[t][0], relu_out != 0, of);
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          fout[((t * 2 + s2) * 2) * PART] = *reinterpret_cast<const uint4 *>(&of[s2].hi);
          fout[((t * 2 + s2) * 2 + 1) * PART] = *reinterpret_cast<const uint4 *>(&of[s2].lo);
        }
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        vn_f32x4 v;
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = act(out[t][0][4 * g + c]);
        yt[(col * EBC_VN_XROW + (8 * g + 4 * half) * 4) / 16] = v;  // units 8 g + 4 half .. + 3 of row `col`
      }
      __builtin_amdgcn_wave_barrier();
      if (store) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int r = rsub + 8 * q, unit = t * 32 + 4 * piece;
          const vn_f32x4 v = yt[(r * EBC_VN_XROW + piece * 16) / 16];
          if (m0 + r < M && unit + 3 < O) *reinterpret_cast<vn_f32x4 *>(Y + (size_t)(m0 + r) * O + unit) = v;
        }
      }
      if (reduce) {
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
          if (half == 0 && unit < O && m0 < M) ex.partial[((size_t)(m0 >> 5) * EBC_VN_SEGS + sg) * O + unit] = c;
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
    return;
  }
  if (m < M) {
    const bool vec = (O & 3) == 0;  // registers 4g .. 4g+3 are four consecutive units: one 16-byte store
#pragma unroll
    for (int t = 0; t < TO; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int unit = t * 32 + 8 * g + 4 * half;
        float v[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = relu_out ? relu_bits(out[t][0][4 * g + c]) : out[t][0][4 * g + c];
        float *dst = Y + (size_t)m * O + unit;
        if (vec && unit + 3 < O) {
          *reinterpret_cast<float4 *>(dst) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
#pragma unroll
          for (int c = 0; c < 4; ++c)
            if (unit + c < O) dst[c] = v[c];
        }
      }
  }
}

// ---- the per-pair glue of the value network (rl/policy/sarl.py:52-78), HBM-bound ------------------
// A "pair" = one (env, action) joint state = R consecutive rows, of which the first n_valid[b] exist
// (NULL: all R).  One wave per pair; lanes walk the feature dimension in 16-byte vectors, rows serially.

// w[b][r] = exp(s) (s != 0) (r < n_valid) / sum over the pair (sarl.py:69-71): the row weights the feature block folds
// into its epilogue (MlpExtra.row_weight).  One thread per pair.
__global__ __launch_bounds__(256) void pair_weights_kernel(const float *scores, const long long *n_valid, int B, int R, float *w) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int nv = n_valid ? (int)n_valid[b] : R;
  const float *s = scores + (size_t)b * R;
  float sum = 0.0f;
  for (int r = 0; r < R; ++r) {
    const float v = s[r];
    sum += (r < nv && v != 0.0f) ? expf(v) : 0.0f;
  }
  for (int r = 0; r < R; ++r) {
    const float v = s[r];
    w[(size_t)b * R + r] = r < nv ? (v != 0.0f ? expf(v) : 0.0f) / sum : 0.0f;  // padding rows: exactly 0, whatever the sum
  }
}

// 1 for the rows of a pair that exist, 0 for its padding (the weights of a masked mean).
__global__ __launch_bounds__(256) void pair_mask_kernel(const long long *n_valid, int B, int R, float *w) {
  const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= (size_t)B * R) return;
  const int b = (int)(q / R), r = (int)(q - (size_t)b * R);
  w[q] = r < (int)n_valid[b] ? 1.0f : 0.0f;
}

// out[b][:] = the sum of the tile partials of pair b's rows (MlpExtra.partial), divided by the pair's row count when
// `mean`.  One wave per pair; a pair of R <= 32 rows lies in at most two tiles.
__global__ __launch_bounds__(256) void pair_combine_kernel(const double *partial, const long long *n_valid, int B, int R, int O, int mean,
                                                           float *out) {
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (b >= B) return;
  const int t0 = (int)(((long long)b * R) >> 5), t1 = (int)(((long long)b * R + R - 1) >> 5);
  double denom = 1.0;
  if (mean) {
    const int nv = n_valid ? (int)n_valid[b] : R;
    denom = (double)(nv < 1 ? 1 : nv);
  }
  for (int c = lane * 4; c + 3 < O; c += 256) {
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    for (int t = t0; t <= t1; ++t) {
      const int sg = b - (t * 32) / R;
      const double *src = partial + ((size_t)t * EBC_VN_SEGS + sg) * O + c;
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[k] += src[k];
    }
    vn_f32x4 v;
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = (float)(mean ? acc[k] / denom : acc[k]);
    *reinterpret_cast<vn_f32x4 *>(out + (size_t)b * O + c) = v;
  }
}

// g[b] = mean over the pair's rows of h[b][r][:]  (sarl.py:56-58: the global state), one pass over h.
__global__ __launch_bounds__(256) void pair_mean_kernel(const float *h, const long long *n_valid, int B, int R, int H,
                                                        float *g) {
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (b >= B) return;
  const int nv = n_valid ? (int)n_valid[b] : R;
  const float denom = (float)(nv < 1 ? 1 : nv);
  const float *base = h + (size_t)b * R * H;
  for (int c = lane * 4; c < H; c += 256) {
    if (c + 3 < H) {
      float4 acc = make_float4(0, 0, 0, 0);
      for (int r = 0; r < nv; ++r) {
        const float4 v = *reinterpret_cast<const float4 *>(base + (size_t)r * H + c);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      }
      *reinterpret_cast<float4 *>(g + (size_t)b * H + c) = make_float4(acc.x / denom, acc.y / denom, acc.z / denom, acc.w / denom);
    } else {
      for (int k = c; k < H; ++k) {
        float acc = 0.0f;
        for (int r = 0; r < nv; ++r) acc += base[(size_t)r * H + k];
        g[(size_t)b * H + k] = acc / denom;
      }
    }
  }
}

// out[b] = sum_r w_r feat[b][r][:],  w = the reference's masked softmax of the pair's scores
// (sarl.py:69-72: exp(s) * (s != 0), normalised), rows past n_valid[b] excluded; one pass over feat.
__global__ __launch_bounds__(256) void pair_attend_kernel(const float *scores, const float *feat, const long long *n_valid,
                                                          int B, int R, int F, float *out) {
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (b >= B) return;
  const int nv = n_valid ? (int)n_valid[b] : R;
  // the R <= 64 scores of the pair, one per lane; their sum across the wave
  float e = 0.0f;
  float total = 0.0f;
  for (int r0 = 0; r0 < R; r0 += 64) {  // R > 64: several passes for the normaliser
    const int r = r0 + lane;
    float er = 0.0f;
    if (r < nv) {
      const float sc = scores[(size_t)b * R + r];
      er = sc != 0.0f ? expf(sc) : 0.0f;
    }
    float t = er;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
    total += t;
    if (r0 == 0) e = er;
  }
  // one 16-byte chunk of the features per lane (F <= 256); the row loop is wave-uniform, so the
  // broadcast of a row's weight reads active lanes only
  const float *base = feat + (size_t)b * R * F;
  const int c = lane * 4;
  const bool act = c < F, vec = c + 3 < F;
  float4 acc = make_float4(0, 0, 0, 0);
  for (int r = 0; r < nv; ++r) {
    float er;
    if (r < 64) {
      er = __shfl(e, r, 64);
    } else {
      const float sc = scores[(size_t)b * R + r];
      er = sc != 0.0f ? expf(sc) : 0.0f;
    }
    const float w = er / total;
    if (act) {
      const float *row = base + (size_t)r * F + c;
      if (vec) {
        const float4 v = *reinterpret_cast<const float4 *>(row);
        acc.x += w * v.x; acc.y += w * v.y; acc.z += w * v.z; acc.w += w * v.w;
      } else {
        acc.x += w * row[0];
        if (c + 1 < F) acc.y += w * row[1];
        if (c + 2 < F) acc.z += w * row[2];
      }
    }
  }
  if (act) {
    float *o = out + (size_t)b * F + c;
    if (vec) {
      *reinterpret_cast<float4 *>(o) = acc;
    } else {
      o[0] = acc.x;
      if (c + 1 < F) o[1] = acc.y;
      if (c + 2 < F) o[2] = acc.z;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// The same two-layer block in plain float32 on the vector ALUs (ebc_mlp2_forward_f32): every product an fmaf into a
// float32 sum, like a float32 GEMM.  For the FEW rows whose value decides an argmax (SarlValueNet.action_values
// re-evaluates the candidates within the split-bf16 error of the best one): at ~2 % of the rows its speed does not
// matter, its accuracy does — and the refinement no longer leaves the library for ~11 small hipBLASLt GEMMs and ~20
// element-wise launches.
struct F32Layer {
  const float *wt;  // [in][out_pad]
  const float *b;   // [out_pad]
  int in, out, out_pad;
};
#define EBC_F32_MAXU 5  // units of a layer <= 64 * this (the padded width of the transposed weights)
#define EBC_F32_TILE_ROWS (1 << 16)  // up to this many rows ebc_mlp2_forward_f32 takes the workgroup-per-tile form

// A 32-row tile per workgroup, on the float32 matrix instruction (v_mfma_f32_32x32x2_f32: an exact fmaf chain per
// output, 1/16 of the bf16 rate — plenty for the ~2 % of the rows that are re-evaluated).  Computed transposed like
// the split-bf16 form: units on the accumulator rows, samples on the lanes, so a finished hidden tile IS the B
// operand of the next layer as it stands (register r of lane half h holds unit 8 (r >> 2) + 4 h + (r & 3) of the
// tile: the two units of a k-step are the two lane halves of ONE register, and the second layer's weights are read
// in that pairing).  A hidden tile is made (K0 / 2 instructions), biased, rectified and consumed at once into the
// output accumulators; the input tile waits transposed in LDS; weights are read transposed [K][units padded to 64]
// (a wave's 64 lanes: two contiguous 128-byte runs), one register set ahead of the instructions that use them.
// A workgroup = four waves = four 32-row tiles that walk the hidden tiles TOGETHER: the weights a hidden tile needs
// (its 32 columns of W1, its 32 rows of W2: 54 KB at 200 -> 200 -> 200) are staged in LDS once per workgroup by wide
// coalesced loads and read from there, one ds_read per matrix instruction — with a 4-byte global load per instruction
// and lane the kernel waited on L2 latency nine tenths of its time (187-287 us per launch instead of ~50).  The wave's
// input rows come from global memory (L2) as 16-byte loads, sixteen k-steps ahead of their use.
typedef float f32mfma_acc __attribute__((ext_vector_type(16)));
template <int T2>  // output tiles (compile time: they index registers)
__global__ __launch_bounds__(256, 1) void mlp2_f32_kernel(const float *x, int M, F32Layer L1, F32Layer L2, int relu_out, float *y,
                                                       const float *row_bias, int group_rows, const float *final_w,
                                                       float final_b) {
  extern __shared__ __align__(16) float f32_lds[];  // A1 [K0 rounded up to a multiple of 4][32] | A2 [32][O_pad]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, n = lane & 31, half = lane >> 5;
  const int row = (blockIdx.x * 4 + wave) * 32 + n;
  const int K0 = L1.in, H = L1.out, O = L2.out, P2 = L2.out_pad;
  const int K4 = (K0 + 3) & ~3;  // k-steps come in pairs (one 16-byte load of the input row = two of them)
  float *A1 = f32_lds, *A2 = f32_lds + (size_t)K4 * 32;
  const float *xr = x + (size_t)(row < M ? row : 0) * K0;
  const bool x16 = (K0 & 3) == 0;  // rows 16-byte aligned: the wide loads
  f32mfma_acc out[T2];
#pragma unroll
  for (int t = 0; t < T2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) out[t][r] = 0.0f;
  const int T1 = (H + 31) / 32;
  const int gq = (row_bias && row < M) ? row / group_rows : 0;
  // The wave's input rows, ONCE, as the B operands of all its k-steps (this lane's row, the columns of its lane half):
  // every hidden tile multiplies the same rows, and read per tile from L2 their latency was two thirds of the launch.
  float xq[112];  // K0 <= 224: 56 column quadruples = 112 k-steps
#pragma unroll
  for (int j = 0; j < 56; ++j) {
    const int k = 4 * j;
    float4 q = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (k < K4) {
      if (x16) {
        q = *reinterpret_cast<const float4 *>(xr + k);
      } else {
        q.x = k < K0 ? xr[k] : 0.0f;
        q.y = k + 1 < K0 ? xr[k + 1] : 0.0f;
        q.z = k + 2 < K0 ? xr[k + 2] : 0.0f;
        q.w = k + 3 < K0 ? xr[k + 3] : 0.0f;
      }
    }
    xq[2 * j] = half ? q.y : q.x;      // k-step (k, k + 1): lane half h takes column k + h
    xq[2 * j + 1] = half ? q.w : q.z;  // k-step (k + 2, k + 3)
  }
  for (int th = 0; th < T1; ++th) {
    __syncthreads();  // the previous tile's weights are no longer being read
    {  // all of a thread's loads go out before its first LDS store (a load-wait-store loop paid one L2 latency per piece:
       // 14 of them per hidden tile, 11 us of the 15 a tile took)
      float4 v1[7], v2[8];  // K0 <= 224: at most 7 pieces of W1t[k][32 th .. 32 th + 31] per thread; P2 <= 256: 8 of W2t
      const int q2 = P2 >> 2;
#pragma unroll
      for (int j = 0; j < 7; ++j) {
        const int idx = threadIdx.x + 256 * j, k = idx >> 3, c = (idx & 7) * 4;
        v1[j] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (idx < K4 * 8 && k < K0) v1[j] = *reinterpret_cast<const float4 *>(L1.wt + (size_t)k * L1.out_pad + th * 32 + c);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int idx = threadIdx.x + 256 * j, u = idx / q2, c = (idx - u * q2) * 4;
        v2[j] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (idx < 32 * q2 && th * 32 + u < H) v2[j] = *reinterpret_cast<const float4 *>(L2.wt + (size_t)(th * 32 + u) * P2 + c);
      }
#pragma unroll
      for (int j = 0; j < 7; ++j) {
        const int idx = threadIdx.x + 256 * j, k = idx >> 3, c = (idx & 7) * 4;
        if (idx < K4 * 8) *reinterpret_cast<float4 *>(A1 + k * 32 + c) = v1[j];
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int idx = threadIdx.x + 256 * j, u = idx / q2, c = (idx - u * q2) * 4;
        if (idx < 32 * q2) *reinterpret_cast<float4 *>(A2 + u * P2 + c) = v2[j];
      }
    }
    __syncthreads();
    // ---- hidden tile th: units 32 th .. 32 th + 31 for the wave's 32 rows
    f32mfma_acc hid;
#pragma unroll
    for (int r = 0; r < 16; ++r) hid[r] = 0.0f;
    // sixteen weights are read from LDS FIRST, then sixteen instructions run back to back (left to the compiler each
    // instruction waited for its own ds_read: 170 instead of 64 cycles apiece)
#pragma unroll
    for (int jb = 0; jb < 7; ++jb) {
      if (32 * jb < K4) {  // group-uniform
        float a[16];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int k = 32 * jb + 4 * j < K4 ? 32 * jb + 4 * j : 0;
          a[2 * j] = A1[(k + half) * 32 + n];
          a[2 * j + 1] = A1[(k + 2 + half) * 32 + n];
        }
        __builtin_amdgcn_sched_barrier(0);  // the scheduler otherwise sinks every read next to its use again
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          if (32 * jb + 4 * j < K4) {
            hid = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2 * j], xq[16 * jb + 2 * j], hid, 0, 0, 0);
            hid = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2 * j + 1], xq[16 * jb + 2 * j + 1], hid, 0, 0, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // bias (+ the group's term), ReLU: the tile is now the second layer's B operand, register by register
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int unit = th * 32 + 8 * (r >> 2) + 4 * half + (r & 3);
      float v = hid[r] + (unit < H ? L1.b[unit] : 0.0f);
      if (row_bias && unit < H && row < M) v += row_bias[(size_t)gq * H + unit];
      hid[r] = v > 0.0f ? v : 0.0f;
    }
    // ---- consumed at once: out[t2] += W2[units of t2][units of th] . hid
    float wa[T2], wb[T2];  // the weights of step r + 1 are read while step r's instructions run
    auto read_w2 = [&](int r, float (&w)[T2]) {
      const float *w2 = A2 + (8 * (r >> 2) + 4 * half + (r & 3)) * P2 + n;
#pragma unroll
      for (int t = 0; t < T2; ++t) w[t] = w2[t * 32];
    };
    read_w2(0, wa);
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      read_w2(r + 1, wb);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < T2; ++t) out[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[t], hid[r], out[t], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (r + 2 < 16) read_w2(r + 2, wa);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < T2; ++t) out[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wb[t], hid[r + 1], out[t], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // ---- epilogue
  float fin = 0.0f;
#pragma unroll
  for (int t = 0; t < T2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int unit = t * 32 + 8 * (r >> 2) + 4 * half + (r & 3);
      if (unit < O) {
        float v = out[t][r] + L2.b[unit];
        if (relu_out || final_w) v = v > 0.0f ? v : 0.0f;  // the one-output tail acts on relu(out), like the matrix-core form
        if (final_w) fin = __builtin_fmaf(final_w[unit], v, fin);
        else if (row < M) y[(size_t)row * O + unit] = v;
      }
    }
  if (final_w) {
    fin += __shfl_xor(fin, 32, 64);
    if (half == 0 && row < M) y[row] = fin + final_b;
  }
}

// The same block when the rows are FEW (a decision re-evaluates ~1000 rows = ~34 tiles): above, a wave walks all
// hidden and output tiles of its 32 rows alone — ~1500 dependent 64-cycle instructions and 14 staging round trips,
// 144 us for a launch that fills 9 of 256 CUs.  Here a workgroup is ONE 32-row tile and its waves share the tile's
// work: wave w makes hidden tiles w, w + NW, ... (all of K0 each), parks them in LDS in the accumulator layout (again
// the next layer's B operand as it stands), and after one barrier makes output tiles w, w + NW, ... from all of them.
// The input tile is loaded once per workgroup, transposed into LDS; weights come straight from L2 in the transposed
// layout (a wave's lanes: two contiguous 128-byte runs per instruction), sixteen instructions' worth per batch, two
// batches ahead of the instructions that use them.  Same products and float32 sums per output, k ascending.
template <int NW>
__global__ __launch_bounds__(NW * 64) void mlp2_f32_tile_kernel(const float *x, int M, F32Layer L1, F32Layer L2, int relu_out, float *y,
                                                                const float *row_bias, int group_rows, const float *final_w,
                                                                float final_b) {
  extern __shared__ __align__(16) float f32t_lds[];  // x [K4][32] | hidden [T1][4][64] float4 | fin [NW][32]
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // (uniform, and known to be: scalar branches and offsets)
  const int lane = threadIdx.x & 63, n = lane & 31, half = lane >> 5;
  const int m0 = blockIdx.x * 32, row = m0 + n;
  const int K0 = L1.in, H = L1.out, O = L2.out, P1 = L1.out_pad, P2 = L2.out_pad;
  const int K4 = (K0 + 3) & ~3, T1 = (H + 31) / 32, T2 = (O + 31) / 32;
  float *xl = f32t_lds;
  float4 *hl = reinterpret_cast<float4 *>(f32t_lds + (size_t)K4 * 32);
  float *finl = f32t_lds + (size_t)K4 * 32 + (size_t)T1 * 1024;
  // weights through buffer descriptors: one 32-bit lane offset per tile, the k-step in the scalar offset, and the
  // range check instead of clamps (with 64-bit addresses per load the compiler kept 112 of them live and spilled)
  const auto r1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(L1.wt), 0, K0 * P1 * 4, 0x00020000);
  const auto r2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(L2.wt), 0, H * P2 * 4, 0x00020000);
  // ---- the input tile, once: thread -> (column quadruple c, row r), transposed into xl[k][r] (lanes = rows: no conflict)
  {
    const bool x16 = (K0 & 3) == 0;
    for (int idx = threadIdx.x; idx < K4 * 8; idx += NW * 64) {
      const int c = idx >> 5, r = idx & 31, k = 4 * c;
      const float *xr = x + (size_t)(m0 + r < M ? m0 + r : 0) * K0;
      float4 q;
      if (x16) {
        q = *reinterpret_cast<const float4 *>(xr + k);
      } else {
        q.x = k < K0 ? xr[k] : 0.0f;
        q.y = k + 1 < K0 ? xr[k + 1] : 0.0f;
        q.z = k + 2 < K0 ? xr[k + 2] : 0.0f;
        q.w = k + 3 < K0 ? xr[k + 3] : 0.0f;
      }
      xl[(k + 0) * 32 + r] = q.x;
      xl[(k + 1) * 32 + r] = q.y;
      xl[(k + 2) * 32 + r] = q.z;
      xl[(k + 3) * 32 + r] = q.w;
    }
  }
  __syncthreads();
  const int gq = (row_bias && row < M) ? row / group_rows : 0;
  // ---- layer 1: hidden tiles wave, wave + NW, ...
  for (int u = wave; u < T1; u += NW) {
    const int voff = (half * P1 + u * 32 + n) * 4;  // the lane's part of the address; the k-step's is wave-uniform
    float a[3][16];
    auto load_a = [&](int jb, float (&w)[16]) {  // batch jb: k-steps (32 jb + 4 j, + 2), this lane half's column of each
#pragma unroll
      for (int j = 0; j < 8; ++j) {  // (rows >= K0 lie past the descriptor's range and read as 0)
        w[2 * j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r1, voff, (32 * jb + 4 * j) * P1 * 4, 0));
        w[2 * j + 1] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r1, voff, (32 * jb + 4 * j + 2) * P1 * 4, 0));
      }
    };
    // No branch inside a batch and no way into one but through the batch before it (the break): the loads two batches
    // ahead are then unconditionally in flight where the compiler counts them, and it waits for exactly the batch it
    // uses.  (Loads and k-steps past K0 cost a batch's tail at most: the descriptor returns 0 for them.)
    load_a(0, a[0]);
    load_a(1, a[1]);
    f32mfma_acc hid;
#pragma unroll
    for (int r = 0; r < 16; ++r) hid[r] = 0.0f;
    auto batch = [&](auto ic) -> bool {
      constexpr int jb = decltype(ic)::value;
      if (32 * jb >= K4) return false;  // uniform
      if (jb + 2 < 7) load_a(jb + 2, a[(jb + 2) % 3]);
      float b[16];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k = 32 * jb + 4 * j < K4 ? 32 * jb + 4 * j : 0;  // (a row of the tile: finite, times weight 0)
        b[2 * j] = xl[(k + half) * 32 + n];
        b[2 * j + 1] = xl[(k + 2 + half) * 32 + n];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        hid = __builtin_amdgcn_mfma_f32_32x32x2f32(a[jb % 3][2 * j], b[2 * j], hid, 0, 0, 0);
        hid = __builtin_amdgcn_mfma_f32_32x32x2f32(a[jb % 3][2 * j + 1], b[2 * j + 1], hid, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      return true;
    };
    (void)(batch(IntC<0>{}) && batch(IntC<1>{}) && batch(IntC<2>{}) && batch(IntC<3>{}) && batch(IntC<4>{}) && batch(IntC<5>{}) &&
           batch(IntC<6>{}));
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int unit = u * 32 + 8 * (r >> 2) + 4 * half + (r & 3);
      float v = hid[r] + (unit < H ? L1.b[unit] : 0.0f);
      if (row_bias && unit < H && row < M) v += row_bias[(size_t)gq * H + unit];
      hid[r] = v > 0.0f ? v : 0.0f;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) hl[(u * 4 + q) * 64 + lane] = make_float4(hid[4 * q], hid[4 * q + 1], hid[4 * q + 2], hid[4 * q + 3]);
  }
  __syncthreads();
  // ---- layer 2: output tiles wave, wave + NW, ... from every hidden tile
  float fin = 0.0f;
  for (int t = wave; t < T2; t += NW) {
    const int voff = (4 * half * P2 + t * 32 + n) * 4;
    float a[3][16];
    auto load_a = [&](int u, float (&w)[16]) {  // hidden tile u: step r pairs units 8 (r >> 2) + (r & 3) and + 4 (the lane halves)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        w[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r2, voff, (u * 32 + 8 * (r >> 2) + (r & 3)) * P2 * 4, 0));
    };
    load_a(0, a[0]);
    load_a(1, a[1]);
    f32mfma_acc out;
#pragma unroll
    for (int r = 0; r < 16; ++r) out[r] = 0.0f;
    auto batch = [&](auto ic) -> bool {
      constexpr int u = decltype(ic)::value;
      if (u >= T1) return false;  // uniform
      if (u + 2 < 10) load_a(u + 2, a[(u + 2) % 3]);
      float b[16];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 v = hl[(u * 4 + q) * 64 + lane];
        b[4 * q] = v.x;
        b[4 * q + 1] = v.y;
        b[4 * q + 2] = v.z;
        b[4 * q + 3] = v.w;
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int r = 0; r < 16; ++r) out = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u % 3][r], b[r], out, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      return true;
    };
    (void)(batch(IntC<0>{}) && batch(IntC<1>{}) && batch(IntC<2>{}) && batch(IntC<3>{}) && batch(IntC<4>{}) && batch(IntC<5>{}) &&
           batch(IntC<6>{}) && batch(IntC<7>{}) && batch(IntC<8>{}) && batch(IntC<9>{}));
    const bool wide = (O & 3) == 0 && !final_w;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int unit0 = t * 32 + 8 * q + 4 * half;
      float v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int unit = unit0 + i;
        v[i] = out[4 * q + i] + (unit < O ? L2.b[unit] : 0.0f);
        if (relu_out || final_w) v[i] = v[i] > 0.0f ? v[i] : 0.0f;  // the one-output tail acts on relu(out)
        if (final_w && unit < O) fin = __builtin_fmaf(final_w[unit], v[i], fin);
      }
      if (final_w || row >= M) continue;
      if (wide) {
        if (unit0 < O) *reinterpret_cast<float4 *>(y + (size_t)row * O + unit0) = make_float4(v[0], v[1], v[2], v[3]);
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (unit0 + i < O) y[(size_t)row * O + unit0 + i] = v[i];
      }
    }
  }
  if (final_w) {  // uniform: the waves' shares of the dot product, added in wave order
    fin += __shfl_xor(fin, 32, 64);
    if (half == 0) finl[wave * 32 + n] = fin;
    __syncthreads();
    if (wave == 0 && half == 0 && row < M) {
      float s = 0.0f;
      for (int w = 0; w < NW && w < T2; ++w) s += finl[w * 32 + n];
      y[row] = s + final_b;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Re-packing a block from weights that live on the DEVICE (ebc_mlp2_update): the training loop changes the network
// every round, and its rollouts are decisions like any others — without this they ran on float32 GEMMs because the
// packed copies were made on the host at creation.  One thread per packed element, the host packer's layout
// (pack_layer / pack_f32 in ebcsim_value_net.hip) and its rounding (round-to-nearest-even bf16 of v and of v - hi):
// the fragments come out bit-equal to the host's.
__device__ __forceinline__ unsigned short bf16_rne_bits(float v) {
  unsigned u = __float_as_uint(v);
  if ((u & 0x7f800000u) == 0x7f800000u) return (unsigned short)(u >> 16);  // inf / nan
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
__global__ __launch_bounds__(256) void mlp2_pack_kernel(const float *W, const float *b, int out, int in, int acc_order,
                                                        unsigned short *frag, float *bias, float *wt, float *bt, int out_pad) {
  const int To = (out + 31) / 32, Ti = (in + 31) / 32;
  const size_t n_frag = (size_t)To * Ti * 2 * 64 * 8;  // (t, u, s, lane, j): hi and lo written together
  const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q < n_frag) {
    const int j = (int)(q & 7), lane = (int)((q >> 3) & 63), sstep = (int)((q >> 9) & 1);
    const size_t tu = q >> 10;
    const int u = (int)(tu % Ti), t = (int)(tu / Ti);
    const int i = lane & 31, hh = lane >> 5;
    const int k = acc_order ? 16 * sstep + 8 * (j >> 2) + 4 * hh + (j & 3) : 16 * sstep + 8 * hh + j;
    const int row = t * 32 + i, colk = u * 32 + k;
    const float v = (row < out && colk < in) ? W[(size_t)row * in + colk] : 0.0f;
    const unsigned short hi = bf16_rne_bits(v);
    const unsigned short lo = bf16_rne_bits(v - __uint_as_float((unsigned)hi << 16));
    const size_t base = ((((size_t)t * Ti + u) * 2 + sstep) * 2) * 64 * 8;
    frag[base + (size_t)lane * 8 + j] = hi;
    frag[base + 64 * 8 + (size_t)lane * 8 + j] = lo;
  }
  if (q < (size_t)To * 32) {  // bias in accumulator order [out tile][lane half][16]
    const int r = (int)(q & 15), hh = (int)((q >> 4) & 1), t = (int)(q >> 5);
    const int unit = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
    bias[q] = unit < out ? b[unit] : 0.0f;
  }
  if (q < (size_t)in * out_pad) {  // float32 transposed copy [in][out_pad]
    const int o = (int)(q % out_pad), k = (int)(q / out_pad);
    wt[q] = o < out ? W[(size_t)o * in + k] : 0.0f;
  }
  if (q < (size_t)out_pad) bt[q] = q < (size_t)out ? b[q] : 0.0f;
}

}  // namespace ebc
