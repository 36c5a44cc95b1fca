// ebc_orca_group.h — ORCA with a GS-lane group per human, K items per lane: (GS, K) = (4, 1),
// (8, 1), (8, 2) or (16, 2) for up to 4 / 8 / 16 / 32 others.
//
// Why: with one lane per human (ebc_device.h) a wave walks 9 neighbours, 9 ORCA lines and the
// incremental LP serially: 9 k instructions per wave, < 1 wave per SIMD at 4096 x 10, half the
// wave's cycles in s_waitcnt (profiles/r01_v1_*).  Here lane j of a group owns "other" j of its
// human: neighbour ranking, line construction, the violated-line search of linearProgram2 and
// the interval reductions of linearProgram1 all run across the group, and a wave carries
// 64 / GS humans, so the same batch fills every SIMD several waves deep.
//
// Arithmetic is the float RVO2 arithmetic of ebc_device.h / oracle/ebc_oracle.c, operation for
// operation.  What is re-associated is only min / max over candidate interval ends (exact, order
// free) and the position of early exits (a failing prefix of linearProgram1 fails at the end as
// well: tLeft only grows, tRight only shrinks).  Results are bit-identical to the scalar form;
// tests/test_gpu_parity.py holds that bar.
#pragma once

#include "ebc_device.h"

namespace ebc {

// LDS scratch is private to a wave and a wave's LDS instructions execute in order, so lanes of
// one wave only need the COMPILER to keep a write before the reads that follow it; no
// s_barrier (workgroups of several waves run different numbers of these).
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- group collectives (all lanes of the group are active together) -----------------------
// min / max all-reduce over the group with DPP row operations fused into the arithmetic
// instruction (v_min_f32_dpp: dst = min(permuted src0, src1)).  Written as asm because the
// builtin route costs a v_mov_b32_dpp, two canonicalising v_max and the min per step; the
// "s_nop 1" is the two wait states a DPP read needs after the VALU write of its source
// (hipcc does not insert hazards inside asm).  quad_perm [1,0,3,2] / [2,3,0,1] exchange within
// 4 lanes, row_half_mirror within 8, row_mirror within 16; 32-lane groups add one shuffle.
#define EBC_DPP_STEP(op, ctrl)                                                                     \
  asm volatile("s_nop 1\n\t" op " %0, %1, %1 " ctrl " row_mask:0xf bank_mask:0xf bound_ctrl:1"   \
               : "=v"(r)                                                                           \
               : "v"(v));                                                                          \
  v = r;

// Groups whose size is not 4 / 8 / 16 (9 others -> 9 lanes, 7 humans per wave instead of 4) reduce
// round a ring instead: lane j takes the value of the lane 1, 2, 4, 8 ... places further round its
// group (ds_bpermute_b32: a cross-lane read through the LDS crossbar, no LDS memory), which
// covers the whole group after ceil(log2 GS) steps because min / max are idempotent.
template <int GS>
struct GroupRing {
  static constexpr bool dpp = GS == 4 || GS == 8 || GS == 16 || GS == 32;
  static constexpr int steps = GS <= 2 ? 1 : GS <= 4 ? 2 : GS <= 8 ? 3 : GS <= 16 ? 4 : 5;
  int addr[steps];  // byte address (lane * 4) of each step's source lane
  __device__ __forceinline__ GroupRing(int lane, int j) {
    if (!dpp) {
#pragma unroll
      for (int t = 0; t < steps; ++t) addr[t] = (lane + (1 << t) - (j + (1 << t) >= GS ? GS : 0)) * 4;
    }
  }
};

template <int GS>
__device__ __forceinline__ float group_min(float v, const GroupRing<GS> &ring) {
  if constexpr (GroupRing<GS>::dpp) {
    float r;
    EBC_DPP_STEP("v_min_f32_dpp", "quad_perm:[1,0,3,2]")
    EBC_DPP_STEP("v_min_f32_dpp", "quad_perm:[2,3,0,1]")
    if (GS >= 8) { EBC_DPP_STEP("v_min_f32_dpp", "row_half_mirror") }
    if (GS >= 16) { EBC_DPP_STEP("v_min_f32_dpp", "row_mirror") }
    if (GS >= 32) v = fminf(v, __shfl_xor(v, 16, 64));
  } else {
#pragma unroll
    for (int t = 0; t < GroupRing<GS>::steps; ++t) {
      const float o = __int_as_float(__builtin_amdgcn_ds_bpermute(ring.addr[t], __float_as_int(v)));
      asm("v_min_f32_e32 %0, %1, %2" : "=v"(v) : "v"(o), "v"(v));  // no NaN reaches here
    }
  }
  return v;
}
template <int GS>
__device__ __forceinline__ float group_max(float v, const GroupRing<GS> &ring) {
  if constexpr (GroupRing<GS>::dpp) {
    float r;
    EBC_DPP_STEP("v_max_f32_dpp", "quad_perm:[1,0,3,2]")
    EBC_DPP_STEP("v_max_f32_dpp", "quad_perm:[2,3,0,1]")
    if (GS >= 8) { EBC_DPP_STEP("v_max_f32_dpp", "row_half_mirror") }
    if (GS >= 16) { EBC_DPP_STEP("v_max_f32_dpp", "row_mirror") }
    if (GS >= 32) v = fmaxf(v, __shfl_xor(v, 16, 64));
  } else {
#pragma unroll
    for (int t = 0; t < GroupRing<GS>::steps; ++t) {
      const float o = __int_as_float(__builtin_amdgcn_ds_bpermute(ring.addr[t], __float_as_int(v)));
      asm("v_max_f32_e32 %0, %1, %2" : "=v"(v) : "v"(o), "v"(v));
    }
  }
  return v;
}
// bits of `pred` over the lanes of this lane's group, bit j = lane j of the group
template <int GS>
__device__ __forceinline__ unsigned group_ballot(bool pred, int group) {
  const unsigned long long b = __ballot(pred);
  return (unsigned)((b >> (group * GS)) & (GS == 32 ? 0xFFFFFFFFull : ((1ull << GS) - 1)));
}

// wave mask of the lanes whose index within their GS-lane group is > rel
template <int GS>
__device__ __forceinline__ constexpr unsigned long long lanes_above(int rel) {
  unsigned long long m = 0;
  for (int l = 0; l < 64; ++l)
    if ((l % GS) > rel) m |= 1ull << l;
  return m;
}

struct Line4 {
  float px, py, dx, dy;
};

// A group of GS lanes works on one human; lane j carries K "slots": slot s of lane j is item
// s * GS + j (an "other" before the sort, an ORCA line after it).  (GS, K) = (8, 2) serves the
// 9 others of a 10-human scene with 8 humans per wave instead of the 4 of (16, 1): the
// group-uniform part of the LP (the larger half of a wave's instructions) is shared by twice
// the humans.
// lines per group in LDS: min(maxNeighbors <= 10, items of the group)
template <int GS, int K>
struct OrcaShape {
  static constexpr int ITEMS = GS * K;
  static constexpr int DIST = (ITEMS + 3) / 4 * 4;  // floats per group, float4-readable
  static constexpr int LINES = ITEMS < EBC_MAXNB ? ITEMS : EBC_MAXNB;
};

// linearProgram1 on line `k` (held by every lane as `lk`) against lines 0..k-1, item
// s * GS + j living in own[s].  Returns success; on success (rx, ry) is the new result.
template <int GS, int K>
__device__ __forceinline__ bool lp1_group(const Line4 (&own)[K], int j, const Line4 &lk, int k,
                                          float radius, float ovx, float ovy, bool dirOpt, int group,
                                          float &rx, float &ry) {
  const float dotProduct = lk.px * lk.dx + lk.py * lk.dy;
  const float discriminant = dotProduct * dotProduct + radius * radius - (lk.px * lk.px + lk.py * lk.py);
  const float sq = sqrtf(discriminant < 0.0f ? 0.0f : discriminant);
  float tLeft = -dotProduct - sq;
  float tRight = -dotProduct + sq;
  float candL = -INFINITY, candR = INFINITY;
  bool bad = false;
#pragma unroll
  for (int s = 0; s < K; ++s) {
    if (s * GS + j < k) {
      const float denominator = det2(lk.dx, lk.dy, own[s].dx, own[s].dy);
      const float numerator = det2(own[s].dx, own[s].dy, lk.px - own[s].px, lk.py - own[s].py);
      if (fabsf(denominator) <= RVO_EPS) {
        bad = bad || numerator < 0.0f;
      } else {
        const float t = numerator / denominator;
        if (denominator >= 0.0f)
          candR = fminf(candR, t);
        else
          candL = fmaxf(candL, t);
      }
    }
  }
  const GroupRing<GS> ring(group * GS + j, j);
  tRight = fminf(tRight, group_min<GS>(candR, ring));
  tLeft = fmaxf(tLeft, group_max<GS>(candL, ring));
  const bool any_bad = group_ballot<GS>(bad, group) != 0;
  if (discriminant < 0.0f || any_bad || tLeft > tRight) return false;
  float t;
  if (dirOpt) {
    t = (ovx * lk.dx + ovy * lk.dy > 0.0f) ? tRight : tLeft;
  } else {
    t = lk.dx * (ovx - lk.px) + lk.dy * (ovy - lk.py);
    t = t < tLeft ? tLeft : (t > tRight ? tRight : t);
  }
  rx = lk.px + t * lk.dx;
  ry = lk.py + t * lk.dy;
  return true;
}

// linearProgram2 over n lines (own[s] = line s * GS + j; every line also readable from LDS as
// float4 lines_lds[k], group-uniform address -> broadcast).  Returns lineFail (n = ok).
// Equivalent to the serial scan: the result only changes at a violated line, so the next line
// the serial loop acts on is the first violated one at or after `start`.
template <int GS, int K>
__device__ __forceinline__ int lp2_group(const Line4 (&own)[K], int j, const float4 *lines_lds, int n,
                                         float radius, float ovx, float ovy, bool dirOpt, int group,
                                         float &rx, float &ry) {
  if (dirOpt) {
    rx = ovx * radius;
    ry = ovy * radius;
  } else if (ovx * ovx + ovy * ovy > radius * radius) {
    const float inv = 1.0f / sqrtf(ovx * ovx + ovy * ovy);
    rx = (ovx * inv) * radius;
    ry = (ovy * inv) * radius;
  } else {
    rx = ovx;
    ry = ovy;
  }
  int start = 0;
  while (true) {
    int k = -1;
#pragma unroll
    for (int s = 0; s < K; ++s) {
      const int idx = s * GS + j;
      const bool viol = idx >= start && idx < n &&
                        det2(own[s].dx, own[s].dy, own[s].px - rx, own[s].py - ry) > 0.0f;
      const unsigned m = group_ballot<GS>(viol, group);
      if (k < 0 && m != 0) k = s * GS + __ffs(m) - 1;
    }
    if (k < 0) return n;
    const float4 q = lines_lds[k];
    const Line4 lk{q.x, q.y, q.z, q.w};
    float nx = rx, ny = ry;
    if (!lp1_group<GS, K>(own, j, lk, k, radius, ovx, ovy, dirOpt, group, nx, ny)) return k;
    rx = nx;
    ry = ny;
    start = k + 1;
  }
}

// One human's ORCA velocity, computed by its group.  Slot s of lane j describes "other"
// s * GS + j (already float, as rvo2 holds it); valid[s] = that other exists.  dist_lds:
// OrcaShape::DIST floats, lines_lds / proj_lds: OrcaShape::LINES float4 each, private to the group.  All lanes
// return the same (out_x, out_y).
template <int GS, int K>
__device__ __forceinline__ void orca_group(const EbcParams &p, int j, int group, const bool (&valid)[K],
                                           float posx, float posy, float velx, float vely,
                                           float radius, float maxSpeed, float prefx, float prefy,
                                           const float (&opx)[K], const float (&opy)[K],
                                           const float (&ovx)[K], const float (&ovy)[K],
                                           const float (&orad)[K], float *dist_lds, float4 *lines_lds,
                                           float4 *proj_lds, int max_others, float rangeSq,
                                           float invTimeHorizon, float invTimeStep, float &out_x,
                                           float &out_y) {
  const int maxN = p.orca_max_neighbors < EBC_MAXNB ? p.orca_max_neighbors : EBC_MAXNB;

  // Agent::insertAgentNeighbor: in range, ascending distSq, stable -> rank by counting
  float rpx[K], rpy[K], distSqN[K];
  bool inRange[K];
#pragma unroll
  for (int s = 0; s < K; ++s) {
    rpx[s] = opx[s] - posx;  // relativePosition
    rpy[s] = opy[s] - posy;
    const float ddx = posx - opx[s], ddy = posy - opy[s];
    distSqN[s] = ddx * ddx + ddy * ddy;
    inRange[s] = valid[s] && distSqN[s] < rangeSq;
    dist_lds[s * GS + j] = inRange[s] ? distSqN[s] : INFINITY;
  }
  if (OrcaShape<GS, K>::DIST > GS * K && j < OrcaShape<GS, K>::DIST - GS * K)
    dist_lds[GS * K + j] = INFINITY;  // the tail of the last float4
  wave_sync();
  // rank = #{k : d_k < d or (d_k == d and k < item)}.  "d_k <= d" is "d_k < next float above d"
  // (d >= 0, finite), so each entry costs a select (items before / after this lane's, a constant
  // lane mask), a compare and an add-with-carry.
  int rank[K];
  float dUp[K];
#pragma unroll
  for (int s = 0; s < K; ++s) {
    rank[s] = 0;
    dUp[s] = __uint_as_float(__float_as_uint(distSqN[s]) + 1u);
  }
#pragma unroll
  for (int k4 = 0; k4 < OrcaShape<GS, K>::DIST / 4; ++k4) {
    if (k4 * 4 < max_others) {  // kernel-uniform: items past N - 1 (+ robot) never hold an other
      const float4 d4 = reinterpret_cast<const float4 *>(dist_lds)[k4];
      const float dv[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int k = k4 * 4 + c;
#pragma unroll
        for (int s = 0; s < K; ++s) {
          const int rel = k - s * GS;  // lanes with j > rel see item k before their own
          if (k >= GS * K) {
            // padding, +inf: counts for nobody
          } else if (rel < 0) {        // every lane of the group: k < item
            asm("v_cmp_lt_f32_e32 vcc, %1, %2\n\tv_addc_co_u32_e32 %0, vcc, 0, %0, vcc"
                : "+v"(rank[s]) : "v"(dv[c]), "v"(dUp[s]) : "vcc");
          } else if (rel >= GS - 1) {  // no lane: k >= item
            asm("v_cmp_lt_f32_e32 vcc, %1, %2\n\tv_addc_co_u32_e32 %0, vcc, 0, %0, vcc"
                : "+v"(rank[s]) : "v"(dv[c]), "v"(distSqN[s]) : "vcc");
          } else {
            const unsigned long long after = lanes_above<GS>(rel);
            float x;
            asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(x) : "v"(distSqN[s]), "v"(dUp[s]), "s"(after));
            asm("v_cmp_lt_f32_e32 vcc, %1, %2\n\tv_addc_co_u32_e32 %0, vcc, 0, %0, vcc"
                : "+v"(rank[s]) : "v"(dv[c]), "v"(x) : "vcc");
          }
        }
      }
    }
  }
  int nn = 0;
  bool included[K];
#pragma unroll
  for (int s = 0; s < K; ++s) {
    included[s] = inRange[s] && rank[s] < maxN;
    nn += __popc(group_ballot<GS>(included[s], group));
  }

#pragma unroll
  for (int s = 0; s < K; ++s) {
    if (included[s]) {
      // Agent::computeNewVelocity: the ORCA line of this neighbour.  The three cases of the source
      // (cut-off circle, legs, already colliding) each take one sqrt and one reciprocal of
      // different operands: select the operands, then do one of each.
      const float rvx = velx - ovx[s], rvy = vely - ovy[s];  // relativeVelocity
      const float distSq = rpx[s] * rpx[s] + rpy[s] * rpy[s];
      const float combinedRadius = radius + orad[s];
      const float combinedRadiusSq = combinedRadius * combinedRadius;
      const bool colliding = !(distSq > combinedRadiusSq);
      const float invT = colliding ? invTimeStep : invTimeHorizon;
      const float wx = rvx - invT * rpx[s], wy = rvy - invT * rpy[s];
      const float wLengthSq = wx * wx + wy * wy;
      const float dotProduct1 = wx * rpx[s] + wy * rpy[s];
      const bool circle = colliding || (dotProduct1 < 0.0f && dotProduct1 * dotProduct1 > combinedRadiusSq * wLengthSq);
      const float root = sqrtf(circle ? wLengthSq : distSq - combinedRadiusSq);  // wLength | leg
      const float inv = 1.0f / (circle ? root : distSq);
      float dirx, diry, ux, uy;
      if (circle) {
        const float unx = wx * inv, uny = wy * inv;
        dirx = uny;
        diry = -unx;
        const float sc = combinedRadius * invT - root;
        ux = sc * unx;
        uy = sc * uny;
      } else {
        if (det2(rpx[s], rpy[s], wx, wy) > 0.0f) {
          dirx = (rpx[s] * root - rpy[s] * combinedRadius) * inv;
          diry = (rpx[s] * combinedRadius + rpy[s] * root) * inv;
        } else {
          dirx = -((rpx[s] * root + rpy[s] * combinedRadius) * inv);
          diry = -((-rpx[s] * combinedRadius + rpy[s] * root) * inv);
        }
        const float dotProduct2 = rvx * dirx + rvy * diry;
        ux = dotProduct2 * dirx - rvx;
        uy = dotProduct2 * diry - rvy;
      }
      lines_lds[rank[s]] = make_float4(velx + 0.5f * ux, vely + 0.5f * uy, dirx, diry);
    }
  }
  wave_sync();
  Line4 own[K];
#pragma unroll
  for (int s = 0; s < K; ++s) {
    own[s] = Line4{0, 0, 0, 0};
    if (s * GS + j < nn) {
      const float4 q = lines_lds[s * GS + j];
      own[s] = Line4{q.x, q.y, q.z, q.w};
    }
  }

  float rx, ry;
  const int lineFail = lp2_group<GS, K>(own, j, lines_lds, nn, maxSpeed, prefx, prefy, false, group, rx, ry);

  // linearProgram3 (numObstLines = 0): only groups whose LP2 failed enter; every bound below is
  // group-uniform
  if (lineFail < nn) {
    float distance = 0.0f;
    for (int i = lineFail; i < nn; ++i) {
      const float4 qi = lines_lds[i];
      const Line4 li{qi.x, qi.y, qi.z, qi.w};
      if (!(det2(li.dx, li.dy, li.px - rx, li.py - ry) > distance)) continue;
      // projected lines of items < i, compacted in item order (the serial push_back order)
      bool keep[K];
      float qx[K], qy[K];
      int np = 0, pos[K];
#pragma unroll
      for (int s = 0; s < K; ++s) {
        keep[s] = false;
        qx[s] = qy[s] = 0;
        if (s * GS + j < i) {
          const float determinant = det2(li.dx, li.dy, own[s].dx, own[s].dy);
          if (fabsf(determinant) <= RVO_EPS) {
            if (!(li.dx * own[s].dx + li.dy * own[s].dy > 0.0f)) {
              keep[s] = true;
              qx[s] = 0.5f * (li.px + own[s].px);
              qy[s] = 0.5f * (li.py + own[s].py);
            }
          } else {
            keep[s] = true;
            const float t = det2(own[s].dx, own[s].dy, li.px - own[s].px, li.py - own[s].py) / determinant;
            qx[s] = li.px + t * li.dx;
            qy[s] = li.py + t * li.dy;
          }
        }
        const unsigned km = group_ballot<GS>(keep[s], group);
        pos[s] = np + __popc(km & ((1u << j) - 1u));
        np += __popc(km);
      }
      wave_sync();  // the previous round's reads of proj_lds are done
#pragma unroll
      for (int s = 0; s < K; ++s) {
        if (keep[s]) {
          const float ex = own[s].dx - li.dx, ey = own[s].dy - li.dy;
          const float inv = 1.0f / sqrtf(ex * ex + ey * ey);
          proj_lds[pos[s]] = make_float4(qx[s], qy[s], ex * inv, ey * inv);
        }
      }
      wave_sync();
      Line4 pown[K];
#pragma unroll
      for (int s = 0; s < K; ++s) {
        pown[s] = Line4{0, 0, 0, 0};
        if (s * GS + j < np) {
          const float4 q = proj_lds[s * GS + j];
          pown[s] = Line4{q.x, q.y, q.z, q.w};
        }
      }
      float tx = rx, ty = ry;
      if (lp2_group<GS, K>(pown, j, proj_lds, np, maxSpeed, -li.dy, li.dx, true, group, tx, ty) >= np) {
        rx = tx;
        ry = ty;
      }
      distance = det2(li.dx, li.dy, li.px - rx, li.py - ry);
    }
  }
  out_x = rx;
  out_y = ry;
}

}  // namespace ebc
