// ebc_orca_group.h — ORCA with a GS-lane group per human (GS = 8, 16 or 32).
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

template <int GS>
__device__ __forceinline__ float group_min(float v) {
  float r;
  EBC_DPP_STEP("v_min_f32_dpp", "quad_perm:[1,0,3,2]")
  EBC_DPP_STEP("v_min_f32_dpp", "quad_perm:[2,3,0,1]")
  EBC_DPP_STEP("v_min_f32_dpp", "row_half_mirror")
  if (GS >= 16) { EBC_DPP_STEP("v_min_f32_dpp", "row_mirror") }
  if (GS >= 32) v = fminf(v, __shfl_xor(v, 16, 64));
  return v;
}
template <int GS>
__device__ __forceinline__ float group_max(float v) {
  float r;
  EBC_DPP_STEP("v_max_f32_dpp", "quad_perm:[1,0,3,2]")
  EBC_DPP_STEP("v_max_f32_dpp", "quad_perm:[2,3,0,1]")
  EBC_DPP_STEP("v_max_f32_dpp", "row_half_mirror")
  if (GS >= 16) { EBC_DPP_STEP("v_max_f32_dpp", "row_mirror") }
  if (GS >= 32) v = fmaxf(v, __shfl_xor(v, 16, 64));
  return v;
}
// bits of `pred` over the lanes of this lane's group, bit j = lane j of the group
template <int GS>
__device__ __forceinline__ unsigned group_ballot(bool pred, int group) {
  const unsigned long long b = __ballot(pred);
  return (unsigned)((b >> (group * GS)) & (GS == 32 ? 0xFFFFFFFFull : ((1ull << GS) - 1)));
}

struct Line4 {
  float px, py, dx, dy;
};

// linearProgram1 on line `k` (held by every lane as `lk`) against lines 0..k-1, lane j holding
// line j in `own`.  Returns success; on success (rx, ry) is the new result.
template <int GS>
__device__ __forceinline__ bool lp1_group(const Line4 &own, int j, const Line4 &lk, int k, float radius,
                                          float ovx, float ovy, bool dirOpt, int group, float &rx,
                                          float &ry) {
  const float dotProduct = lk.px * lk.dx + lk.py * lk.dy;
  const float discriminant = dotProduct * dotProduct + radius * radius - (lk.px * lk.px + lk.py * lk.py);
  const float sq = sqrtf(discriminant < 0.0f ? 0.0f : discriminant);
  float tLeft = -dotProduct - sq;
  float tRight = -dotProduct + sq;
  float candL = -INFINITY, candR = INFINITY;
  bool bad = false;
  if (j < k) {
    const float denominator = det2(lk.dx, lk.dy, own.dx, own.dy);
    const float numerator = det2(own.dx, own.dy, lk.px - own.px, lk.py - own.py);
    if (fabsf(denominator) <= RVO_EPS) {
      bad = numerator < 0.0f;
    } else {
      const float t = numerator / denominator;
      if (denominator >= 0.0f)
        candR = t;
      else
        candL = t;
    }
  }
  tRight = fminf(tRight, group_min<GS>(candR));
  tLeft = fmaxf(tLeft, group_max<GS>(candL));
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

// linearProgram2 over n lines: lane j holds line j (`own`), every line is also readable from
// LDS as float4 lines_lds[k] (group-uniform address -> broadcast).  Returns lineFail (n = ok).
// Equivalent to the serial scan: the result only changes at a violated line, so the next line
// the serial loop acts on is the first violated one at or after `start`.
template <int GS>
__device__ __forceinline__ int lp2_group(const Line4 &own, int j, const float4 *lines_lds, int n,
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
    const bool viol = j >= start && j < n && det2(own.dx, own.dy, own.px - rx, own.py - ry) > 0.0f;
    const unsigned m = group_ballot<GS>(viol, group);
    if (m == 0) return n;
    const int k = __ffs(m) - 1;
    const float4 q = lines_lds[k];
    const Line4 lk{q.x, q.y, q.z, q.w};
    float nx = rx, ny = ry;
    if (!lp1_group<GS>(own, j, lk, k, radius, ovx, ovy, dirOpt, group, nx, ny)) return k;
    rx = nx;
    ry = ny;
    start = k + 1;
  }
}

// One human's ORCA velocity, computed by its group.  Lane j describes "other" j (already float,
// as rvo2 holds it); `valid` = this lane has an other.  lines_lds / proj_lds: this group's LDS
// scratch, GS float4 each.  dist_lds: GS floats.  All lanes return the same (out_x, out_y).
template <int GS>
__device__ __forceinline__ void orca_group(const EbcParams &p, int j, int group, bool valid,
                                           float posx, float posy, float velx, float vely,
                                           float radius, float maxSpeed, float prefx, float prefy,
                                           float opx, float opy, float ovx, float ovy, float orad,
                                           float *dist_lds, float4 *lines_lds, float4 *proj_lds,
                                           int max_others, float &out_x, float &out_y) {
  const float rangeSq = p.orca_neighbor_dist * p.orca_neighbor_dist;
  const float invTimeHorizon = 1.0f / p.orca_time_horizon;
  const float timeStep = (float)p.time_step;
  const int maxN = p.orca_max_neighbors < EBC_MAXNB ? p.orca_max_neighbors : EBC_MAXNB;

  // Agent::insertAgentNeighbor: in range, ascending distSq, stable -> rank by counting
  const float rpx = opx - posx, rpy = opy - posy;  // relativePosition
  const float ddx = posx - opx, ddy = posy - opy;
  const float distSqN = ddx * ddx + ddy * ddy;
  const bool inRange = valid && distSqN < rangeSq;
  dist_lds[j] = inRange ? distSqN : INFINITY;
  wave_sync();
  int rank = 0;
#pragma unroll
  for (int k4 = 0; k4 < GS / 4; ++k4) {
    if (k4 * 4 < max_others) {  // kernel-uniform: slots past N - 1 (+ robot) never hold an other
      const float4 d4 = reinterpret_cast<const float4 *>(dist_lds)[k4];
      const float dv[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int k = k4 * 4 + c;
        rank += (dv[c] < distSqN || (dv[c] == distSqN && k < j)) ? 1 : 0;
      }
    }
  }
  const bool included = inRange && rank < maxN;
  const int nn = __popc(group_ballot<GS>(included, group));

  if (included) {
    // Agent::computeNewVelocity: the ORCA line of this neighbour
    const float rvx = velx - ovx, rvy = vely - ovy;  // relativeVelocity
    const float distSq = rpx * rpx + rpy * rpy;
    const float combinedRadius = radius + orad;
    const float combinedRadiusSq = combinedRadius * combinedRadius;
    float dirx, diry, ux, uy;
    if (distSq > combinedRadiusSq) {
      const float wx = rvx - invTimeHorizon * rpx, wy = rvy - invTimeHorizon * rpy;
      const float wLengthSq = wx * wx + wy * wy;
      const float dotProduct1 = wx * rpx + wy * rpy;
      if (dotProduct1 < 0.0f && dotProduct1 * dotProduct1 > combinedRadiusSq * wLengthSq) {
        const float wLength = sqrtf(wLengthSq);
        const float inv = 1.0f / wLength;
        const float unx = wx * inv, uny = wy * inv;
        dirx = uny;
        diry = -unx;
        const float s = combinedRadius * invTimeHorizon - wLength;
        ux = s * unx;
        uy = s * uny;
      } else {
        const float leg = sqrtf(distSq - combinedRadiusSq);
        const float inv = 1.0f / distSq;
        if (det2(rpx, rpy, wx, wy) > 0.0f) {
          dirx = (rpx * leg - rpy * combinedRadius) * inv;
          diry = (rpx * combinedRadius + rpy * leg) * inv;
        } else {
          dirx = -((rpx * leg + rpy * combinedRadius) * inv);
          diry = -((-rpx * combinedRadius + rpy * leg) * inv);
        }
        const float dotProduct2 = rvx * dirx + rvy * diry;
        ux = dotProduct2 * dirx - rvx;
        uy = dotProduct2 * diry - rvy;
      }
    } else {
      const float invTimeStep = 1.0f / timeStep;
      const float wx = rvx - invTimeStep * rpx, wy = rvy - invTimeStep * rpy;
      const float wLength = sqrtf(wx * wx + wy * wy);
      const float inv = 1.0f / wLength;
      const float unx = wx * inv, uny = wy * inv;
      dirx = uny;
      diry = -unx;
      const float s = combinedRadius * invTimeStep - wLength;
      ux = s * unx;
      uy = s * uny;
    }
    lines_lds[rank] = make_float4(velx + 0.5f * ux, vely + 0.5f * uy, dirx, diry);
  }
  wave_sync();
  Line4 own{0, 0, 0, 0};
  if (j < nn) {
    const float4 q = lines_lds[j];
    own = Line4{q.x, q.y, q.z, q.w};
  }

  float rx, ry;
  const int lineFail = lp2_group<GS>(own, j, lines_lds, nn, maxSpeed, prefx, prefy, false, group, rx, ry);

  // linearProgram3 (numObstLines = 0): only groups whose LP2 failed enter; every bound below is
  // group-uniform
  if (lineFail < nn) {
    float distance = 0.0f;
    for (int i = lineFail; i < nn; ++i) {
      const float4 qi = lines_lds[i];
      const Line4 li{qi.x, qi.y, qi.z, qi.w};
      if (!(det2(li.dx, li.dy, li.px - rx, li.py - ry) > distance)) continue;
      // projected lines of lanes j < i, compacted in lane order (the serial push_back order)
      bool keep = false;
      float qx = 0, qy = 0;
      if (j < i) {
        const float determinant = det2(li.dx, li.dy, own.dx, own.dy);
        if (fabsf(determinant) <= RVO_EPS) {
          if (!(li.dx * own.dx + li.dy * own.dy > 0.0f)) {
            keep = true;
            qx = 0.5f * (li.px + own.px);
            qy = 0.5f * (li.py + own.py);
          }
        } else {
          keep = true;
          const float s = det2(own.dx, own.dy, li.px - own.px, li.py - own.py) / determinant;
          qx = li.px + s * li.dx;
          qy = li.py + s * li.dy;
        }
      }
      const unsigned km = group_ballot<GS>(keep, group);
      const int np = __popc(km);
      wave_sync();  // the previous round's reads of proj_lds are done
      if (keep) {
        const float ex = own.dx - li.dx, ey = own.dy - li.dy;
        const float inv = 1.0f / sqrtf(ex * ex + ey * ey);
        proj_lds[__popc(km & ((1u << j) - 1u))] = make_float4(qx, qy, ex * inv, ey * inv);
      }
      wave_sync();
      Line4 pown{0, 0, 0, 0};
      if (j < np) {
        const float4 q = proj_lds[j];
        pown = Line4{q.x, q.y, q.z, q.w};
      }
      float tx = rx, ty = ry;
      if (lp2_group<GS>(pown, j, proj_lds, np, maxSpeed, -li.dy, li.dx, true, group, tx, ty) >= np) {
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
