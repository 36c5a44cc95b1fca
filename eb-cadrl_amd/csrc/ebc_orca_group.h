// ebc_orca_group.h — ORCA with a group of GS lanes per human, GS = its number of "others"
// (rounded up to an instantiated size): 9 others -> 9 lanes, 7 humans per wave.
//
// Why: with one lane per human (ebc_device.h) a wave walks 9 neighbours, 9 ORCA lines and the
// incremental LP serially: 9 k instructions per wave, < 1 wave per SIMD at 4096 x 10, half the
// wave's cycles in s_waitcnt (profiles/r01_v1_*).  Here lane j of a group owns "other" j of its
// human, then ORCA line j: neighbour ranking, line construction, the feasible segment of every
// line (linearProgram1's constraint work) and the violated-line search of linearProgram2 all run
// across the group.
//
// The shape of the LP matters more than its instruction count: a launch ends with its slowest
// wave, and the slowest waves are the ones whose humans go deep into linearProgram2 / 3
// (tools/wave_timeline.py: median wave 4 us, slowest 12 us with the cross-lane reduction per
// linearProgram1 call this file used to have).  linearProgram1(k) is two things: the segment of
// line k that lines 0..k-1 and the speed circle leave feasible — which depends on the LINES only —
// and the choice of the point of that segment nearest the objective.  So every lane computes the
// segment of its own line up front, all lines in parallel and with no cross-lane traffic, and the
// serial loop that remains per call is: find the first violated line (one ballot), read its line
// and segment (LDS broadcast), clamp, move.
//
// Arithmetic is the float RVO2 arithmetic of ebc_device.h / oracle/ebc_oracle.c, operation for
// operation.  What is re-associated is only min / max over candidate interval ends (exact, order
// free) and the position of early exits (a failing prefix of linearProgram1 fails at the end as
// well: tLeft only grows, tRight only shrinks).  Results are bit-identical to the scalar form;
// tests/test_gpu_parity.py holds that bar.
#pragma once

#include "ebc_device.h"
#include "ebc_trace.h"

namespace ebc {

// LDS scratch is private to a wave and a wave's LDS instructions execute in order, so lanes of
// one wave only need the COMPILER to keep a write before the reads that follow it; no
// s_barrier (workgroups of several waves run different numbers of these).
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
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

// LDS of one group: distances (float4-readable), then lines, segments and projected lines
template <int GS>
struct OrcaShape {
  static constexpr int DIST = (GS + 3) / 4 * 4;                   // floats
  static constexpr int LINES = GS < EBC_MAXNB ? GS : EBC_MAXNB;  // float4 each (maxNeighbors <= 10)
};

// float min / max into an LDS word without a return value (ds_min_f32 / ds_max_f32).  Exact:
// min / max of a set does not depend on order (no NaN reaches here).
__device__ __forceinline__ void lds_fmin(float *p, float v) {
  asm volatile("ds_min_f32 %0, %1" ::"v"((unsigned)(size_t)p), "v"(v) : "memory");
}
__device__ __forceinline__ void lds_fmax(float *p, float v) {
  asm volatile("ds_max_f32 %0, %1" ::"v"((unsigned)(size_t)p), "v"(v) : "memory");
}

// linearProgram1's constraint work for every line at once.  segs_lds[k] = (tLeft, tRight) of line
// k against lines 0..k-1 and the circle of `radius`; infeasible = tLeft > tRight.  The n(n-1)/2
// line pairs are spread evenly over the group as in a round-robin tournament: in round d lane j
// takes the pair {j, (j + d) mod n}, n / 2 rounds in all (the last one half full when n is even),
// works out where the later line of the pair is cut by the earlier one, and folds that into the
// later line's segment with an LDS float min / max.  No lane waits for another.
template <int GS>
__device__ __forceinline__ void lp1_segments(const Line4 &own, int j, const float4 *lines_lds,
                                             float4 *segs_lds, int n, float radius) {
  const bool live = j < n;
  const float dotProduct = own.px * own.dx + own.py * own.dy;
  const float discriminant = dotProduct * dotProduct + radius * radius - (own.px * own.px + own.py * own.py);
  const float sq = sqrtf(discriminant < 0.0f ? 0.0f : discriminant);
  if (live)
    segs_lds[j] = discriminant < 0.0f ? make_float4(INFINITY, -INFINITY, 0.0f, 0.0f)
                                      : make_float4(-dotProduct - sq, -dotProduct + sq, 0.0f, 0.0f);
  wave_sync();
  const int rounds = n >> 1;  // group-uniform
  constexpr int R = OrcaShape<GS>::LINES / 2;
  // every round's partner line is read first (unconditionally: idle lanes and idle rounds read
  // line 0), so the rounds do not queue behind one another's LDS latency
  float4 o[R];
  int partner[R];
#pragma unroll
  for (int d = 1; d <= R; ++d) {
    int pd = j + d;
    if (pd >= n) pd -= n;
    partner[d - 1] = (live && pd < n) ? pd : 0;
    o[d - 1] = lines_lds[partner[d - 1]];
  }
#pragma unroll
  for (int d = 1; d <= R; ++d) {
    if (d <= rounds && live && (2 * d < n || j < d)) {
      const float4 ol = o[d - 1];
      const bool mine = j > partner[d - 1];  // this lane's line is the later one of the pair (lineNo)
      const float kpx = mine ? own.px : ol.x, kpy = mine ? own.py : ol.y;
      const float kdx = mine ? own.dx : ol.z, kdy = mine ? own.dy : ol.w;
      const float qpx = mine ? ol.x : own.px, qpy = mine ? ol.y : own.py;
      const float qdx = mine ? ol.z : own.dx, qdy = mine ? ol.w : own.dy;
      float *seg = reinterpret_cast<float *>(segs_lds + (mine ? j : partner[d - 1]));
      const float denominator = det2(kdx, kdy, qdx, qdy);
      const float numerator = det2(qdx, qdy, kpx - qpx, kpy - qpy);
      if (fabsf(denominator) <= RVO_EPS) {
        if (numerator < 0.0f) lds_fmin(seg + 1, -INFINITY);
      } else {
        const float t = numerator / denominator;
        if (denominator >= 0.0f)
          lds_fmin(seg + 1, t);
        else
          lds_fmax(seg, t);
      }
    }
  }
  wave_sync();
}

// linearProgram2 over n lines (own = line j; lines also in LDS).  Returns lineFail (n = ok).
// Equivalent to the serial scan: the result only changes at a violated line, so the next line the
// serial loop acts on is the first violated one at or after `start`.  The segments are worked out
// when the first violated line turns up: most humans walk at their preferred velocity with no
// line violated, and a wave none of whose humans needs them skips that work altogether.
template <int GS>
__device__ __forceinline__ int lp2_group(const Line4 &own, int j, const float4 *lines_lds,
                                         float4 *segs_lds, int n, float radius, float ovx,
                                         float ovy, bool dirOpt, int group, float &rx, float &ry) {
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
  bool have_segments = false;
  while (true) {
    const bool viol = j >= start && j < n && det2(own.dx, own.dy, own.px - rx, own.py - ry) > 0.0f;
    const unsigned m = group_ballot<GS>(viol, group);
    if (m == 0) return n;
    if (!have_segments) {
      lp1_segments<GS>(own, j, lines_lds, segs_lds, n, radius);
      have_segments = true;
    }
    const int k = __ffs(m) - 1;
    EBC_COUNT(0);
    const float4 lk = lines_lds[k];
    const float4 sg = segs_lds[k];
    if (sg.x > sg.y) return k;  // linearProgram1 fails on line k
    float t;
    if (dirOpt) {
      t = (ovx * lk.z + ovy * lk.w > 0.0f) ? sg.y : sg.x;
    } else {
      t = lk.z * (ovx - lk.x) + lk.w * (ovy - lk.y);
      t = t < sg.x ? sg.x : (t > sg.y ? sg.y : t);
    }
    rx = lk.x + t * lk.z;
    ry = lk.y + t * lk.w;
    start = k + 1;
  }
}

// One human's ORCA velocity, computed by its group.  Lane j describes "other" j (already float,
// as rvo2 holds it); valid = that other exists.  dist_lds: OrcaShape::DIST floats; lines_lds,
// segs_lds, proj_lds: OrcaShape::LINES float4 each, private to the group.  All lanes return the
// same (out_x, out_y).
struct NoHook {
  __device__ __forceinline__ void operator()() const {}
};

// `after_rank`: called once by every lane after the ranking phase (the second form of the step parks loads it
// issued at the wave's start there: they have had the ranking's time to come back).
template <int GS, typename Hook = NoHook>
__device__ __forceinline__ void orca_group(const EbcParams &p, int j, int group, bool valid,
                                           float posx, float posy, float velx, float vely,
                                           float radius, float maxSpeed, float prefx, float prefy,
                                           float opx, float opy, float ovx, float ovy, float orad,
                                           float *dist_lds, float4 *lines_lds, float4 *segs_lds,
                                           float4 *proj_lds, int max_others, float rangeSq,
                                           float invTimeHorizon, float invTimeStep, float &out_x,
                                           float &out_y, Hook after_rank = Hook()) {
  const int maxN = p.orca_max_neighbors < EBC_MAXNB ? p.orca_max_neighbors : EBC_MAXNB;

  // Agent::insertAgentNeighbor: in range, ascending distSq, stable -> rank by counting
  const float rpx = opx - posx, rpy = opy - posy;  // relativePosition
  const float ddx = posx - opx, ddy = posy - opy;
  const float distSqN = ddx * ddx + ddy * ddy;
  const bool inRange = valid && distSqN < rangeSq;
  dist_lds[j] = inRange ? distSqN : INFINITY;
  if (OrcaShape<GS>::DIST > GS && j < OrcaShape<GS>::DIST - GS) dist_lds[GS + j] = INFINITY;  // tail of the last float4
  EBC_MARK(0);
  wave_sync();
  // rank = #{k : d_k < d or (d_k == d and k < j)}.  "d_k <= d" is "d_k < next float above d"
  // (d >= 0, finite), so each entry costs a select (lanes after / before k, a constant lane
  // mask), a compare and an add-with-carry.
  int rank = 0;
  const float dUp = __uint_as_float(__float_as_uint(distSqN) + 1u);
#pragma unroll
  for (int k4 = 0; k4 < OrcaShape<GS>::DIST / 4; ++k4) {
    if (k4 * 4 < max_others) {  // kernel-uniform: items past N - 1 (+ robot) never hold an other
      const float4 d4 = reinterpret_cast<const float4 *>(dist_lds)[k4];
      const float dv[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int k = k4 * 4 + c;
        if (k >= GS) {
          // padding, +inf: counts for nobody
        } else if (k >= GS - 1) {  // no lane has j > k
          asm("v_cmp_lt_f32_e32 vcc, %1, %2\n\tv_addc_co_u32_e32 %0, vcc, 0, %0, vcc"
              : "+v"(rank) : "v"(dv[c]), "v"(distSqN) : "vcc");
        } else {
          const unsigned long long after = lanes_above<GS>(k);
          float x;
          asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(x) : "v"(distSqN), "v"(dUp), "s"(after));
          asm("v_cmp_lt_f32_e32 vcc, %1, %2\n\tv_addc_co_u32_e32 %0, vcc, 0, %0, vcc"
              : "+v"(rank) : "v"(dv[c]), "v"(x) : "vcc");
        }
      }
    }
  }
  EBC_MARK(1);
  after_rank();
  const bool included = inRange && rank < maxN;
  const int nn = __popc(group_ballot<GS>(included, group));

  if (included) {
    // Agent::computeNewVelocity: the ORCA line of this neighbour.  The three cases of the source
    // (cut-off circle, legs, already colliding) each take one sqrt and one reciprocal of
    // different operands: select the operands, then do one of each.
    const float rvx = velx - ovx, rvy = vely - ovy;  // relativeVelocity
    const float distSq = rpx * rpx + rpy * rpy;
    const float combinedRadius = radius + orad;
    const float combinedRadiusSq = combinedRadius * combinedRadius;
    const bool colliding = !(distSq > combinedRadiusSq);
    const float invT = colliding ? invTimeStep : invTimeHorizon;
    const float wx = rvx - invT * rpx, wy = rvy - invT * rpy;
    const float wLengthSq = wx * wx + wy * wy;
    const float dotProduct1 = wx * rpx + wy * rpy;
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
      if (det2(rpx, rpy, wx, wy) > 0.0f) {
        dirx = (rpx * root - rpy * combinedRadius) * inv;
        diry = (rpx * combinedRadius + rpy * root) * inv;
      } else {
        dirx = -((rpx * root + rpy * combinedRadius) * inv);
        diry = -((-rpx * combinedRadius + rpy * root) * inv);
      }
      const float dotProduct2 = rvx * dirx + rvy * diry;
      ux = dotProduct2 * dirx - rvx;
      uy = dotProduct2 * diry - rvy;
    }
    lines_lds[rank] = make_float4(velx + 0.5f * ux, vely + 0.5f * uy, dirx, diry);
  }
  wave_sync();
  Line4 own{0, 0, 0, 0};
  if (j < nn) {
    const float4 q = lines_lds[j];
    own = Line4{q.x, q.y, q.z, q.w};
  }
  EBC_MARK(2);
  float rx, ry;
  const int lineFail = lp2_group<GS>(own, j, lines_lds, segs_lds, nn, maxSpeed, prefx, prefy, false, group, rx, ry);
  EBC_MARK(3);

  // linearProgram3 (numObstLines = 0): only groups whose LP2 failed enter; every bound below is
  // group-uniform
  if (lineFail < nn) {
    EBC_COUNT(1);
    float distance = 0.0f;
    // The serial loop walks i = lineFail .. nn - 1 and acts only on lines the current result
    // violates by more than `distance`; result and distance change only there.  So the next line
    // it acts on is the first such line at or after `cur`: every lane tests its own line, one
    // ballot finds it.  A wave then runs the body max-over-its-humans times, not once per index at
    // which any of its humans (neighbours in one crowded env) happens to be violated.
    int cur = lineFail;
    while (true) {
      const bool far = j >= cur && j < nn && det2(own.dx, own.dy, own.px - rx, own.py - ry) > distance;
      const unsigned fm = group_ballot<GS>(far, group);
      if (fm == 0) break;
      const int i = __ffs(fm) - 1;
      cur = i + 1;
      const float4 qi = lines_lds[i];
      const Line4 li{qi.x, qi.y, qi.z, qi.w};
      EBC_COUNT(2);
      // projected lines of items < i, compacted in item order (the serial push_back order)
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
          const float t = det2(own.dx, own.dy, li.px - own.px, li.py - own.py) / determinant;
          qx = li.px + t * li.dx;
          qy = li.py + t * li.dy;
        }
      }
      const unsigned km = group_ballot<GS>(keep, group);
      const int pos = __popc(km & ((1u << j) - 1u));
      const int np = __popc(km);
      wave_sync();  // the previous round's reads of proj_lds / segs_lds are done
      if (keep) {
        const float ex = own.dx - li.dx, ey = own.dy - li.dy;
        const float inv = 1.0f / sqrtf(ex * ex + ey * ey);
        proj_lds[pos] = make_float4(qx, qy, ex * inv, ey * inv);
      }
      wave_sync();
      Line4 pown{0, 0, 0, 0};
      if (j < np) {
        const float4 q = proj_lds[j];
        pown = Line4{q.x, q.y, q.z, q.w};
      }
      float tx = rx, ty = ry;
      if (lp2_group<GS>(pown, j, proj_lds, segs_lds, np, maxSpeed, -li.dy, li.dx, true, group, tx, ty) >= np) {
        rx = tx;
        ry = ty;
      }
      distance = det2(li.dx, li.dy, li.px - rx, li.py - ry);
    }
  }
  EBC_MARK(4);
  out_x = rx;
  out_y = ry;
}

}  // namespace ebc
