// ebc_device.h — device-side arithmetic of the simulation hot path (gfx950).
//
// Every function names the reference lines it computes (paths relative to the
// reference tree).  Simulator state is double, ORCA is float (as inside rvo2),
// rotated observations are float (as torch.Tensor).  Built with
// -ffp-contract=off: the reference is interpreted Python / unfused C, so each
// operation rounds once; the one fused operation (norm2) is explicit.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ebcsim.h"

#define EBC_WAVE 64
#define EBC_MAXNB 10       // rvo2 maxNeighbors of the reference (simulator/policy/orca.py:65)
#define EBC_TILE_SLOTS 128 // per-wave human tile incl. one robot slot per env

namespace ebc {

// np.linalg.norm((x, y)): BLAS ddot fuses the second product (see oracle/ebc_oracle.c).
__device__ __forceinline__ double norm2(double x, double y) { return sqrt(fma(y, y, x * x)); }

// simulator/utils/collisions.py:4-26, with (x3, y3) = (0, 0)
__device__ __forceinline__ double point_to_segment_dist0(double x1, double y1, double x2, double y2) {
  double px = x2 - x1, py = y2 - y1;
  if (px == 0 && py == 0) return norm2(0 - x1, 0 - y1);
  double u = ((0 - x1) * px + (0 - y1) * py) / (px * px + py * py);
  u = u > 1 ? 1 : (u < 0 ? 0 : u);
  double x = x1 + u * px, y = y1 + u * py;
  return norm2(x - 0, y - 0);
}

// simulator/utils/collisions.py:29-57: closest boundary distance over the step
// (negative = collision).  (rvx, rvy) is the robot's velocity under the action:
// ActionXY itself, or v*(cos, sin)(r + theta) for ActionRot (collisions.py:41-42).
__device__ __forceinline__ double closest_dist(double hpx, double hpy, double hvx, double hvy,
                                               double hr, double rpx, double rpy, double rr,
                                               double rvx, double rvy, double dt) {
  double px = hpx - rpx, py = hpy - rpy;
  double vx = hvx - rvx, vy = hvy - rvy;
  double ex = px + vx * dt, ey = py + vy * dt;
  return point_to_segment_dist0(px, py, ex, ey) - hr - rr;
}

// simulator/policy/linear.py:17-23
__device__ __forceinline__ void linear_policy(double px, double py, double gx, double gy,
                                              double v_pref, double &vx, double &vy) {
  double theta = atan2(gy - py, gx - px);
  vx = cos(theta) * v_pref;
  vy = sin(theta) * v_pref;
}

// Agent.compute_position, simulator/agents/agent.py:164-188
__device__ __forceinline__ void robot_next_position(const double *rb, int kin, double a0, double a1,
                                                    double dt, double &nx, double &ny) {
  if (kin == EBC_HOLONOMIC) {
    nx = rb[0] + a0 * dt;
    ny = rb[1] + a1 * dt;
  } else {
    double th = rb[8] + a1;
    nx = rb[0] + cos(th) * a0 * dt;
    ny = rb[1] + sin(th) * a0 * dt;
  }
}

// simulator/env.py:227-271: window of the 128-bit-row occupancy grid around the
// robot's NEXT position; round() is half-to-even = rint.
__device__ __forceinline__ int grid_collision(const uint64_t *grid, int G, double map_size_m,
                                              double map_resolution, double px, double py,
                                              double radius, const double *border) {
  int collision = 0;
  if (grid) {
    long ix = (long)rint((px + map_size_m / 2.0) / map_resolution);
    long iy = (long)rint((py + map_size_m / 2.0) / map_resolution);
    long h = (long)ceil(radius / sqrt(2.0) / map_resolution);
    long lim = (long)rint(map_size_m / map_resolution);
    long sx = ix - h, ex = sx + h * 2;
    sx = sx < 0 ? 0 : sx;
    ex = ex > lim ? lim : ex;
    long sy = iy - h, ey = sy + h * 2;
    sy = sy < 0 ? 0 : sy;
    ey = ey > lim ? lim : ey;
    if (ex > sx && ey > sy) {
      ex = ex > G ? G : ex;
      ey = ey > G ? G : ey;
      if (ey > sy) {
        // mask of columns [sy, ey) in the two 64-bit halves of a row
        uint64_t lo = 0, hi = 0;
        for (long y = sy; y < ey; ++y) {
          if (y < 64) lo |= 1ull << y; else hi |= 1ull << (y - 64);
        }
        for (long x = sx; x < ex; ++x)
          if ((grid[x * 2] & lo) | (grid[x * 2 + 1] & hi)) collision = 1;
      }
    }
  }
  if (border) {
    if (px <= border[0] + radius || px >= border[1] - radius || py <= border[2] + radius ||
        py >= border[3] - radius)
      collision = 1;
  }
  return collision;
}

struct RewardOut {
  double reward, dist_to_goal;
  int done, info;
};

// simulator/utils/reward.py:8-14
__device__ __forceinline__ double time_reward(double x, double time_max, double time_good) {
  if (x < time_good) return 1;
  if (time_good <= x && x <= time_max) return (time_max - x) / (time_max - time_good);
  return 0;
}

// simulator/utils/reward.py:80-181.  coll = adult, bicycle, child, obstacle.
__device__ __forceinline__ RewardOut reward_compute(const EbcParams &p, double nx, double ny,
                                                    double gx, double gy, double radius, double a1,
                                                    double global_time, const double dmin[3],
                                                    const int coll[4]) {
  RewardOut o;
  o.dist_to_goal = norm2(nx - gx, ny - gy);
  int reaching_goal = o.dist_to_goal < radius;
  double reward = 0;
  if (p.new_reward) reward = 1 - o.dist_to_goal / p.max_goal_distance;
  if (global_time >= p.time_limit) {
    o.done = 1; o.info = EBC_INFO_TIMEOUT;
  } else if (coll[2]) {
    reward += p.collision_penalty[2]; o.done = 1; o.info = EBC_INFO_COLLISION_CHILD;
  } else if (coll[1]) {
    reward += p.collision_penalty[1]; o.done = 1; o.info = EBC_INFO_COLLISION_BICYCLE;
  } else if (coll[0]) {
    reward += p.collision_penalty[0]; o.done = 1; o.info = EBC_INFO_COLLISION_ADULT;
  } else if (coll[3]) {
    reward += p.collision_penalty[3]; o.done = 1; o.info = EBC_INFO_COLLISION_OBSTACLE;
  } else if (reaching_goal) {
    reward += p.new_reward ? time_reward(global_time, p.time_max, p.time_good) : p.success_reward;
    o.done = 1; o.info = EBC_INFO_REACH_GOAL;
  } else if (dmin[2] < p.discomfort_dist[2]) {
    reward = (dmin[2] - p.discomfort_dist[2]) * p.discomfort_factor[2] * p.time_step;
    o.done = 0; o.info = EBC_INFO_DANGER;
  } else if (dmin[1] < p.discomfort_dist[1]) {
    reward = (dmin[1] - p.discomfort_dist[1]) * p.discomfort_factor[1] * p.time_step;
    o.done = 0; o.info = EBC_INFO_DANGER;
  } else if (dmin[0] < p.discomfort_dist[0]) {
    reward = (dmin[0] - p.discomfort_dist[0]) * p.discomfort_factor[0] * p.time_step;
    o.done = 0; o.info = EBC_INFO_DANGER;
  } else if (p.robot_kinematics != EBC_HOLONOMIC && fabs(a1) > 0 &&
             p.rotation_penalty_factor != 0) {
    reward = fabs(a1) * p.rotation_penalty_factor; o.done = 0; o.info = EBC_INFO_NOTHING;
  } else {
    reward = 0; o.done = 0; o.info = EBC_INFO_NOTHING;
  }
  o.reward = reward;
  return o;
}

// ---------------------------------------------------------------------------------
// ORCA (RVO2 v2.0 Agent.cpp / Vector2.h) in float, one lane per human.  ORCA lines
// live in LDS, lane-interleaved: element c of line k of lane l is L[(k*4+c)*64 + l]
// (bank = l % 32: conflict-free).  Line = {point.x, point.y, dir.x, dir.y}.
// ---------------------------------------------------------------------------------
#define RVO_EPS 0.00001f

struct LineSet {
  float *base;  // LDS, [EBC_MAXNB*4][64]
  int lane;
  __device__ __forceinline__ float get(int k, int c) const { return base[(k * 4 + c) * EBC_WAVE + lane]; }
  __device__ __forceinline__ void set(int k, float px, float py, float dx, float dy) {
    base[(k * 4 + 0) * EBC_WAVE + lane] = px;
    base[(k * 4 + 1) * EBC_WAVE + lane] = py;
    base[(k * 4 + 2) * EBC_WAVE + lane] = dx;
    base[(k * 4 + 3) * EBC_WAVE + lane] = dy;
  }
};

__device__ __forceinline__ float det2(float ax, float ay, float bx, float by) { return ax * by - ay * bx; }

// linearProgram1
__device__ __forceinline__ bool lp1(const LineSet &L, int lineNo, float radius, float ovx, float ovy,
                                    bool dirOpt, float &rx, float &ry) {
  const float ppx = L.get(lineNo, 0), ppy = L.get(lineNo, 1);
  const float pdx = L.get(lineNo, 2), pdy = L.get(lineNo, 3);
  const float dotProduct = ppx * pdx + ppy * pdy;
  const float discriminant = dotProduct * dotProduct + radius * radius - (ppx * ppx + ppy * ppy);
  if (discriminant < 0.0f) return false;
  const float sq = sqrtf(discriminant);
  float tLeft = -dotProduct - sq;
  float tRight = -dotProduct + sq;
  for (int i = 0; i < lineNo; ++i) {
    const float ipx = L.get(i, 0), ipy = L.get(i, 1), idx = L.get(i, 2), idy = L.get(i, 3);
    const float denominator = det2(pdx, pdy, idx, idy);
    const float numerator = det2(idx, idy, ppx - ipx, ppy - ipy);
    if (fabsf(denominator) <= RVO_EPS) {
      if (numerator < 0.0f) return false;
      continue;
    }
    const float t = numerator / denominator;
    if (denominator >= 0.0f)
      tRight = fminf(tRight, t);
    else
      tLeft = fmaxf(tLeft, t);
    if (tLeft > tRight) return false;
  }
  float t;
  if (dirOpt) {
    t = (ovx * pdx + ovy * pdy > 0.0f) ? tRight : tLeft;
  } else {
    t = pdx * (ovx - ppx) + pdy * (ovy - ppy);
    t = t < tLeft ? tLeft : (t > tRight ? tRight : t);
  }
  rx = ppx + t * pdx;
  ry = ppy + t * pdy;
  return true;
}

// linearProgram2
__device__ __forceinline__ int lp2(const LineSet &L, int n, float radius, float ovx, float ovy,
                                   bool dirOpt, float &rx, float &ry) {
  if (dirOpt) {
    rx = ovx * radius;
    ry = ovy * radius;
  } else if (ovx * ovx + ovy * ovy > radius * radius) {
    const float inv = 1.0f / sqrtf(ovx * ovx + ovy * ovy);  // normalize(): v * (1 / |v|)
    rx = (ovx * inv) * radius;
    ry = (ovy * inv) * radius;
  } else {
    rx = ovx;
    ry = ovy;
  }
  for (int i = 0; i < n; ++i) {
    if (det2(L.get(i, 2), L.get(i, 3), L.get(i, 0) - rx, L.get(i, 1) - ry) > 0.0f) {
      const float tx = rx, ty = ry;
      if (!lp1(L, i, radius, ovx, ovy, dirOpt, rx, ry)) {
        rx = tx;
        ry = ty;
        return i;
      }
    }
  }
  return n;
}

// linearProgram3 (numObstLines = 0); P = projected-lines scratch in LDS
__device__ __forceinline__ void lp3(const LineSet &L, LineSet &P, int n, int beginLine, float radius,
                                    float &rx, float &ry) {
  float distance = 0.0f;
  for (int i = beginLine; i < n; ++i) {
    const float ipx = L.get(i, 0), ipy = L.get(i, 1), idx = L.get(i, 2), idy = L.get(i, 3);
    if (det2(idx, idy, ipx - rx, ipy - ry) > distance) {
      int np = 0;
      for (int j = 0; j < i; ++j) {
        const float jpx = L.get(j, 0), jpy = L.get(j, 1), jdx = L.get(j, 2), jdy = L.get(j, 3);
        float qx, qy;
        const float determinant = det2(idx, idy, jdx, jdy);
        if (fabsf(determinant) <= RVO_EPS) {
          if (idx * jdx + idy * jdy > 0.0f) continue;
          qx = 0.5f * (ipx + jpx);
          qy = 0.5f * (ipy + jpy);
        } else {
          const float s = det2(jdx, jdy, ipx - jpx, ipy - jpy) / determinant;
          qx = ipx + s * idx;
          qy = ipy + s * idy;
        }
        const float ex = jdx - idx, ey = jdy - idy;
        const float inv = 1.0f / sqrtf(ex * ex + ey * ey);
        P.set(np++, qx, qy, ex * inv, ey * inv);
      }
      const float tx = rx, ty = ry;
      if (lp2(P, np, radius, -idy, idx, true, rx, ry) < np) {
        rx = tx;
        ry = ty;
      }
      distance = det2(idx, idy, ipx - rx, ipy - ry);
    }
  }
}

// One human's ORCA velocity.  tile_* = this env's humans (and robot slot) in LDS as the
// floats rvo2 holds: position, velocity, radius + 0.01 + safety.  Others are visited in
// ob order (simulator/env.py:396-402): humans j != self, then the robot if visible.
// Neighbour selection = Agent::insertAgentNeighbor: the maxNeighbors nearest within
// neighborDist, ascending, ties by arrival -> stable rank, computed by counting.
__device__ __forceinline__ void orca_velocity(const EbcParams &p, int self, int n_agents,
                                              const float *tpx, const float *tpy, const float *tvx,
                                              const float *tvy, const float *trad, float maxSpeed,
                                              float prefx, float prefy, LineSet &L, LineSet &P,
                                              float &out_x, float &out_y) {
  const float posx = tpx[self], posy = tpy[self];
  const float velx = tvx[self], vely = tvy[self];
  const float radius = trad[self];
  const float rangeSq = p.orca_neighbor_dist * p.orca_neighbor_dist;
  const float invTimeHorizon = 1.0f / p.orca_time_horizon;
  const float timeStep = (float)p.time_step;
  const int maxN = p.orca_max_neighbors < EBC_MAXNB ? p.orca_max_neighbors : EBC_MAXNB;
  int nn = 0;
  for (int j = 0; j < n_agents; ++j) {
    if (j == self) continue;
    const float rpx = tpx[j] - posx, rpy = tpy[j] - posy;      // relativePosition
    const float ddx = posx - tpx[j], ddy = posy - tpy[j];      // position_ - other (insertAgentNeighbor)
    const float distSqN = ddx * ddx + ddy * ddy;
    if (!(distSqN < rangeSq)) continue;
    int rank = 0;
    for (int k = 0; k < n_agents; ++k) {
      if (k == self || k == j) continue;
      const float ex = posx - tpx[k], ey = posy - tpy[k];
      const float dk = ex * ex + ey * ey;
      rank += (dk < distSqN || (dk == distSqN && k < j)) ? 1 : 0;
    }
    if (rank >= maxN) continue;
    ++nn;
    // Agent::computeNewVelocity, one ORCA line
    const float rvx = velx - tvx[j], rvy = vely - tvy[j];       // relativeVelocity
    const float distSq = rpx * rpx + rpy * rpy;
    const float combinedRadius = radius + trad[j];
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
    L.set(rank, velx + 0.5f * ux, vely + 0.5f * uy, dirx, diry);
  }
  float rx, ry;
  const int lineFail = lp2(L, nn, maxSpeed, prefx, prefy, false, rx, ry);
  if (lineFail < nn) lp3(L, P, nn, lineFail, maxSpeed, rx, ry);
  out_x = rx;
  out_y = ry;
}

// ORCA.predict's Python-side preferred velocity (simulator/policy/orca.py:136-140)
__device__ __forceinline__ void orca_pref_velocity(double px, double py, double gx, double gy,
                                                   float &prefx, float &prefy) {
  double dx = gx - px, dy = gy - py;
  double speed = norm2(dx, dy);
  if (speed > 1) {
    prefx = (float)(dx / speed);
    prefy = (float)(dy / speed);
  } else {
    prefx = (float)dx;
    prefy = (float)dy;
  }
}

// rotate(): rl/policy/cadrl.py:236-337.  The robot-side terms depend only on the robot
// state, so they are computed once (RotFrame) and shared by all rows of that robot.
struct RotFrame {
  float px, py, r, vpref, c, s, dg, vx, vy, theta;
};

__device__ __forceinline__ RotFrame rot_frame(const double *rb, int rotate_unicycle) {
  RotFrame f;
  f.px = (float)rb[0];
  f.py = (float)rb[1];
  const float vx = (float)rb[2], vy = (float)rb[3];
  f.r = (float)rb[4];
  const float gx = (float)rb[5], gy = (float)rb[6];
  f.vpref = (float)rb[7];
  const float dx = gx - f.px, dy = gy - f.py;
  const float rot = atan2f(dy, dx);
  f.dg = sqrtf(dx * dx + dy * dy);
  f.c = cosf(rot);
  f.s = sinf(rot);
  f.vx = vx * f.c + vy * f.s;
  f.vy = vy * f.c - vx * f.s;
  f.theta = rotate_unicycle ? ((float)rb[8] - rot) : 0.0f;
  return f;
}

template <int T>
__device__ __forceinline__ void rotate_row(const RotFrame &f, double opx, double opy, double ovx,
                                           double ovy, double orad, int otype, float *out) {
  const float px1 = (float)opx, py1 = (float)opy, vx1 = (float)ovx, vy1 = (float)ovy;
  const float r1 = (float)orad;
  out[0] = f.dg;
  out[1] = f.vpref;
  out[2] = f.theta;
  out[3] = f.r;
  out[4] = f.vx;
  out[5] = f.vy;
  out[6] = (px1 - f.px) * f.c + (py1 - f.py) * f.s;
  out[7] = (py1 - f.py) * f.c - (px1 - f.px) * f.s;
  out[8] = vx1 * f.c + vy1 * f.s;
  out[9] = vy1 * f.c - vx1 * f.s;
  out[10] = r1;
  const float ex = f.px - px1, ey = f.py - py1;
  out[11] = sqrtf(ex * ex + ey * ey);
  out[12] = f.r + r1;
  if (T == 17) {
#pragma unroll
    for (int k = 0; k < 4; ++k) out[13 + k] = (otype == k) ? 1.0f : 0.0f;
  }
}

// Python float % for positive divisor (agent.py:214)
__device__ __forceinline__ double py_mod(double a, double b) {
  double m = fmod(a, b);
  if (m != 0) {
    if ((b < 0) != (m < 0)) m += b;
  } else {
    m = copysign(0.0, b);
  }
  return m;
}

}  // namespace ebc
