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
        const int w = (int)(ey - sy);
        const unsigned __int128 mask = (((unsigned __int128)1 << w) - 1) << sy;
        const uint64_t lo = (uint64_t)mask, hi = (uint64_t)(mask >> 64);
        // rows [sx, ex): the first eight are loaded together (robot radius <= 0.56 m at 0.1 m
        // cells never needs more), the rest in a plain loop
        uint64_t acc = 0;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const long x = sx + r < ex ? sx + r : sx;
          acc |= (grid[x * 2] & lo) | (grid[x * 2 + 1] & hi);
        }
        for (long x = sx + 8; x < ex; ++x) acc |= (grid[x * 2] & lo) | (grid[x * 2 + 1] & hi);
        collision = acc != 0;
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

#define RVO_EPS 0.00001f  // RVO_EPSILON

__device__ __forceinline__ float det2(float ax, float ay, float bx, float by) { return ax * by - ay * bx; }

// ORCA.predict's Python-side preferred velocity (simulator/policy/orca.py:136-140)
// (dx, dy) = goal - position, speed = norm2(dx, dy)
__device__ __forceinline__ void orca_pref_from(double dx, double dy, double speed, float &prefx, float &prefy) {
  if (speed > 1) {
    prefx = (float)(dx / speed);
    prefy = (float)(dy / speed);
  } else {
    prefx = (float)dx;
    prefy = (float)dy;
  }
}
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
