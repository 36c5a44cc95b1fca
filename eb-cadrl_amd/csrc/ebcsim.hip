// ebcsim.hip — kernels and C ABI of libebcsim.so (gfx950 / MI355X).
//
// Kernels
//   step_kernel<POLICY, T>      one env.step(update=True) for every env
//                                (simulator/env.py:388-466), one lane per human,
//                                one wave per workgroup, floor(64 / N) envs per wave.
//   policy_kernel<POLICY>       human velocities only (feeds look-ahead and CACHED steps)
//   lookahead_kernel<T>         the |A|-way onestep_lookahead sweep
//                                (rl/policy/multi_human_rl.py:38-61), one wave per env.
// Layout in HBM: struct-of-arrays doubles [E][N] per human field (lane = e*N + i, so a
// wave's loads are contiguous), robot [E][9], grid [E][G][2] 64-bit words.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "ebc_device.h"
#include "ebc_orca_group.h"

namespace {

struct DevState {
  int E, N, S, G;
  int *n_humans;
  double *px, *py, *vx, *vy, *gx, *gy, *radius, *v_pref;
  uint8_t *type;
  int *n_static;
  double *spx, *spy, *sradius;
  uint64_t *grid;  // nullptr when every map is free
  double *robot;
  double *time;
  double *arrival;
  uint8_t *done;
  double *hact;  // [E][N][2]
  double *px0, *py0, *vx0, *vy0, *robot0;
};

struct StepIO {
  const double *robot_action;
  double border[4];
  int has_border;
  int robot_policy;
  int auto_reset;
  double *reward;
  uint8_t *done;
  uint8_t *info;
  double *dmin;
  double *dist_to_goal;
  double *robot_action_out;
  double *human_action;
  double *ob;
  float *obs_rotated;
};

struct LookIO {
  const double *actions;
  int A;
  double border[4];
  int has_border;
  double *reward;
  uint8_t *done;
  uint8_t *info;
  double *dmin;
  double *next_ob;
  float *rows;
};

// Human velocity for lane `i` of env-slot `el`.  The per-wave LDS tile holds, for each
// env of the wave, N human slots followed by one robot slot (stride N + 1).
template <int POLICY>
__device__ __forceinline__ void human_policy(const EbcParams &p, const DevState &s, bool active,
                                             int e, int i, int n, int tile_base, double px, double py,
                                             double gx, double gy, double v_pref, const float *tpx,
                                             const float *tpy, const float *tvx, const float *tvy,
                                             const float *trad, ebc::LineSet &L, ebc::LineSet &P,
                                             double &ax, double &ay) {
  ax = 0;
  ay = 0;
  if (!active) return;
  if (POLICY == EBC_HUMAN_EXTERNAL || POLICY == EBC_HUMAN_CACHED) {
    ax = s.hact[((size_t)e * s.N + i) * 2];
    ay = s.hact[((size_t)e * s.N + i) * 2 + 1];
  } else if (POLICY == EBC_HUMAN_LINEAR) {
    ebc::linear_policy(px, py, gx, gy, v_pref, ax, ay);
  } else {
    float prefx, prefy, ox, oy;
    ebc::orca_pref_velocity(px, py, gx, gy, prefx, prefy);
    // others in ob order: humans 0..n-1 (self skipped) then the robot slot (index n) if visible.
    // The robot slot sits at tile index N; when n < N it is addressed through a remap below.
    const int n_agents = n + (p.robot_visible ? 1 : 0);
    ebc::orca_velocity(p, i, n_agents, tpx + tile_base, tpy + tile_base, tvx + tile_base,
                       tvy + tile_base, trad + tile_base, (float)v_pref, prefx, prefy, L, P, ox, oy);
    ax = (double)ox;  // getAgentVelocity -> Python float
    ay = (double)oy;
  }
}

template <int POLICY, int T>
__global__ __launch_bounds__(EBC_WAVE) void step_kernel(EbcParams p, DevState s, StepIO io) {
  __shared__ float tpx[EBC_TILE_SLOTS], tpy[EBC_TILE_SLOTS], tvx[EBC_TILE_SLOTS],
      tvy[EBC_TILE_SLOTS], trad[EBC_TILE_SLOTS];
  __shared__ double sh_d[EBC_WAVE];
  __shared__ double sh_ract[EBC_WAVE][2];
  __shared__ double sh_robot[EBC_WAVE][9];
  __shared__ float lines[POLICY == EBC_HUMAN_ORCA ? EBC_MAXNB * 4 * EBC_WAVE : 1];
  __shared__ float plines[POLICY == EBC_HUMAN_ORCA ? EBC_MAXNB * 4 * EBC_WAVE : 1];

  const int N = s.N, S = s.S, R = N + S;
  const int lane = threadIdx.x;
  const int epb = EBC_WAVE / N;  // envs per wave
  const int el = lane / N;
  const int i = lane - el * N;
  const int e = blockIdx.x * epb + el;
  const bool env_ok = el < epb && e < s.E;
  const int n = env_ok ? s.n_humans[e] : 0;
  const bool active = env_ok && i < n;
  const bool leader = env_ok && i == 0;
  const size_t k = (size_t)(env_ok ? e : 0) * N + i;
  const double dt = p.time_step;

  const bool restart = env_ok && io.auto_reset && s.done[e];
  const double *rb_in = (restart ? s.robot0 : s.robot) + (size_t)(env_ok ? e : 0) * 9;
  double rb[9];
#pragma unroll
  for (int c = 0; c < 9; ++c) rb[c] = env_ok ? rb_in[c] : 0.0;
  const double gtime = (env_ok && !restart) ? s.time[e] : 0.0;

  double px = 0, py = 0, vx = 0, vy = 0, gx = 0, gy = 0, rad = 0, vpref = 0, arrival = 0;
  int type = 0;
  if (active) {
    px = restart ? s.px0[k] : s.px[k];
    py = restart ? s.py0[k] : s.py[k];
    vx = restart ? s.vx0[k] : s.vx[k];
    vy = restart ? s.vy0[k] : s.vy[k];
    gx = s.gx[k];
    gy = s.gy[k];
    rad = s.radius[k];
    vpref = s.v_pref[k];
    type = s.type[k];
    arrival = restart ? 0.0 : s.arrival[k];
  }

  // --- stage the wave's envs in LDS as the floats rvo2 holds (orca.py:110-133)
  const int tile_base = el * (N + 1);
  if (POLICY == EBC_HUMAN_ORCA) {
    if (active) {
      tpx[tile_base + i] = (float)px;
      tpy[tile_base + i] = (float)py;
      tvx[tile_base + i] = (float)vx;
      tvy[tile_base + i] = (float)vy;
      trad[tile_base + i] = (float)(rad + 0.01 + p.orca_safety_space);
    }
    if (leader) {  // robot as the last "other" (env.py:401-402): slot n
      tpx[tile_base + n] = (float)rb[0];
      tpy[tile_base + n] = (float)rb[1];
      tvx[tile_base + n] = (float)rb[2];
      tvy[tile_base + n] = (float)rb[3];
      trad[tile_base + n] = (float)(rb[4] + 0.01 + p.orca_safety_space);
    }
  }
  // --- robot action (leader), shared through LDS
  if (leader) {
    double a0, a1;
    if (io.robot_policy == EBC_ROBOT_LINEAR) {
      ebc::linear_policy(rb[0], rb[1], rb[5], rb[6], rb[7], a0, a1);
    } else {
      a0 = io.robot_action[2 * (size_t)e];
      a1 = io.robot_action[2 * (size_t)e + 1];
    }
    sh_ract[el][0] = a0;
    sh_ract[el][1] = a1;
  }
  __syncthreads();

  // --- humans choose their velocity on the pre-step state (env.py:393-405)
  ebc::LineSet L{lines, lane}, P{plines, lane};
  double ax, ay;
  human_policy<POLICY>(p, s, active, e, i, n, tile_base, px, py, gx, gy, vpref, tpx, tpy, tvx, tvy,
                       trad, L, P, ax, ay);

  // --- swept robot-human distance with the humans' CURRENT velocity (collisions.py:35-42)
  const double a0 = sh_ract[env_ok ? el : 0][0], a1 = sh_ract[env_ok ? el : 0][1];
  double rvx, rvy;
  if (p.robot_kinematics == EBC_HOLONOMIC) {
    rvx = a0;
    rvy = a1;
  } else {
    rvx = a0 * cos(a1 + rb[8]);
    rvy = a0 * sin(a1 + rb[8]);
  }
  sh_d[lane] = active ? ebc::closest_dist(px, py, vx, vy, rad, rb[0], rb[1], rb[4], rvx, rvy, dt) : 0.0;
  // type rides along in the tile of the next phase: reuse tvx slot? keep it simple: own array
  __shared__ uint8_t sh_type[EBC_WAVE];
  sh_type[lane] = (uint8_t)type;
  __syncthreads();

  // --- leader: ordered per-type reduction with break at first hit (env.py:303-338),
  //     grid window, reward, robot update
  if (leader) {
    double dmin[3] = {INFINITY, INFINITY, INFINITY};
    int coll[4] = {0, 0, 0, 0};
    for (int j = 0; j < n; ++j) {
      const int t = sh_type[lane + j];
      const double d = sh_d[lane + j];
      if (t < 3 && !coll[t]) {
        if (d < 0)
          coll[t] = 1;
        else if (d < dmin[t])
          dmin[t] = d;
      }
    }
    double nx, ny;
    ebc::robot_next_position(rb, p.robot_kinematics, a0, a1, dt, nx, ny);
    coll[3] = ebc::grid_collision(s.grid ? s.grid + (size_t)e * s.G * 2 : nullptr, s.G, p.map_size_m,
                                  p.map_resolution, nx, ny, rb[4], io.has_border ? io.border : nullptr);
    const ebc::RewardOut ro = ebc::reward_compute(p, nx, ny, rb[5], rb[6], rb[4], a1, gtime, dmin, coll);
    // Agent.step for the robot (agent.py:202-228)
    rb[0] = nx;
    rb[1] = ny;
    if (p.robot_kinematics == EBC_HOLONOMIC) {
      rb[2] = a0;
      rb[3] = a1;
    } else {
      rb[8] = ebc::py_mod(rb[8] + a1, 2 * M_PI);
      rb[2] = a0 * cos(rb[8]);
      rb[3] = a0 * sin(rb[8]);
    }
    double *rb_out = s.robot + (size_t)e * 9;
#pragma unroll
    for (int c = 0; c < 9; ++c) {
      rb_out[c] = rb[c];
      sh_robot[el][c] = rb[c];
    }
    s.time[e] = gtime + dt;
    s.done[e] = (uint8_t)ro.done;
    if (io.reward) io.reward[e] = ro.reward;
    if (io.done) io.done[e] = (uint8_t)ro.done;
    if (io.info) io.info[e] = (uint8_t)ro.info;
    if (io.dmin) {
      io.dmin[3 * (size_t)e] = dmin[0];
      io.dmin[3 * (size_t)e + 1] = dmin[1];
      io.dmin[3 * (size_t)e + 2] = dmin[2];
    }
    if (io.dist_to_goal) io.dist_to_goal[e] = ro.dist_to_goal;
    if (io.robot_action_out) {
      io.robot_action_out[2 * (size_t)e] = a0;
      io.robot_action_out[2 * (size_t)e + 1] = a1;
    }
  }
  __syncthreads();

  // --- humans move (agent.py:202-211), first-arrival times (env.py:365-378)
  if (active) {
    px = px + ax * dt;
    py = py + ay * dt;
    vx = ax;
    vy = ay;
    const double tnew = gtime + dt;
    if (arrival == 0 && ebc::norm2(px - gx, py - gy) < rad) arrival = tnew;
    s.px[k] = px;
    s.py[k] = py;
    s.vx[k] = vx;
    s.vy[k] = vy;
    s.arrival[k] = arrival;
  }
  if (env_ok && io.human_action) {
    io.human_action[k * 2] = active ? ax : 0.0;
    io.human_action[k * 2 + 1] = active ? ay : 0.0;
  }

  // --- returned observation: humans then static rows (env.py:381-382, :457-458), raw and
  //     rotated into the robot frame (cadrl.py:236-337)
  if (env_ok && (io.ob || io.obs_rotated)) {
    const int ns = S ? s.n_static[e] : 0;
    const ebc::RotFrame f = ebc::rot_frame(sh_robot[el], p.rotate_unicycle);
    for (int r = i; r < R; r += N) {
      double opx = 0, opy = 0, ovx = 0, ovy = 0, orad = 0;
      int otype = 0;
      bool valid = false;
      if (r < n) {  // r == i: this lane's own human
        opx = px; opy = py; ovx = vx; ovy = vy; orad = rad; otype = type;
        valid = true;
      } else if (r - n < ns) {
        const size_t q = (size_t)e * S + (r - n);
        opx = s.spx[q]; opy = s.spy[q]; orad = s.sradius[q]; otype = EBC_ADULT_STATIC;
        valid = true;
      }
      const size_t row = (size_t)e * R + r;
      if (io.ob) {
        double *o = io.ob + row * 5;
        o[0] = opx; o[1] = opy; o[2] = ovx; o[3] = ovy; o[4] = orad;
      }
      if (io.obs_rotated) {
        float out[T];
        if (valid) {
          ebc::rotate_row<T>(f, opx, opy, ovx, ovy, orad, otype, out);
        } else {
#pragma unroll
          for (int c = 0; c < T; ++c) out[c] = 0.0f;
        }
        float *o = io.obs_rotated + row * T;
#pragma unroll
        for (int c = 0; c < T; ++c) o[c] = out[c];
      }
    }
  }
}

// Human velocities only -> s.hact (look-ahead, CACHED steps).  Same mapping as step_kernel.
template <int POLICY>
__global__ __launch_bounds__(EBC_WAVE) void policy_kernel(EbcParams p, DevState s) {
  __shared__ float tpx[EBC_TILE_SLOTS], tpy[EBC_TILE_SLOTS], tvx[EBC_TILE_SLOTS],
      tvy[EBC_TILE_SLOTS], trad[EBC_TILE_SLOTS];
  __shared__ float lines[POLICY == EBC_HUMAN_ORCA ? EBC_MAXNB * 4 * EBC_WAVE : 1];
  __shared__ float plines[POLICY == EBC_HUMAN_ORCA ? EBC_MAXNB * 4 * EBC_WAVE : 1];
  const int N = s.N;
  const int lane = threadIdx.x;
  const int epb = EBC_WAVE / N;
  const int el = lane / N;
  const int i = lane - el * N;
  const int e = blockIdx.x * epb + el;
  const bool env_ok = el < epb && e < s.E;
  const int n = env_ok ? s.n_humans[e] : 0;
  const bool active = env_ok && i < n;
  const size_t k = (size_t)(env_ok ? e : 0) * N + i;
  double px = 0, py = 0, gx = 0, gy = 0, vpref = 0;
  const int tile_base = el * (N + 1);
  if (active) {
    px = s.px[k];
    py = s.py[k];
    gx = s.gx[k];
    gy = s.gy[k];
    vpref = s.v_pref[k];
    if (POLICY == EBC_HUMAN_ORCA) {
      tpx[tile_base + i] = (float)px;
      tpy[tile_base + i] = (float)py;
      tvx[tile_base + i] = (float)s.vx[k];
      tvy[tile_base + i] = (float)s.vy[k];
      trad[tile_base + i] = (float)(s.radius[k] + 0.01 + p.orca_safety_space);
    }
  }
  if (POLICY == EBC_HUMAN_ORCA && env_ok && i == 0) {
    const double *rb = s.robot + (size_t)e * 9;
    tpx[tile_base + n] = (float)rb[0];
    tpy[tile_base + n] = (float)rb[1];
    tvx[tile_base + n] = (float)rb[2];
    tvy[tile_base + n] = (float)rb[3];
    trad[tile_base + n] = (float)(rb[4] + 0.01 + p.orca_safety_space);
  }
  __syncthreads();
  ebc::LineSet L{lines, lane}, P{plines, lane};
  double ax, ay;
  human_policy<POLICY>(p, s, active, e, i, n, tile_base, px, py, gx, gy, vpref, tpx, tpy, tvx, tvy,
                       trad, L, P, ax, ay);
  if (env_ok) {
    s.hact[k * 2] = active ? ax : 0.0;
    s.hact[k * 2 + 1] = active ? ay : 0.0;
  }
}

// ORCA with a GS-lane group per human (ebc_orca_group.h) -> s.hact.  64 / GS humans per wave;
// lane j of a group loads "other" j of its human in ob order (env.py:396-402): the humans
// before and after it, then the robot when it is visible.
template <int GS>
__global__ __launch_bounds__(EBC_WAVE) void orca_group_kernel(EbcParams p, DevState s, int auto_reset) {
  constexpr int HPW = EBC_WAVE / GS;
  __shared__ __align__(16) float dist_lds[EBC_WAVE];
  __shared__ float4 lines_lds[EBC_WAVE];
  __shared__ float4 proj_lds[EBC_WAVE];
  const int N = s.N;
  const int lane = threadIdx.x;
  const int group = lane / GS;
  const int j = lane - group * GS;
  const long h = (long)blockIdx.x * HPW + group;
  const bool h_ok = h < (long)s.E * N;
  const int e = h_ok ? (int)(h / N) : 0;
  const int i = h_ok ? (int)(h - (long)e * N) : 0;
  const int n = h_ok ? s.n_humans[e] : 0;
  const bool human_ok = h_ok && i < n;
  const bool restart = h_ok && auto_reset && s.done[e];
  const double *PX = restart ? s.px0 : s.px, *PY = restart ? s.py0 : s.py;
  const double *VX = restart ? s.vx0 : s.vx, *VY = restart ? s.vy0 : s.vy;
  const double *RB = (restart ? s.robot0 : s.robot) + (size_t)e * 9;
  const size_t base = (size_t)e * N;

  float posx = 0, posy = 0, velx = 0, vely = 0, radius = 0, maxSpeed = 0, prefx = 0, prefy = 0;
  if (human_ok) {
    const size_t k = base + i;
    const double px = PX[k], py = PY[k];
    posx = (float)px;
    posy = (float)py;
    velx = (float)VX[k];
    vely = (float)VY[k];
    radius = (float)(s.radius[k] + 0.01 + p.orca_safety_space);  // orca.py:116
    maxSpeed = (float)s.v_pref[k];                                // orca.py:117
    ebc::orca_pref_velocity(px, py, s.gx[k], s.gy[k], prefx, prefy);
  }
  const int n_others = human_ok ? (n - 1 + (p.robot_visible ? 1 : 0)) : 0;
  const bool valid = j < n_others;
  float opx = 0, opy = 0, ovx = 0, ovy = 0, orad = 0;
  if (valid) {
    if (j < n - 1) {
      const size_t k = base + (j < i ? j : j + 1);
      opx = (float)PX[k];
      opy = (float)PY[k];
      ovx = (float)VX[k];
      ovy = (float)VY[k];
      orad = (float)(s.radius[k] + 0.01 + p.orca_safety_space);  // orca.py:122-126
    } else {  // the robot's observable state, last in ob (env.py:401-402)
      opx = (float)RB[0];
      opy = (float)RB[1];
      ovx = (float)RB[2];
      ovy = (float)RB[3];
      orad = (float)(RB[4] + 0.01 + p.orca_safety_space);
    }
  }
  float ox, oy;
  ebc::orca_group<GS>(p, j, group, valid, posx, posy, velx, vely, radius, maxSpeed, prefx, prefy, opx,
                      opy, ovx, ovy, orad, dist_lds + group * GS, lines_lds + group * GS,
                      proj_lds + group * GS, ox, oy);
  if (h_ok && j == 0) {
    s.hact[(size_t)h * 2] = human_ok ? (double)ox : 0.0;  // getAgentVelocity -> Python float
    s.hact[(size_t)h * 2 + 1] = human_ok ? (double)oy : 0.0;
  }
}

// Look-ahead: one wave per env.  Human velocities come from s.hact (policy_kernel ran).
//   phase A  lanes over humans/static rows: next observable rows into LDS
//   phase B  lanes over actions: collisions (ordered, serial over humans), grid, reward,
//            and the robot-frame terms of rotate() for that action
//   phase C  lanes over (action, row): rotated rows
#define EBC_LA_MAX_ROWS 128
#define EBC_LA_MAX_ACTIONS 128
template <int T>
__global__ __launch_bounds__(EBC_WAVE) void lookahead_kernel(EbcParams p, DevState s, LookIO io) {
  __shared__ double hpx[EBC_WAVE], hpy[EBC_WAVE], hvx[EBC_WAVE], hvy[EBC_WAVE], hrad[EBC_WAVE];
  __shared__ uint8_t htype[EBC_WAVE];
  __shared__ double row[EBC_LA_MAX_ROWS][5];
  __shared__ uint8_t row_type[EBC_LA_MAX_ROWS];
  __shared__ ebc::RotFrame frames[EBC_LA_MAX_ACTIONS];
  const int N = s.N, S = s.S, R = N + S, A = io.A;
  const int lane = threadIdx.x;
  const int e = blockIdx.x;
  const int n = s.n_humans[e];
  const int ns = S ? s.n_static[e] : 0;
  const double dt = p.time_step;
  const double *rbp = s.robot + (size_t)e * 9;
  double rb[9];
#pragma unroll
  for (int c = 0; c < 9; ++c) rb[c] = rbp[c];

  // phase A: get_next_observable_state (agent.py:80-93), static rows appended (env.py:457-458)
  for (int r = lane; r < R; r += EBC_WAVE) {
    double o[5] = {0, 0, 0, 0, 0};
    int t = 0;
    if (r < n) {
      const size_t k = (size_t)e * N + r;
      const double ax = s.hact[k * 2], ay = s.hact[k * 2 + 1];
      hpx[r] = s.px[k]; hpy[r] = s.py[k]; hvx[r] = s.vx[k]; hvy[r] = s.vy[k];
      hrad[r] = s.radius[k]; htype[r] = s.type[k];
      o[0] = hpx[r] + ax * dt;
      o[1] = hpy[r] + ay * dt;
      o[2] = ax;
      o[3] = ay;
      o[4] = hrad[r];
      t = htype[r];
    } else if (r - n < ns) {
      const size_t q = (size_t)e * S + (r - n);
      o[0] = s.spx[q]; o[1] = s.spy[q]; o[4] = s.sradius[q];
      t = EBC_ADULT_STATIC;
    }
#pragma unroll
    for (int c = 0; c < 5; ++c) row[r][c] = o[c];
    row_type[r] = (uint8_t)t;
    if (io.next_ob) {
      double *dst = io.next_ob + ((size_t)e * R + r) * 5;
#pragma unroll
      for (int c = 0; c < 5; ++c) dst[c] = o[c];
    }
  }
  __syncthreads();

  // phase B
  const double gtime = s.time[e];
  for (int a = lane; a < A; a += EBC_WAVE) {
    const double a0 = io.actions[2 * a], a1 = io.actions[2 * a + 1];
    double rvx, rvy;
    if (p.robot_kinematics == EBC_HOLONOMIC) {
      rvx = a0;
      rvy = a1;
    } else {
      rvx = a0 * cos(a1 + rb[8]);
      rvy = a0 * sin(a1 + rb[8]);
    }
    double dmin[3] = {INFINITY, INFINITY, INFINITY};
    int coll[4] = {0, 0, 0, 0};
    for (int j = 0; j < n; ++j) {
      const int t = htype[j];
      if (t < 3 && !coll[t]) {
        const double d = ebc::closest_dist(hpx[j], hpy[j], hvx[j], hvy[j], hrad[j], rb[0], rb[1],
                                           rb[4], rvx, rvy, dt);
        if (d < 0)
          coll[t] = 1;
        else if (d < dmin[t])
          dmin[t] = d;
      }
    }
    double nx, ny;
    ebc::robot_next_position(rb, p.robot_kinematics, a0, a1, dt, nx, ny);
    coll[3] = ebc::grid_collision(s.grid ? s.grid + (size_t)e * s.G * 2 : nullptr, s.G, p.map_size_m,
                                  p.map_resolution, nx, ny, rb[4], io.has_border ? io.border : nullptr);
    const ebc::RewardOut ro = ebc::reward_compute(p, nx, ny, rb[5], rb[6], rb[4], a1, gtime, dmin, coll);
    const size_t o = (size_t)e * A + a;
    if (io.reward) io.reward[o] = ro.reward;
    if (io.done) io.done[o] = (uint8_t)ro.done;
    if (io.info) io.info[o] = (uint8_t)ro.info;
    if (io.dmin) {
      io.dmin[3 * o] = dmin[0];
      io.dmin[3 * o + 1] = dmin[1];
      io.dmin[3 * o + 2] = dmin[2];
    }
    if (io.rows) {
      // CADRL.propagate for the robot (cadrl.py:118-165)
      double nb[9];
#pragma unroll
      for (int c = 0; c < 9; ++c) nb[c] = rb[c];
      if (p.robot_kinematics == EBC_HOLONOMIC) {
        nb[0] = rb[0] + a0 * dt;
        nb[1] = rb[1] + a1 * dt;
        nb[2] = a0;
        nb[3] = a1;
      } else {
        const double nth = rb[8] + a1;
        const double nvx = a0 * cos(nth), nvy = a0 * sin(nth);
        nb[0] = rb[0] + nvx * dt;
        nb[1] = rb[1] + nvy * dt;
        nb[2] = nvx;
        nb[3] = nvy;
        nb[8] = nth;
      }
      frames[a] = ebc::rot_frame(nb, p.rotate_unicycle);
    }
  }
  if (!io.rows) return;
  __syncthreads();

  // phase C: rows_rotated[e][a][r][:]
  const int total = A * R;
  for (int idx = lane; idx < total; idx += EBC_WAVE) {
    const int a = idx / R, r = idx - a * R;
    float out[T];
    if (r < n + ns) {
      ebc::rotate_row<T>(frames[a], row[r][0], row[r][1], row[r][2], row[r][3], row[r][4],
                         row_type[r], out);
    } else {
#pragma unroll
      for (int c = 0; c < T; ++c) out[c] = 0.0f;
    }
    float *dst = io.rows + ((size_t)e * A * R + idx) * T;
#pragma unroll
    for (int c = 0; c < T; ++c) dst[c] = out[c];
  }
}

// ------------------------------------------------------------------------------ host
thread_local std::string g_err;

int fail(int code, const std::string &msg) {
  g_err = msg;
  return code;
}

#define HIP_TRY(x)                                                                          \
  do {                                                                                      \
    hipError_t err__ = (x);                                                                 \
    if (err__ != hipSuccess)                                                                \
      return fail(EBC_ERR_DEVICE, std::string(#x) + ": " + hipGetErrorString(err__));       \
  } while (0)

struct Handle {
  int device = 0;
  EbcParams p;
  DevState s;
  int T = 13;
  std::vector<void *> allocs;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  bool has_reset = false;
  bool has_grid = false;
  int orca_gs = 0;  // lanes per human of the ORCA kernel; 0 = one lane per human, fused in step_kernel
  uint64_t *grid_alloc = nullptr;
  // staging for host-location calls
  void *stage = nullptr;
  size_t stage_bytes = 0;
  // timing
  bool timing = false;
  std::vector<hipEvent_t> ev;
  size_t ev_used = 0;
  double ms_sum = 0;
  int64_t launches = 0;
};

template <typename Tp>
int dev_alloc(Handle *h, Tp **out, size_t count) {
  void *ptr = nullptr;
  HIP_TRY(hipMalloc(&ptr, count * sizeof(Tp) + 16));
  HIP_TRY(hipMemset(ptr, 0, count * sizeof(Tp) + 16));
  h->allocs.push_back(ptr);
  *out = (Tp *)ptr;
  return EBC_OK;
}

int ensure_stage(Handle *h, size_t bytes) {
  if (bytes <= h->stage_bytes) return EBC_OK;
  if (h->stage) HIP_TRY(hipFree(h->stage));
  h->stage = nullptr;
  h->stage_bytes = 0;
  HIP_TRY(hipMalloc(&h->stage, bytes));
  h->stage_bytes = bytes;
  return EBC_OK;
}

// A bump allocator over the staging buffer for one host-location call.
struct Stager {
  Handle *h;
  size_t off = 0;
  struct Out { void *host; void *dev; size_t bytes; };
  std::vector<Out> outs;
  template <typename Tp>
  Tp *out(Tp *host, size_t count) {
    if (!host) return nullptr;
    Tp *d = (Tp *)((char *)h->stage + off);
    outs.push_back({(void *)host, (void *)d, count * sizeof(Tp)});
    off += (count * sizeof(Tp) + 255) & ~(size_t)255;
    return d;
  }
  template <typename Tp>
  int in(const Tp *host, size_t count, const Tp **dev) {
    *dev = nullptr;
    if (!host) return EBC_OK;
    Tp *d = (Tp *)((char *)h->stage + off);
    off += (count * sizeof(Tp) + 255) & ~(size_t)255;
    HIP_TRY(hipMemcpyAsync(d, host, count * sizeof(Tp), hipMemcpyHostToDevice, h->stream));
    *dev = d;
    return EBC_OK;
  }
  int finish() {
    for (auto &o : outs)
      HIP_TRY(hipMemcpyAsync(o.host, o.dev, o.bytes, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return EBC_OK;
  }
};

size_t pad256(size_t b) { return (b + 255) & ~(size_t)255; }

int check_handle(void *handle, Handle **out) {
  if (!handle) return fail(EBC_ERR_INVALID, "null handle");
  *out = (Handle *)handle;
  hipError_t e = hipSetDevice((*out)->device);
  if (e != hipSuccess) return fail(EBC_ERR_DEVICE, std::string("hipSetDevice: ") + hipGetErrorString(e));
  return EBC_OK;
}

template <int POLICY>
void launch_step_T(Handle *h, const StepIO &io, int blocks) {
  if (h->T == 17)
    hipLaunchKernelGGL((step_kernel<POLICY, 17>), dim3(blocks), dim3(EBC_WAVE), 0, h->stream, h->p, h->s, io);
  else
    hipLaunchKernelGGL((step_kernel<POLICY, 13>), dim3(blocks), dim3(EBC_WAVE), 0, h->stream, h->p, h->s, io);
}

int launch_orca_group(Handle *h, int auto_reset) {
  const long humans = (long)h->s.E * h->s.N;
  const int hpw = EBC_WAVE / h->orca_gs;
  const int blocks = (int)((humans + hpw - 1) / hpw);
  if (h->orca_gs == 8)
    hipLaunchKernelGGL((orca_group_kernel<8>), dim3(blocks), dim3(EBC_WAVE), 0, h->stream, h->p, h->s, auto_reset);
  else if (h->orca_gs == 16)
    hipLaunchKernelGGL((orca_group_kernel<16>), dim3(blocks), dim3(EBC_WAVE), 0, h->stream, h->p, h->s, auto_reset);
  else
    hipLaunchKernelGGL((orca_group_kernel<32>), dim3(blocks), dim3(EBC_WAVE), 0, h->stream, h->p, h->s, auto_reset);
  HIP_TRY(hipGetLastError());
  return EBC_OK;
}

int launch_policy(Handle *h, int policy) {
  if (policy == EBC_HUMAN_ORCA && h->orca_gs) return launch_orca_group(h, 0);
  const int epb = EBC_WAVE / h->s.N;
  const int blocks = (h->s.E + epb - 1) / epb;
  if (policy == EBC_HUMAN_ORCA)
    hipLaunchKernelGGL((policy_kernel<EBC_HUMAN_ORCA>), dim3(blocks), dim3(EBC_WAVE), 0, h->stream, h->p, h->s);
  else if (policy == EBC_HUMAN_LINEAR)
    hipLaunchKernelGGL((policy_kernel<EBC_HUMAN_LINEAR>), dim3(blocks), dim3(EBC_WAVE), 0, h->stream, h->p, h->s);
  HIP_TRY(hipGetLastError());
  return EBC_OK;
}

}  // namespace

extern "C" {

int ebc_abi_version(void) { return EBC_ABI_VERSION; }

const char *ebc_last_error(void) { return g_err.c_str(); }

int ebc_params_default(EbcParams *p) {
  if (!p) return fail(EBC_ERR_INVALID, "null params");
  memset(p, 0, sizeof(*p));
  p->struct_size = sizeof(EbcParams);
  p->robot_kinematics = EBC_HOLONOMIC;
  p->time_step = 0.25;
  p->time_limit = 25;
  p->map_size_m = 9.0;
  p->map_resolution = 0.1;
  p->time_max = NAN;
  p->time_good = 10.0;
  p->max_goal_distance = NAN;
  p->success_reward = 1.0;
  for (int i = 0; i < 4; ++i) p->collision_penalty[i] = NAN;
  for (int i = 0; i < 3; ++i) {
    p->discomfort_dist[i] = 0.1;
    p->discomfort_factor[i] = 0.5;
  }
  p->orca_neighbor_dist = 10.0f;
  p->orca_time_horizon = 5.0f;
  p->orca_max_neighbors = 10;
  return EBC_OK;
}

int ebc_create(int device_id, int n_envs, int max_humans, int max_static, const EbcParams *params,
               void **handle_out) {
  if (!params || !handle_out) return fail(EBC_ERR_INVALID, "null argument");
  if (params->struct_size != sizeof(EbcParams))
    return fail(EBC_ERR_INVALID, "EbcParams.struct_size does not match this library");
  if (n_envs <= 0 || max_humans <= 0 || max_static < 0) return fail(EBC_ERR_INVALID, "bad dimensions");
  if (max_humans > EBC_WAVE - 1)
    return fail(EBC_ERR_UNSUPPORTED, "max_humans > 63 (one wave per scene group)");
  if (max_humans + max_static > EBC_LA_MAX_ROWS)
    return fail(EBC_ERR_UNSUPPORTED, "max_humans + max_static > 128 observation rows");
  if (params->orca_max_neighbors > EBC_MAXNB || params->orca_max_neighbors < 0)
    return fail(EBC_ERR_UNSUPPORTED, "orca_max_neighbors > 10");
  if (params->robot_kinematics != EBC_HOLONOMIC && params->robot_kinematics != EBC_UNICYCLE)
    return fail(EBC_ERR_INVALID, "robot_kinematics");
  const int G = (int)std::nearbyint(params->map_size_m / params->map_resolution);
  if (G <= 0 || G > 128) return fail(EBC_ERR_UNSUPPORTED, "occupancy grid wider than 128 cells");
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0)
    return fail(EBC_ERR_DEVICE, "no HIP device: libebcsim has no CPU fallback");
  if (device_id < 0 || device_id >= count) return fail(EBC_ERR_INVALID, "device_id out of range");
  HIP_TRY(hipSetDevice(device_id));
  Handle *h = new Handle();
  h->device = device_id;
  h->p = *params;
  h->T = params->with_agent_type ? 17 : 13;
  DevState &s = h->s;
  memset(&s, 0, sizeof(s));
  s.E = n_envs;
  s.N = max_humans;
  s.S = max_static;
  s.G = G;
  const size_t EN = (size_t)n_envs * max_humans, ES = (size_t)n_envs * (max_static ? max_static : 1);
  int rc = EBC_OK;
#define A_(field, cnt) if (rc == EBC_OK) rc = dev_alloc(h, &s.field, (cnt))
  A_(n_humans, n_envs); A_(px, EN); A_(py, EN); A_(vx, EN); A_(vy, EN); A_(gx, EN); A_(gy, EN);
  A_(radius, EN); A_(v_pref, EN); A_(type, EN); A_(n_static, n_envs); A_(spx, ES); A_(spy, ES);
  A_(sradius, ES); A_(robot, (size_t)n_envs * 9); A_(time, n_envs); A_(arrival, EN);
  A_(done, n_envs); A_(hact, EN * 2); A_(px0, EN); A_(py0, EN); A_(vx0, EN); A_(vy0, EN);
  A_(robot0, (size_t)n_envs * 9);
#undef A_
  if (rc == EBC_OK) rc = dev_alloc(h, &h->grid_alloc, (size_t)n_envs * G * 2);
  if (rc != EBC_OK) {
    for (void *ptr : h->allocs) (void)hipFree(ptr);
    delete h;
    return rc;
  }
  s.grid = nullptr;
  if (hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking) != hipSuccess) {
    for (void *ptr : h->allocs) (void)hipFree(ptr);
    delete h;
    return fail(EBC_ERR_DEVICE, "hipStreamCreate failed");
  }
  h->stream = h->own_stream;
  {
    // ORCA mapping: a group of 8 / 16 / 32 lanes per human when its "others" fit, else one lane
    // per human inside step_kernel.  EBCSIM_ORCA=fused forces the latter (A/B measurements).
    const int others = max_humans - 1 + (params->robot_visible ? 1 : 0);
    const char *force = getenv("EBCSIM_ORCA");
    h->orca_gs = others <= 8 ? 8 : others <= 16 ? 16 : others <= 32 ? 32 : 0;
    if (force && strcmp(force, "fused") == 0) h->orca_gs = 0;
  }
  *handle_out = h;
  return EBC_OK;
}

int ebc_destroy(void *handle) {
  Handle *h;
  int rc = check_handle(handle, &h);
  if (rc) return rc;
  (void)hipStreamSynchronize(h->stream);
  for (void *ptr : h->allocs) (void)hipFree(ptr);
  if (h->stage) (void)hipFree(h->stage);
  for (hipEvent_t ev : h->ev) (void)hipEventDestroy(ev);
  if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
  delete h;
  return EBC_OK;
}

int ebc_set_stream(void *handle, void *hip_stream) {
  Handle *h;
  int rc = check_handle(handle, &h);
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(h->stream));
  h->stream = (hipStream_t)hip_stream;  // NULL = the HIP null stream (torch's default stream)
  return EBC_OK;
}

int ebc_synchronize(void *handle) {
  Handle *h;
  int rc = check_handle(handle, &h);
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(h->stream));
  return EBC_OK;
}

int ebc_dims(void *handle, int32_t out[5]) {
  Handle *h;
  int rc = check_handle(handle, &h);
  if (rc) return rc;
  out[0] = h->s.E; out[1] = h->s.N; out[2] = h->s.S; out[3] = h->T; out[4] = h->s.G;
  return EBC_OK;
}

int ebc_reset(void *handle, const int32_t *env_ids, const EbcScene *sc) {
  Handle *h;
  int rc = check_handle(handle, &h);
  if (rc) return rc;
  if (!sc || sc->struct_size != sizeof(EbcScene)) return fail(EBC_ERR_INVALID, "EbcScene.struct_size");
  DevState &s = h->s;
  const int n = sc->n, N = s.N, S = s.S, G = s.G;
  if (n <= 0 || n > s.E) return fail(EBC_ERR_INVALID, "scene.n out of range");
  if (!sc->n_humans || !sc->px || !sc->py || !sc->vx || !sc->vy || !sc->gx || !sc->gy ||
      !sc->radius || !sc->v_pref || !sc->type || !sc->robot)
    return fail(EBC_ERR_INVALID, "scene array missing");
  if (S > 0 && (!sc->n_static || !sc->spx || !sc->spy || !sc->sradius))
    return fail(EBC_ERR_INVALID, "static rows missing");
  // validate on the host what the kernels assume
  for (int r = 0; r < n; ++r) {
    const int e = env_ids ? env_ids[r] : r;
    if (e < 0 || e >= s.E) return fail(EBC_ERR_INVALID, "env id out of range");
    if (sc->n_humans[r] < 0 || sc->n_humans[r] > N) return fail(EBC_ERR_INVALID, "n_humans > max_humans");
    if (S > 0 && (sc->n_static[r] < 0 || sc->n_static[r] > S))
      return fail(EBC_ERR_INVALID, "n_static > max_static");
    for (int i = 0; i < sc->n_humans[r]; ++i)
      if (sc->type[(size_t)r * N + i] > EBC_CHILD) return fail(EBC_ERR_INVALID, "human type must be 0..2");
  }
  HIP_TRY(hipStreamSynchronize(h->stream));
  const bool contiguous = [&] {
    if (!env_ids) return true;
    for (int r = 0; r < n; ++r)
      if (env_ids[r] != env_ids[0] + r) return false;
    return true;
  }();
  auto up = [&](void *dst_base, const void *src, size_t row_bytes) -> int {
    if (!src) return EBC_OK;
    if (contiguous) {
      const int e0 = env_ids ? env_ids[0] : 0;
      HIP_TRY(hipMemcpy((char *)dst_base + (size_t)e0 * row_bytes, src, (size_t)n * row_bytes,
                        hipMemcpyHostToDevice));
    } else {
      for (int r = 0; r < n; ++r)
        HIP_TRY(hipMemcpy((char *)dst_base + (size_t)env_ids[r] * row_bytes,
                          (const char *)src + (size_t)r * row_bytes, row_bytes, hipMemcpyHostToDevice));
    }
    return EBC_OK;
  };
  const size_t rowN = (size_t)N * sizeof(double);
#define UP_(dst, src, bytes) if ((rc = up((void *)(dst), (const void *)(src), (bytes))) != EBC_OK) return rc
  UP_(s.n_humans, sc->n_humans, sizeof(int));
  UP_(s.px, sc->px, rowN); UP_(s.py, sc->py, rowN); UP_(s.vx, sc->vx, rowN); UP_(s.vy, sc->vy, rowN);
  UP_(s.gx, sc->gx, rowN); UP_(s.gy, sc->gy, rowN); UP_(s.radius, sc->radius, rowN);
  UP_(s.v_pref, sc->v_pref, rowN); UP_(s.type, sc->type, (size_t)N);
  UP_(s.px0, sc->px, rowN); UP_(s.py0, sc->py, rowN); UP_(s.vx0, sc->vx, rowN); UP_(s.vy0, sc->vy, rowN);
  UP_(s.robot, sc->robot, 9 * sizeof(double)); UP_(s.robot0, sc->robot, 9 * sizeof(double));
  if (S > 0) {
    UP_(s.n_static, sc->n_static, sizeof(int));
    UP_(s.spx, sc->spx, (size_t)S * sizeof(double)); UP_(s.spy, sc->spy, (size_t)S * sizeof(double));
    UP_(s.sradius, sc->sradius, (size_t)S * sizeof(double));
  }
  const size_t grow = (size_t)G * 2 * sizeof(uint64_t);
  if (sc->grid) {
    UP_(h->grid_alloc, sc->grid, grow);
    h->has_grid = true;
    s.grid = h->grid_alloc;
  } else if (h->has_grid) {  // these envs get a free map
    std::vector<uint64_t> zeros((size_t)n * G * 2, 0);
    UP_(h->grid_alloc, zeros.data(), grow);
  }
#undef UP_
  // global_time = 0, arrival times = 0, done = 0 (env.py:149-151)
  for (int r = 0; r < n; ++r) {
    const int e = env_ids ? env_ids[r] : r;
    const int cnt = contiguous ? n : 1;
    HIP_TRY(hipMemset(s.time + e, 0, (size_t)cnt * sizeof(double)));
    HIP_TRY(hipMemset(s.arrival + (size_t)e * N, 0, (size_t)cnt * rowN));
    HIP_TRY(hipMemset(s.done + e, 0, (size_t)cnt));
    if (contiguous) break;
  }
  h->has_reset = true;
  return EBC_OK;
}

int ebc_set_human_actions(void *handle, int location, const double *act) {
  Handle *h;
  int rc = check_handle(handle, &h);
  if (rc) return rc;
  if (!act) return fail(EBC_ERR_INVALID, "null actions");
  const size_t bytes = (size_t)h->s.E * h->s.N * 2 * sizeof(double);
  HIP_TRY(hipMemcpyAsync(h->s.hact, act, bytes,
                         location == EBC_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice,
                         h->stream));
  if (location != EBC_DEVICE) HIP_TRY(hipStreamSynchronize(h->stream));
  return EBC_OK;
}

int ebc_step(void *handle, const EbcStepArgs *a) {
  Handle *h;
  int rc = check_handle(handle, &h);
  if (rc) return rc;
  if (!a || a->struct_size != sizeof(EbcStepArgs)) return fail(EBC_ERR_INVALID, "EbcStepArgs.struct_size");
  if (!h->has_reset) return fail(EBC_ERR_STATE, "ebc_step before ebc_reset");
  if (a->human_policy < EBC_HUMAN_EXTERNAL || a->human_policy > EBC_HUMAN_CACHED)
    return fail(EBC_ERR_INVALID, "human_policy");
  if (a->robot_policy == EBC_ROBOT_LINEAR && h->p.robot_kinematics != EBC_HOLONOMIC)
    return fail(EBC_ERR_UNSUPPORTED, "the linear robot policy is holonomic (simulator/policy/linear.py:11)");
  if (a->robot_policy == EBC_ROBOT_EXTERNAL && !a->robot_action)
    return fail(EBC_ERR_INVALID, "robot_action is NULL");
  if (a->robot_policy != EBC_ROBOT_EXTERNAL && a->robot_policy != EBC_ROBOT_LINEAR)
    return fail(EBC_ERR_INVALID, "robot_policy");
  if ((a->flags & EBC_FLAG_BORDER) && !a->border) return fail(EBC_ERR_INVALID, "border is NULL");
  const DevState &s = h->s;
  const size_t E = s.E, N = s.N, R = s.N + s.S, T = h->T;
  StepIO io;
  memset(&io, 0, sizeof(io));
  io.robot_policy = a->robot_policy;
  io.auto_reset = (a->flags & EBC_FLAG_AUTO_RESET) ? 1 : 0;
  io.has_border = (a->flags & EBC_FLAG_BORDER) ? 1 : 0;
  if (io.has_border) memcpy(io.border, a->border, sizeof(io.border));
  Stager st{h};
  if (a->location == EBC_DEVICE) {
    io.robot_action = a->robot_action;
    io.reward = a->reward; io.done = a->done; io.info = a->info; io.dmin = a->dmin;
    io.dist_to_goal = a->dist_to_goal; io.robot_action_out = a->robot_action_out;
    io.human_action = a->human_action; io.ob = a->ob; io.obs_rotated = a->obs_rotated;
  } else {
    const size_t need = pad256(E * 2 * 8) * 2 + pad256(E * 8) * 2 + pad256(E) * 2 + pad256(E * 3 * 8) +
                        pad256(E * N * 2 * 8) + pad256(E * R * 5 * 8) + pad256(E * R * T * 4) + 4096;
    if ((rc = ensure_stage(h, need)) != EBC_OK) return rc;
    if ((rc = st.in(a->robot_policy == EBC_ROBOT_EXTERNAL ? a->robot_action : nullptr, E * 2,
                    &io.robot_action)) != EBC_OK)
      return rc;
    io.reward = st.out(a->reward, E); io.done = st.out(a->done, E); io.info = st.out(a->info, E);
    io.dmin = st.out(a->dmin, E * 3); io.dist_to_goal = st.out(a->dist_to_goal, E);
    io.robot_action_out = st.out(a->robot_action_out, E * 2);
    io.human_action = st.out(a->human_action, E * N * 2); io.ob = st.out(a->ob, E * R * 5);
    io.obs_rotated = st.out(a->obs_rotated, E * R * T);
  }
  const int epb = EBC_WAVE / s.N;
  const int blocks = (s.E + epb - 1) / epb;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (h->timing) {
    if (h->ev_used + 2 > h->ev.size()) {
      for (int q = 0; q < 2; ++q) {
        hipEvent_t ev;
        HIP_TRY(hipEventCreate(&ev));
        h->ev.push_back(ev);
      }
    }
    e0 = h->ev[h->ev_used++];
    e1 = h->ev[h->ev_used++];
    HIP_TRY(hipEventRecord(e0, h->stream));
  }
  switch (a->human_policy) {
    case EBC_HUMAN_ORCA:
      if (h->orca_gs) {
        if ((rc = launch_orca_group(h, io.auto_reset)) != EBC_OK) return rc;
        launch_step_T<EBC_HUMAN_EXTERNAL>(h, io, blocks);
      } else {
        launch_step_T<EBC_HUMAN_ORCA>(h, io, blocks);
      }
      break;
    case EBC_HUMAN_LINEAR: launch_step_T<EBC_HUMAN_LINEAR>(h, io, blocks); break;
    default: launch_step_T<EBC_HUMAN_EXTERNAL>(h, io, blocks); break;
  }
  HIP_TRY(hipGetLastError());
  if (h->timing) HIP_TRY(hipEventRecord(e1, h->stream));
  if (a->location != EBC_DEVICE) return st.finish();
  return EBC_OK;
}

int ebc_lookahead(void *handle, const EbcLookaheadArgs *a) {
  Handle *h;
  int rc = check_handle(handle, &h);
  if (rc) return rc;
  if (!a || a->struct_size != sizeof(EbcLookaheadArgs))
    return fail(EBC_ERR_INVALID, "EbcLookaheadArgs.struct_size");
  if (!h->has_reset) return fail(EBC_ERR_STATE, "ebc_lookahead before ebc_reset");
  if (!a->actions || a->n_actions <= 0) return fail(EBC_ERR_INVALID, "actions");
  if (a->n_actions > EBC_LA_MAX_ACTIONS) return fail(EBC_ERR_UNSUPPORTED, "more than 128 actions");
  if (a->human_policy != EBC_HUMAN_ORCA && a->human_policy != EBC_HUMAN_LINEAR &&
      a->human_policy != EBC_HUMAN_EXTERNAL && a->human_policy != EBC_HUMAN_CACHED)
    return fail(EBC_ERR_INVALID, "human_policy");
  if ((a->flags & EBC_FLAG_BORDER) && !a->border) return fail(EBC_ERR_INVALID, "border is NULL");
  const DevState &s = h->s;
  const size_t E = s.E, R = s.N + s.S, T = h->T, A = a->n_actions;
  LookIO io;
  memset(&io, 0, sizeof(io));
  io.A = a->n_actions;
  io.has_border = (a->flags & EBC_FLAG_BORDER) ? 1 : 0;
  if (io.has_border) memcpy(io.border, a->border, sizeof(io.border));
  Stager st{h};
  if (a->location == EBC_DEVICE) {
    io.actions = a->actions;
    io.reward = a->reward; io.done = a->done; io.info = a->info; io.dmin = a->dmin;
    io.next_ob = a->next_ob; io.rows = a->rows_rotated;
  } else {
    const size_t need = pad256(A * 2 * 8) + pad256(E * A * 8) + pad256(E * A) * 2 + pad256(E * A * 3 * 8) +
                        pad256(E * R * 5 * 8) + (a->rows_rotated ? pad256(E * A * R * T * 4) : 0) + 4096;
    if ((rc = ensure_stage(h, need)) != EBC_OK) return rc;
    if ((rc = st.in(a->actions, A * 2, &io.actions)) != EBC_OK) return rc;
    io.reward = st.out(a->reward, E * A); io.done = st.out(a->done, E * A);
    io.info = st.out(a->info, E * A); io.dmin = st.out(a->dmin, E * A * 3);
    io.next_ob = st.out(a->next_ob, E * R * 5); io.rows = st.out(a->rows_rotated, E * A * R * T);
  }
  if (a->human_policy == EBC_HUMAN_ORCA || a->human_policy == EBC_HUMAN_LINEAR)
    if ((rc = launch_policy(h, a->human_policy)) != EBC_OK) return rc;
  if (h->T == 17)
    hipLaunchKernelGGL((lookahead_kernel<17>), dim3(s.E), dim3(EBC_WAVE), 0, h->stream, h->p, h->s, io);
  else
    hipLaunchKernelGGL((lookahead_kernel<13>), dim3(s.E), dim3(EBC_WAVE), 0, h->stream, h->p, h->s, io);
  HIP_TRY(hipGetLastError());
  if (a->location != EBC_DEVICE) return st.finish();
  return EBC_OK;
}

int ebc_get_state(void *handle, const EbcStateView *v) {
  Handle *h;
  int rc = check_handle(handle, &h);
  if (rc) return rc;
  if (!v || v->struct_size != sizeof(EbcStateView)) return fail(EBC_ERR_INVALID, "EbcStateView.struct_size");
  const DevState &s = h->s;
  const size_t E = s.E, EN = (size_t)s.E * s.N;
  const hipMemcpyKind kind = v->location == EBC_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
#define CP_(dst, src, bytes) if (dst) HIP_TRY(hipMemcpyAsync((dst), (src), (bytes), kind, h->stream))
  CP_(v->px, s.px, EN * 8); CP_(v->py, s.py, EN * 8); CP_(v->vx, s.vx, EN * 8); CP_(v->vy, s.vy, EN * 8);
  CP_(v->gx, s.gx, EN * 8); CP_(v->gy, s.gy, EN * 8); CP_(v->radius, s.radius, EN * 8);
  CP_(v->v_pref, s.v_pref, EN * 8); CP_(v->type, s.type, EN); CP_(v->n_humans, s.n_humans, E * 4);
  CP_(v->robot, s.robot, E * 9 * 8); CP_(v->global_time, s.time, E * 8);
  CP_(v->arrival_time, s.arrival, EN * 8); CP_(v->done, s.done, E);
#undef CP_
  if (v->location != EBC_DEVICE) HIP_TRY(hipStreamSynchronize(h->stream));
  return EBC_OK;
}

int ebc_timing(void *handle, int enable) {
  Handle *h;
  int rc = check_handle(handle, &h);
  if (rc) return rc;
  h->timing = enable != 0;
  return EBC_OK;
}

int ebc_timing_read(void *handle, int reset, double *avg_ms, int64_t *launches) {
  Handle *h;
  int rc = check_handle(handle, &h);
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(h->stream));
  for (size_t q = 0; q + 1 < h->ev_used; q += 2) {
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, h->ev[q], h->ev[q + 1]));
    h->ms_sum += ms;
    h->launches += 1;
  }
  h->ev_used = 0;
  if (avg_ms) *avg_ms = h->launches ? h->ms_sum / (double)h->launches : 0.0;
  if (launches) *launches = h->launches;
  if (reset) {
    h->ms_sum = 0;
    h->launches = 0;
  }
  return EBC_OK;
}

}  // extern "C"
