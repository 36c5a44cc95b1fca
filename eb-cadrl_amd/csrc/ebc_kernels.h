// ebc_kernels.h — the HIP kernels of libebcsim.so (gfx950 / MI355X).
//
// One env.step (simulator/env.py:388-466) is two launches:
//
//   phase1_kernel<GS>   heterogeneous grid.  Workgroups [0, env_blocks) take the ENV role,
//                       the rest the ORCA role; both read only pre-step state, so they run
//                       side by side and the robot-side serial work (one lane per env) hides
//                       behind the ORCA waves.
//       ENV role   lane = human, floor(64 / N) envs per wave.  Robot action, swept
//                  robot-human distance with the humans' CURRENT velocity
//                  (collisions.py:35-42), ordered per-type reduction (env.py:303-313), grid
//                  window (env.py:227-271), reward / done / info (reward.py:80-181), robot
//                  update (agent.py:202-228) into the "next" robot/time buffers.
//       ORCA role  GS lanes per human (ebc_orca_group.h) over the float tile the previous
//                  phase 2 (or reset) left in HBM -> hact.
//   phase2_kernel<POLICY, T>  lane = human.  Humans move (agent.py:202-211), first-arrival
//                  times (env.py:365-378), returned observation raw and rotated
//                  (env.py:381-382, :457-458; cadrl.py:236-337), auto-reset, and the float
//                  tile of the state the NEXT step will see (the casts rvo2 makes,
//                  orca.py:110-140, fused into the producer).
//
//   lookahead_kernel<T> the |A|-way onestep_lookahead sweep (multi_human_rl.py:38-61).
//
// HBM layout: struct-of-arrays [E][N] per human field (lane = e*N + i -> contiguous wave
// accesses), robot [E][9] and time [E] double-buffered (cur / next), grid [E][G][2] u64.
#pragma once

#include "ebc_device.h"
#include "ebc_orca_group.h"

namespace ebc {

struct DevState {
  int E, N, S, G;
  int *n_humans;
  double *px, *py, *vx, *vy, *gx, *gy, *radius, *v_pref;
  uint8_t *type;
  int *n_static;
  double *spx, *spy, *sradius;
  uint64_t *grid;  // nullptr when every map is free
  double *robot, *robot_n;  // current / next [E][9]
  double *time, *time_n;    // current / next [E]
  double *arrival;
  uint8_t *done;  // terminal flag of the last step
  double *hact;   // [E][N][2] human velocities: ORCA role, ebc_set_human_actions, or look-ahead cache
  double *px0, *py0, *vx0, *vy0, *robot0;  // reset() copies for auto-reset
  // what rvo2 would hold for the current state (float): position, velocity,
  // radius + 0.01 + safety, maxSpeed, preferred velocity
  float *fpx, *fpy, *fvx, *fvy, *frad, *fmax, *fprefx, *fprefy;
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

// lane -> (env, human) of the lane-per-human mapping
struct HumanLane {
  int el, i, e, n;
  bool env_ok, active, leader;
  size_t k;
};
__device__ __forceinline__ HumanLane human_lane(const DevState &s, int block) {
  HumanLane m;
  const int N = s.N, lane = threadIdx.x;
  const int epb = EBC_WAVE / N;
  m.el = lane / N;
  m.i = lane - m.el * N;
  m.e = block * epb + m.el;
  m.env_ok = m.el < epb && m.e < s.E;
  m.n = m.env_ok ? s.n_humans[m.e] : 0;
  m.active = m.env_ok && m.i < m.n;
  m.leader = m.env_ok && m.i == 0;
  m.k = (size_t)(m.env_ok ? m.e : 0) * N + m.i;
  return m;
}

// The float tile entry of one human (orca.py:110-140): written by reset and by phase 2.
__device__ __forceinline__ void store_tile(const EbcParams &p, const DevState &s, size_t k, double px,
                                           double py, double vx, double vy, double gx, double gy,
                                           double rad, double vpref) {
  float prefx, prefy;
  orca_pref_velocity(px, py, gx, gy, prefx, prefy);
  s.fpx[k] = (float)px;
  s.fpy[k] = (float)py;
  s.fvx[k] = (float)vx;
  s.fvy[k] = (float)vy;
  s.frad[k] = (float)(rad + 0.01 + p.orca_safety_space);  // orca.py:116, :122-126
  s.fmax[k] = (float)vpref;                                // orca.py:117
  s.fprefx[k] = prefx;
  s.fprefy[k] = prefy;
}

__global__ __launch_bounds__(256) void tile_kernel(EbcParams p, DevState s) {
  const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= (size_t)s.E * s.N) return;
  store_tile(p, s, k, s.px[k], s.py[k], s.vx[k], s.vy[k], s.gx[k], s.gy[k], s.radius[k], s.v_pref[k]);
}

// ------------------------------------------------------------------------- ENV role
__device__ __forceinline__ void env_role(const EbcParams &p, const DevState &s, const StepIO &io,
                                         int block) {
  __shared__ double sh_d[EBC_WAVE];
  __shared__ double sh_ract[EBC_WAVE][2];
  __shared__ uint8_t sh_type[EBC_WAVE];
  const HumanLane m = human_lane(s, block);
  const int lane = threadIdx.x;
  const double dt = p.time_step;
  const double *rb_in = s.robot + (size_t)(m.env_ok ? m.e : 0) * 9;
  double rb[9];
#pragma unroll
  for (int c = 0; c < 9; ++c) rb[c] = m.env_ok ? rb_in[c] : 0.0;
  const double gtime = m.env_ok ? s.time[m.e] : 0.0;
  double px = 0, py = 0, vx = 0, vy = 0, rad = 0;
  int type = 0;
  if (m.active) {
    px = s.px[m.k];
    py = s.py[m.k];
    vx = s.vx[m.k];
    vy = s.vy[m.k];
    rad = s.radius[m.k];
    type = s.type[m.k];
  }
  if (m.leader) {
    double a0, a1;
    if (io.robot_policy == EBC_ROBOT_LINEAR) {
      linear_policy(rb[0], rb[1], rb[5], rb[6], rb[7], a0, a1);
    } else {
      a0 = io.robot_action[2 * (size_t)m.e];
      a1 = io.robot_action[2 * (size_t)m.e + 1];
    }
    sh_ract[m.el][0] = a0;
    sh_ract[m.el][1] = a1;
  }
  __syncthreads();
  const double a0 = sh_ract[m.env_ok ? m.el : 0][0], a1 = sh_ract[m.env_ok ? m.el : 0][1];
  double rvx, rvy;
  if (p.robot_kinematics == EBC_HOLONOMIC) {
    rvx = a0;
    rvy = a1;
  } else {
    rvx = a0 * cos(a1 + rb[8]);
    rvy = a0 * sin(a1 + rb[8]);
  }
  sh_d[lane] = m.active ? closest_dist(px, py, vx, vy, rad, rb[0], rb[1], rb[4], rvx, rvy, dt) : 0.0;
  sh_type[lane] = (uint8_t)type;
  __syncthreads();
  if (!m.leader) return;

  // ordered per-type reduction with break at the first hit (env.py:303-313)
  double dm0 = INFINITY, dm1 = INFINITY, dm2 = INFINITY;
  int c0 = 0, c1 = 0, c2 = 0;
  for (int j = 0; j < m.n; ++j) {
    const int t = sh_type[lane + j];
    const double d = sh_d[lane + j];
    const bool hit = d < 0;
    if (t == 0 && !c0) { c0 = hit; dm0 = (!hit && d < dm0) ? d : dm0; }
    if (t == 1 && !c1) { c1 = hit; dm1 = (!hit && d < dm1) ? d : dm1; }
    if (t == 2 && !c2) { c2 = hit; dm2 = (!hit && d < dm2) ? d : dm2; }
  }
  const double dmin[3] = {dm0, dm1, dm2};
  double nx, ny;
  robot_next_position(rb, p.robot_kinematics, a0, a1, dt, nx, ny);
  int coll[4] = {c0, c1, c2, 0};
  coll[3] = grid_collision(s.grid ? s.grid + (size_t)m.e * s.G * 2 : nullptr, s.G, p.map_size_m,
                           p.map_resolution, nx, ny, rb[4], io.has_border ? io.border : nullptr);
  const RewardOut ro = reward_compute(p, nx, ny, rb[5], rb[6], rb[4], a1, gtime, dmin, coll);
  // Agent.step for the robot (agent.py:202-228)
  rb[0] = nx;
  rb[1] = ny;
  if (p.robot_kinematics == EBC_HOLONOMIC) {
    rb[2] = a0;
    rb[3] = a1;
  } else {
    rb[8] = py_mod(rb[8] + a1, 2 * M_PI);
    rb[2] = a0 * cos(rb[8]);
    rb[3] = a0 * sin(rb[8]);
  }
  double *rb_out = s.robot_n + (size_t)m.e * 9;
#pragma unroll
  for (int c = 0; c < 9; ++c) rb_out[c] = rb[c];
  s.time_n[m.e] = gtime + dt;
  s.done[m.e] = (uint8_t)ro.done;
  const size_t e = (size_t)m.e;
  if (io.reward) io.reward[e] = ro.reward;
  if (io.done) io.done[e] = (uint8_t)ro.done;
  if (io.info) io.info[e] = (uint8_t)ro.info;
  if (io.dmin) {
    io.dmin[3 * e] = dm0;
    io.dmin[3 * e + 1] = dm1;
    io.dmin[3 * e + 2] = dm2;
  }
  if (io.dist_to_goal) io.dist_to_goal[e] = ro.dist_to_goal;
  if (io.robot_action_out) {
    io.robot_action_out[2 * e] = a0;
    io.robot_action_out[2 * e + 1] = a1;
  }
}

// ------------------------------------------------------------------------- ORCA role
// 64 / GS humans per wave; lane j of a group loads "other" j of its human in ob order
// (env.py:396-402): the humans before and after it, then the robot when it is visible.
template <int GS>
__device__ __forceinline__ void orca_role(const EbcParams &p, const DevState &s, int block) {
  constexpr int HPW = EBC_WAVE / GS;
  __shared__ __align__(16) float dist_lds[EBC_WAVE];
  __shared__ float4 lines_lds[EBC_WAVE];
  __shared__ float4 proj_lds[EBC_WAVE];
  const int N = s.N;
  const int lane = threadIdx.x;
  const int group = lane / GS;
  const int j = lane - group * GS;
  const long h = (long)block * HPW + group;
  const bool h_ok = h < (long)s.E * N;
  const int e = h_ok ? (int)(h / N) : 0;
  const int i = h_ok ? (int)(h - (long)e * N) : 0;
  const int n = h_ok ? s.n_humans[e] : 0;
  const bool human_ok = h_ok && i < n;
  const size_t base = (size_t)e * N;
  // tile loads do not wait for n_humans: indices are clamped into the env's row, validity is
  // decided afterwards (padded slots hold zeros)
  const size_t ks = base + i;
  const int oj = j < i ? j : j + 1;
  const size_t ko = base + (oj < N ? oj : N - 1);
  float posx = 0, posy = 0, velx = 0, vely = 0, radius = 0, maxSpeed = 0, prefx = 0, prefy = 0;
  float opx = 0, opy = 0, ovx = 0, ovy = 0, orad = 0;
  if (h_ok) {
    posx = s.fpx[ks];
    posy = s.fpy[ks];
    velx = s.fvx[ks];
    vely = s.fvy[ks];
    radius = s.frad[ks];
    maxSpeed = s.fmax[ks];
    prefx = s.fprefx[ks];
    prefy = s.fprefy[ks];
    opx = s.fpx[ko];
    opy = s.fpy[ko];
    ovx = s.fvx[ko];
    ovy = s.fvy[ko];
    orad = s.frad[ko];
  }
  const int n_others = human_ok ? (n - 1 + (p.robot_visible ? 1 : 0)) : 0;
  const bool valid = j < n_others;
  if (p.robot_visible && valid && j == n - 1) {  // the robot, last in ob (env.py:401-402)
    const double *rb = s.robot + (size_t)e * 9;
    opx = (float)rb[0];
    opy = (float)rb[1];
    ovx = (float)rb[2];
    ovy = (float)rb[3];
    orad = (float)(rb[4] + 0.01 + p.orca_safety_space);
  }
  float ox, oy;
  orca_group<GS>(p, j, group, valid, posx, posy, velx, vely, radius, maxSpeed, prefx, prefy, opx, opy,
                 ovx, ovy, orad, dist_lds + group * GS, lines_lds + group * GS, proj_lds + group * GS,
                 ox, oy);
  if (h_ok && j == 0) {
    s.hact[(size_t)h * 2] = human_ok ? (double)ox : 0.0;  // getAgentVelocity -> Python float
    s.hact[(size_t)h * 2 + 1] = human_ok ? (double)oy : 0.0;
  }
}

template <int GS>
__global__ __launch_bounds__(EBC_WAVE) void phase1_kernel(EbcParams p, DevState s, StepIO io,
                                                          int env_blocks) {
  if ((int)blockIdx.x < env_blocks)
    env_role(p, s, io, blockIdx.x);
  else
    orca_role<GS>(p, s, blockIdx.x - env_blocks);
}

// ------------------------------------------------------------------------- phase 2
template <int POLICY, int T>
__global__ __launch_bounds__(EBC_WAVE) void phase2_kernel(EbcParams p, DevState s, StepIO io) {
  const HumanLane m = human_lane(s, blockIdx.x);
  const int N = s.N, S = s.S, R = N + S;
  const double dt = p.time_step;
  if (!m.env_ok) return;
  const size_t e = (size_t)m.e;
  double rbn[9];
  const double *rbp = s.robot_n + e * 9;
#pragma unroll
  for (int c = 0; c < 9; ++c) rbn[c] = rbp[c];
  const double tnew = s.time_n[e];
  const bool restore = io.auto_reset && s.done[e];
  const int ns = S ? s.n_static[e] : 0;

  double px = 0, py = 0, vx = 0, vy = 0, gx = 0, gy = 0, rad = 0, vpref = 0, arrival = 0, ax = 0, ay = 0;
  int type = 0;
  if (m.active) {
    px = s.px[m.k];
    py = s.py[m.k];
    gx = s.gx[m.k];
    gy = s.gy[m.k];
    rad = s.radius[m.k];
    vpref = s.v_pref[m.k];
    type = s.type[m.k];
    arrival = s.arrival[m.k];
    if (POLICY == EBC_HUMAN_LINEAR) {
      linear_policy(px, py, gx, gy, vpref, ax, ay);  // on the pre-step state (env.py:393-405)
    } else {
      ax = s.hact[m.k * 2];
      ay = s.hact[m.k * 2 + 1];
    }
    // Agent.step (agent.py:202-211), first arrival (env.py:365-378)
    px = px + ax * dt;
    py = py + ay * dt;
    vx = ax;
    vy = ay;
    if (arrival == 0 && norm2(px - gx, py - gy) < rad) arrival = tnew;
  }
  if (io.human_action) {
    io.human_action[m.k * 2] = ax;
    io.human_action[m.k * 2 + 1] = ay;
  }

  // returned observation: humans then static rows, raw and in the robot frame
  if (io.ob || io.obs_rotated) {
    const RotFrame f = rot_frame(rbn, p.rotate_unicycle);
    for (int r = m.i; r < R; r += N) {
      double opx = 0, opy = 0, ovx = 0, ovy = 0, orad = 0;
      int otype = 0;
      bool valid = false;
      if (r < m.n) {  // r == i: this lane's own human
        opx = px; opy = py; ovx = vx; ovy = vy; orad = rad; otype = type;
        valid = true;
      } else if (r - m.n < ns) {
        const size_t q = e * S + (r - m.n);
        opx = s.spx[q]; opy = s.spy[q]; orad = s.sradius[q]; otype = EBC_ADULT_STATIC;
        valid = true;
      }
      const size_t row = e * R + r;
      if (io.ob) {
        double *o = io.ob + row * 5;
        o[0] = opx; o[1] = opy; o[2] = ovx; o[3] = ovy; o[4] = orad;
      }
      if (io.obs_rotated) {
        float out[T];
        if (valid) {
          rotate_row<T>(f, opx, opy, ovx, ovy, orad, otype, out);
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

  // commit: the moved state, or the reset() scene when this step was terminal and auto-reset is on
  if (m.active) {
    if (restore) {
      px = s.px0[m.k];
      py = s.py0[m.k];
      vx = s.vx0[m.k];
      vy = s.vy0[m.k];
      arrival = 0;
    }
    s.px[m.k] = px;
    s.py[m.k] = py;
    s.vx[m.k] = vx;
    s.vy[m.k] = vy;
    s.arrival[m.k] = arrival;
    store_tile(p, s, m.k, px, py, vx, vy, gx, gy, rad, vpref);
  }
  if (m.leader && restore) {
    const double *r0 = s.robot0 + e * 9;
    double *rw = s.robot_n + e * 9;
#pragma unroll
    for (int c = 0; c < 9; ++c) rw[c] = r0[c];
    s.time_n[e] = 0.0;
  }
}

// ------------------------------------------------------------------------- look-ahead
// One wave per env.  Human velocities come from s.hact (ORCA role / set_human_actions), or the
// linear policy.
//   phase A  lanes over rows: next observable rows into LDS (agent.py:80-93; env.py:457-458)
//   phase B  lanes over actions: ordered collisions, grid, reward, robot frame of rotate()
//   phase C  lanes over (action, row): rotated rows
#define EBC_LA_MAX_ROWS 128
#define EBC_LA_MAX_ACTIONS 128
template <int POLICY, int T>
__global__ __launch_bounds__(EBC_WAVE) void lookahead_kernel(EbcParams p, DevState s, LookIO io) {
  __shared__ double hpx[EBC_WAVE], hpy[EBC_WAVE], hvx[EBC_WAVE], hvy[EBC_WAVE], hrad[EBC_WAVE];
  __shared__ uint8_t htype[EBC_WAVE];
  __shared__ double row[EBC_LA_MAX_ROWS][5];
  __shared__ uint8_t row_type[EBC_LA_MAX_ROWS];
  __shared__ RotFrame frames[EBC_LA_MAX_ACTIONS];
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

  for (int r = lane; r < R; r += EBC_WAVE) {
    double o[5] = {0, 0, 0, 0, 0};
    int t = 0;
    if (r < n) {
      const size_t k = (size_t)e * N + r;
      const double px = s.px[k], py = s.py[k];
      double ax, ay;
      if (POLICY == EBC_HUMAN_LINEAR) {
        linear_policy(px, py, s.gx[k], s.gy[k], s.v_pref[k], ax, ay);
        s.hact[k * 2] = ax;  // cache for a following EBC_HUMAN_CACHED step
        s.hact[k * 2 + 1] = ay;
      } else {
        ax = s.hact[k * 2];
        ay = s.hact[k * 2 + 1];
      }
      hpx[r] = px; hpy[r] = py; hvx[r] = s.vx[k]; hvy[r] = s.vy[k];
      hrad[r] = s.radius[k]; htype[r] = s.type[k];
      o[0] = px + ax * dt;
      o[1] = py + ay * dt;
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
    double dm0 = INFINITY, dm1 = INFINITY, dm2 = INFINITY;
    int c0 = 0, c1 = 0, c2 = 0;
    for (int j = 0; j < n; ++j) {
      const int t = htype[j];
      const bool live = (t == 0 && !c0) || (t == 1 && !c1) || (t == 2 && !c2);
      if (live) {
        const double d = closest_dist(hpx[j], hpy[j], hvx[j], hvy[j], hrad[j], rb[0], rb[1], rb[4],
                                      rvx, rvy, dt);
        const bool hit = d < 0;
        if (t == 0) { c0 = hit; dm0 = (!hit && d < dm0) ? d : dm0; }
        if (t == 1) { c1 = hit; dm1 = (!hit && d < dm1) ? d : dm1; }
        if (t == 2) { c2 = hit; dm2 = (!hit && d < dm2) ? d : dm2; }
      }
    }
    const double dmin[3] = {dm0, dm1, dm2};
    double nx, ny;
    robot_next_position(rb, p.robot_kinematics, a0, a1, dt, nx, ny);
    int coll[4] = {c0, c1, c2, 0};
    coll[3] = grid_collision(s.grid ? s.grid + (size_t)e * s.G * 2 : nullptr, s.G, p.map_size_m,
                             p.map_resolution, nx, ny, rb[4], io.has_border ? io.border : nullptr);
    const RewardOut ro = reward_compute(p, nx, ny, rb[5], rb[6], rb[4], a1, gtime, dmin, coll);
    const size_t o = (size_t)e * A + a;
    if (io.reward) io.reward[o] = ro.reward;
    if (io.done) io.done[o] = (uint8_t)ro.done;
    if (io.info) io.info[o] = (uint8_t)ro.info;
    if (io.dmin) {
      io.dmin[3 * o] = dm0;
      io.dmin[3 * o + 1] = dm1;
      io.dmin[3 * o + 2] = dm2;
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
      frames[a] = rot_frame(nb, p.rotate_unicycle);
    }
  }
  if (!io.rows) return;
  __syncthreads();

  const int total = A * R;
  for (int idx = lane; idx < total; idx += EBC_WAVE) {
    const int a = idx / R, r = idx - a * R;
    float out[T];
    if (r < n + ns) {
      rotate_row<T>(frames[a], row[r][0], row[r][1], row[r][2], row[r][3], row[r][4], row_type[r], out);
    } else {
#pragma unroll
      for (int c = 0; c < T; ++c) out[c] = 0.0f;
    }
    float *dst = io.rows + ((size_t)e * A * R + idx) * T;
#pragma unroll
    for (int c = 0; c < T; ++c) dst[c] = out[c];
  }
}

}  // namespace ebc
