// ebc_kernels.h — the HIP kernels of libebcsim.so (gfx950 / MI355X).
//
// One env.step (simulator/env.py:388-466) = robot-side "service" work (lane per human: robot
// action, collisions, reward, then human update + observation) plus the humans' ORCA velocities
// (GS lanes per human, ebc_orca_group.h, over the float tile the previous step left in HBM).
//   orca_step_kernel<GS,T>     ORCA humans, one launch of one-wave workgroups in four roles
//                              (ENV, ORCA, ROWS, STATE) that meet through mailbox words.
//   step_kernel<POLICY,T>      humans on the linear policy or with supplied / cached velocities:
//                              one launch, service_env + service_commit in the same wave.
// (A first fused form -- service wave + ORCA waves per WORKGROUP with an LDS hand-off -- was 1.6x
// slower than two launches at 4096 x 10; roles as separate one-wave workgroups are not: DESIGN.md.)
//   orca_kernel<GS>     ORCA alone -> hact (look-ahead prelude)
//   lookahead_kernel<T> the |A|-way onestep_lookahead sweep (multi_human_rl.py:38-61).
//
// HBM layout: struct-of-arrays [E][N] per human field (lane = e*N + i -> contiguous wave
// accesses), robot [E][9] (two buffers), time [E], grid per pool slot [G][2] u64.
#pragma once

#include "ebc_device.h"

// s_sleep units (64 cycles each) between two polls of a mailbox: 2049 consumer waves poll while the ORCA waves work
#ifndef EBC_ENV_PRIO
#define EBC_ENV_PRIO 3
#endif
#ifndef EBC_ORCA_PRIO
#define EBC_ORCA_PRIO 1
#endif
#ifndef EBC_POLL_SLEEP
#define EBC_POLL_SLEEP 2
#endif
#include "ebc_orca_group.h"
#include "ebc_scene_gen.h"

namespace ebc {


// Scenes an env restarts from under EBC_FLAG_AUTO_RESET, in the same SoA layout as the state.
// Slots [0, E) are the envs' own ebc_reset scenes; slots [E, E + P) a host-generated pool installed
// by ebc_set_scene_pool.  cursor[e] is the slot env e restarts from next: e itself without a pool
// (stride 0), else it walks E + (e mod P), + stride, ... (mod P) from episode to episode without
// leaving the device.  Occupancy grids are never copied: an env points at the grid of the slot it
// is running (grid_scene[e]).
struct ScenePool {
  int P, stride;  // custom scenes (0 = none) and the cursor step
  int *cursor;  // [E]
  int *n_humans;
  double *px, *py, *vx, *vy, *gx, *gy, *radius, *v_pref;
  uint8_t *type;
  int *n_static;
  double *spx, *spy, *sradius;
  uint64_t *grid;  // nullptr: free maps
  double *robot;
  float2 *pref;    // [slots][N] ORCA preferred velocity of the scene's humans (float tile, orca.py:136-140)
};

struct DevState {
  int E, N, S, G;
  int *n_humans;
  double *px, *py, *vx, *vy, *gx, *gy, *radius, *v_pref;
  uint8_t *type;
  int *n_static;
  double *spx, *spy, *sradius;
  int *grid_scene;  // [E] pool slot whose occupancy grid env e uses (pool.grid is null when every map is free)
  double *robot;    // [E][9] FullState order
  double *robot_n;  // [E][9] the fused ORCA step writes the robots' next state here; the host then swaps the two
  double *time;   // [E] global_time
  double *arrival;
  uint8_t *done;  // terminal flag of the last step
  double *hact;   // [E][N][2] human velocities: ebc_set_human_actions, or the look-ahead cache
  // Mailboxes of the fused ORCA step (orca_step_kernel).  Every word carries the EPOCH of the launch that
  // wrote it (a counter that never repeats within 2^32 launches and is never 0): a consumer polls until the
  // tag is this launch's, nobody ever empties a box, several roles may read the same one, and a box left
  // over from an aborted launch cannot be mistaken for a fresh one.  (Round 1 kept data + EMPTY in one
  // 8-byte word, one box per consumer, re-emptied by it: four 8-byte fabric writes per human and step
  // instead of one 16-byte one.)
  uint4 *vel;                     // [E][N] ORCA -> STATE, ROWS: {vx bits, epoch, vy bits, epoch}, one 16-byte store
  unsigned long long *env_done;   // [E] ENV -> STATE: epoch << 32 | 1 + done
  unsigned *rows_loaded;          // [E] ROWS -> STATE: = epoch once the pre-step state AND the robot's next state are in registers
  unsigned *robot_ready;          // [E] ENV -> ROWS: = epoch once robot_n[e] holds this step's result
  unsigned *fault;                // [1] a poll gave up (protocol broken); ebc_synchronize reports it
  // Second form of the fused ORCA step (orca_step2_kernel): an ORCA group commits its own human, so the humans'
  // next positions / velocities and the next float tile go to SECOND buffers (the pre-step ones are still being
  // read by other waves of the launch); the host swaps the pairs after the launch, like robot / robot_n.
  double *px_n, *py_n, *vx_n, *vy_n;
  float4 *tile_n;
  uint4 *frame;         // [E][4] ENV1 -> row writers: the robot-centric frame of the NEXT robot state (rotate()), 3 floats + epoch each
  uint4 *ract;          // [E][4] ENV1 -> ENV2: robot action a0, a1 and next position x, y as {lo, hi, epoch, epoch}
  unsigned *committed;  // [E][N] ORCA -> ENV2 (restarts only): = epoch once the human's next state is in memory
  ScenePool pool;  // where auto-reset takes an env's next scene from
  // what rvo2 would hold for the current state (float), one 32-byte record per human slot:
  // tile[2k] = (position, velocity), tile[2k + 1] = (radius + 0.01 + safety, maxSpeed, preferred
  // velocity).  An ORCA lane reads its human with two 16-byte loads and an "other" with 16 + 4.
  float4 *tile;
  // uniform ORCA constants (float, as rvo2 computes them) and the magic number of h / N
  float inv_time_horizon, inv_time_step, range_sq;
  unsigned n_magic, n_shift;
};

#ifndef EBC_SPIN_LIMIT
#define EBC_SPIN_LIMIT (1u << 22)               // polls before a consumer gives up (~1 s)
#endif

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

// The float tile entry of one human (orca.py:110-140): written by reset and by the wave that moves the human.
__device__ __forceinline__ void store_tile(const EbcParams &p, const DevState &s, size_t k, double px,
                                           double py, double vx, double vy, double gx, double gy,
                                           double rad, double vpref) {
  float prefx, prefy;
  orca_pref_velocity(px, py, gx, gy, prefx, prefy);
  s.tile[2 * k] = make_float4((float)px, (float)py, (float)vx, (float)vy);
  // orca.py:116, :122-126 (radius), :117 (maxSpeed)
  s.tile[2 * k + 1] = make_float4((float)(rad + 0.01 + p.orca_safety_space), (float)vpref, prefx, prefy);
}

// The preferred velocities of pool scenes [first, first + count) (in human slots), so that a
// restart copies them instead of redoing a double sqrt and two divisions inside the step.
__global__ __launch_bounds__(256) void pool_pref_kernel(DevState s, size_t first, size_t count) {
  const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= count) return;
  const size_t k = first + q;
  float prefx, prefy;
  orca_pref_velocity(s.pool.px[k], s.pool.py[k], s.pool.gx[k], s.pool.gy[k], prefx, prefy);
  s.pool.pref[k] = make_float2(prefx, prefy);
}

// Before the pool arrays are replaced (ebc_set_scene_pool / ebc_generate_pool on a running batch): an env in the
// middle of an episode on a pool scene keeps only that slot's INDEX for its occupancy grid.  One workgroup per env
// copies the grid into the env's own slot e — a reset slot, never a restart source once a pool is installed — and
// points the env at it, so the grid outlives the pool it came from.
__global__ __launch_bounds__(256) void rehome_grid_kernel(DevState s, uint64_t *grid) {
  const int e = blockIdx.x;
  const int src = s.grid_scene[e];
  if (src == e) return;
  const size_t words = (size_t)s.G * 2;
  for (size_t w = threadIdx.x; w < words; w += blockDim.x) grid[(size_t)e * words + w] = grid[(size_t)src * words + w];
  __syncthreads();  // every thread has read grid_scene[e]
  if (threadIdx.x == 0) s.grid_scene[e] = e;
}

__global__ __launch_bounds__(256) void tile_kernel(EbcParams p, DevState s) {
  const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= (size_t)s.E * s.N) return;
  store_tile(p, s, k, s.px[k], s.py[k], s.vx[k], s.vy[k], s.gx[k], s.gy[k], s.radius[k], s.v_pref[k]);
}

// ------------------------------------------------------------------------- ORCA wave
// One wave = 64 / GS humans; lane j of a group loads "other" j of its human in ob order
// (env.py:396-402): the humans before and after it, then the robot when it is visible.
// `scratch` = this wave's private LDS (EBC_ORCA_LDS bytes).  Returns the group's velocity in
// every lane of the group.
// LDS of one ORCA wave: per group its distances and three line-sized arrays (lines, segments,
// projected lines).  64 / GS groups hold humans; when GS does not divide 64 the left-over lanes
// form one more (idle) group with its own scratch.
template <int GS>
struct OrcaLds {
  using Sh = OrcaShape<GS>;
  static constexpr int HPW = EBC_WAVE / GS;
  static constexpr int GROUPS = (EBC_WAVE + GS - 1) / GS;
  static constexpr int DIST = GROUPS * Sh::DIST * 4;
  static constexpr int BYTES = DIST + 3 * GROUPS * Sh::LINES * 16;
};

// What an ORCA wave needs before it can issue its first load.  In the step launch these are the
// leading kernel arguments, preloaded into scalar registers when the wave starts
// (-amdgpu-kernarg-preload-count): role, human and tile address without a kernarg round trip.
struct OrcaHot {
  int E, N;
  unsigned n_magic, n_shift;
  const float4 *tile;
  const int *n_humans;
};
__device__ __forceinline__ OrcaHot orca_hot(const DevState &s) {
  return OrcaHot{s.E, s.N, s.n_magic, s.n_shift, s.tile, s.n_humans};
}

// What an ORCA lane loads before anything else: its human's tile record and "other" j of it.
struct OrcaTile {
  float4 a, b, c;
  float orad;
  int n;
};
template <int GS>
__device__ __forceinline__ OrcaTile orca_tile_load(const OrcaHot &hot, bool h_ok, int e, int i) {
  const int N = hot.N;
  const int lane = threadIdx.x & (EBC_WAVE - 1);
  const int group = lane / GS;
  const int j = lane - group * GS;  // "other" index in ob order
  OrcaTile t;
  t.n = h_ok ? hot.n_humans[e] : 0;
  const size_t base = (size_t)e * N;
  // tile loads do not wait for n_humans: indices are clamped into the env's row, validity is
  // decided afterwards (padded slots hold zeros)
  const size_t ks = base + i;
  const int oj = j < i ? j : j + 1;  // humans before / after this one
  const size_t ko = base + (oj < N ? oj : N - 1);
  t.a = t.b = t.c = make_float4(0, 0, 0, 0);
  t.orad = 0;
  if (h_ok) {
    t.a = hot.tile[2 * ks];
    t.b = hot.tile[2 * ks + 1];
    t.c = hot.tile[2 * ko];
    t.orad = hot.tile[2 * ko + 1].x;
  }
  return t;
}

template <int GS, typename Hook = NoHook>
__device__ __forceinline__ void orca_wave_compute(const EbcParams &p, const DevState &s, const OrcaHot &hot, const OrcaTile &t,
                                                  bool h_ok, int e, int i, unsigned char *scratch, float &ox, float &oy,
                                                  bool &human_ok, Hook after_rank = Hook()) {
  using L = OrcaLds<GS>;
  float *dist_lds = reinterpret_cast<float *>(scratch);
  float4 *lines_lds = reinterpret_cast<float4 *>(scratch + L::DIST);
  float4 *segs_lds = lines_lds + L::GROUPS * L::Sh::LINES;
  float4 *proj_lds = segs_lds + L::GROUPS * L::Sh::LINES;
  const int N = hot.N;
  const int lane = threadIdx.x & (EBC_WAVE - 1);
  const int group = lane / GS;
  const int j = lane - group * GS;
  const int n = t.n;
  human_ok = h_ok && i < n;
  const float posx = t.a.x, posy = t.a.y, velx = t.a.z, vely = t.a.w;
  const float radius = t.b.x, maxSpeed = t.b.y, prefx = t.b.z, prefy = t.b.w;
  float opx = t.c.x, opy = t.c.y, ovx = t.c.z, ovy = t.c.w, orad = t.orad;
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
  orca_group<GS>(p, j, group, valid, posx, posy, velx, vely, radius, maxSpeed, prefx, prefy, opx, opy, ovx,
                 ovy, orad, dist_lds + group * L::Sh::DIST, lines_lds + group * L::Sh::LINES,
                 segs_lds + group * L::Sh::LINES, proj_lds + group * L::Sh::LINES,
                 N - 1 + (p.robot_visible ? 1 : 0), s.range_sq, s.inv_time_horizon, s.inv_time_step, ox, oy, after_rank);
}

template <int GS>
__device__ __forceinline__ void orca_wave(const EbcParams &p, const DevState &s, const OrcaHot &hot, bool h_ok, int e, int i,
                                          unsigned char *scratch, float &ox, float &oy, bool &human_ok) {
  const OrcaTile t = orca_tile_load<GS>(hot, h_ok, e, i);
  orca_wave_compute<GS>(p, s, hot, t, h_ok, e, i, scratch, ox, oy, human_ok);
}

// (env, slot) of flat human index h < E * N without the ~20-instruction integer divide
__device__ __forceinline__ void split_human(const OrcaHot &s, unsigned h, bool h_ok, int &e, int &i) {
  const unsigned N = (unsigned)s.N;
  const unsigned q = N > 1 ? (__umulhi(h, s.n_magic) >> s.n_shift) : h;
  e = h_ok ? (int)q : 0;
  i = h_ok ? (int)(h - q * N) : 0;
}

// ORCA alone -> s.hact: the prelude of a look-ahead sweep (the following step re-uses it).
template <int GS>
__global__ __launch_bounds__(EBC_WAVE) void orca_kernel(EbcParams p, DevState s) {
  const WaveTrace wt(0);
  constexpr int HPW = EBC_WAVE / GS;
  __shared__ __align__(16) unsigned char scratch[OrcaLds<GS>::BYTES];
  const int group = threadIdx.x / GS, j = threadIdx.x - group * GS;
  const unsigned h = (unsigned)blockIdx.x * HPW + group;
  const unsigned N = (unsigned)s.N;
  const bool h_ok = group < HPW && h < (unsigned)s.E * N;  // lanes past HPW * GS idle
  int e, i;
  const OrcaHot hot = orca_hot(s);
  split_human(hot, h, h_ok, e, i);
  float ox, oy;
  bool human_ok;
  orca_wave<GS>(p, s, hot, h_ok, e, i, scratch, ox, oy, human_ok);
  if (h_ok && j == 0) {
    s.hact[(size_t)h * 2] = human_ok ? (double)ox : 0.0;  // getAgentVelocity -> Python float
    s.hact[(size_t)h * 2 + 1] = human_ok ? (double)oy : 0.0;
  }
}

// ORCA.predict with the ROBOT as the agent (simulator/policy/orca.py:85-157): the demonstrator of the
// imitation-learning stage (rl/train.py:99-143 sets the policy's safety_space itself).  One GS-lane
// group per env; lane j = row j of the observation the policy is given — the humans (env.py:381-382)
// then the static obstacles as pedestrians (env.py:457-458, velocity 0) —, every radius + 0.01 +
// safety_space as orca.py:116-126 hands it to rvo2, the robot's maxSpeed = v_pref.  -> act[E][2].
// The demonstrator's persistent rvo2 simulator (simulator/policy/orca.py:96-133): the policy object keeps ONE simulator
// for all its calls and rebuilds it only when the number of agents changes; otherwise it updates positions and
// velocities — the rows' radii (+ 0.01 + safety_space), the robot's radius and its maxSpeed stay those of the call that
// built it, across env.reset() too (rl/train.py:130-133 makes one il_policy for the whole imitation-learning stage).
// Per env: rows[e] = rows of its simulator (-1: none yet), radius[e][N + S] and self[e] = {radius, maxSpeed} as rvo2
// holds them.  rows == nullptr: no persistent simulator (every call from the current state).
struct RobotSim {
  int *rows;
  float *radius;
  float2 *self;
};

template <int GS>
__global__ __launch_bounds__(EBC_WAVE) void orca_robot_kernel(EbcParams p, DevState s, double safety_space, double *act, RobotSim sim) {
  using L = OrcaLds<GS>;
  constexpr int EPW = EBC_WAVE / GS;
  __shared__ __align__(16) unsigned char scratch[L::BYTES];
  float *dist_lds = reinterpret_cast<float *>(scratch);
  float4 *lines_lds = reinterpret_cast<float4 *>(scratch + L::DIST);
  float4 *segs_lds = lines_lds + L::GROUPS * L::Sh::LINES;
  float4 *proj_lds = segs_lds + L::GROUPS * L::Sh::LINES;
  const int group = threadIdx.x / GS, j = threadIdx.x - group * GS;
  const int e = (int)blockIdx.x * EPW + group;
  const bool e_ok = group < EPW && e < s.E;  // lanes past EPW * GS idle
  const size_t ee = e_ok ? (size_t)e : 0;
  const int N = s.N, S = s.S;
  const int n = e_ok ? s.n_humans[ee] : 0;
  const int ns = (e_ok && S) ? s.n_static[ee] : 0;
  const double *rb = s.robot + ee * 9;
  const float posx = (float)rb[0], posy = (float)rb[1], velx = (float)rb[2], vely = (float)rb[3];
  float radius = (float)(rb[4] + 0.01 + safety_space), maxSpeed = (float)rb[7];
  float prefx, prefy;
  orca_pref_velocity(rb[0], rb[1], rb[5], rb[6], prefx, prefy);
  const bool valid = e_ok && j < n + ns;
  float opx = 0, opy = 0, ovx = 0, ovy = 0, orad = 0;
  if (valid && j < n) {
    const size_t k = ee * N + j;
    opx = (float)s.px[k];
    opy = (float)s.py[k];
    ovx = (float)s.vx[k];
    ovy = (float)s.vy[k];
    orad = (float)(s.radius[k] + 0.01 + safety_space);
  } else if (valid) {
    const size_t q = ee * S + (j - n);
    opx = (float)s.spx[q];
    opy = (float)s.spy[q];
    orad = (float)(s.sradius[q] + 0.01 + safety_space);
  }
  if (sim.rows && e_ok) {
    // every lane of the group reads the simulator's row count before lane 0 of the group replaces it (one wave)
    const bool rebuild = sim.rows[ee] != n + ns;  // sim is None, or getNumAgents() != len(agent_states) + 1
    const size_t q = ee * (size_t)(N + S) + j;
    if (rebuild) {
      if (valid) sim.radius[q] = orad;
      if (j == 0) {
        sim.self[ee] = make_float2(radius, maxSpeed);
        sim.rows[ee] = n + ns;
      }
    } else {
      if (valid) orad = sim.radius[q];
      const float2 me = sim.self[ee];
      radius = me.x;
      maxSpeed = me.y;
    }
  }
  float ox, oy;
  orca_group<GS>(p, j, group, valid, posx, posy, velx, vely, radius, maxSpeed, prefx, prefy, opx, opy, ovx, ovy,
                 orad, dist_lds + group * L::Sh::DIST, lines_lds + group * L::Sh::LINES,
                 segs_lds + group * L::Sh::LINES, proj_lds + group * L::Sh::LINES, N + S, s.range_sq,
                 s.inv_time_horizon, s.inv_time_step, ox, oy);
  if (e_ok && j == 0) {
    act[2 * ee] = (double)ox;  // getAgentVelocity -> Python float
    act[2 * ee + 1] = (double)oy;
  }
}

// ------------------------------------------------------------------------- step
// The "service" work of a step, lane = human slot (floor(64 / N) envs per wave):
//   service_env     robot action, swept robot-human distance with the humans' CURRENT velocity
//                   (collisions.py:35-42), ordered per-type reduction (env.py:303-313), grid
//                   window (env.py:227-271), reward / done / info (reward.py:80-181), robot
//                   update (agent.py:202-228).  Pre-step state only.
//   service_commit  humans move (agent.py:202-211), first arrivals (env.py:365-378), returned
//                   observation raw and rotated (env.py:381-382, :457-458; cadrl.py:236-337),
//                   auto-reset, and the float tile of the state the NEXT step will see (the
//                   casts rvo2 makes, orca.py:110-140, fused into the producer).
// An empty asm that "uses and redefines" a value: the compiler must have the value ready HERE, so
// the s_waitcnt for a load lands at this point (before the step's first store: vector memory
// operations retire in issue order and a wait placed behind stores waits for them too), and an
// address computed from kernel arguments is materialised in VGPRs here instead of re-reading the
// kernarg segment (s_load + lgkmcnt(0)) in the middle of the dependent chain.
template <typename Tp>
__device__ __forceinline__ void pin(Tp &x) {
  asm volatile("" : "+v"(x));
}

// A reference to an object in LDS whose reads the compiler cannot move above this point.  The argument
// block is read from LDS (below); left alone the compiler hoists those ds_reads to the top of the wave —
// and then SPILLS the values to scratch when they are only needed at its end (found in the ENV role: 14
// spilled registers were the time step, grid pointer, global time, ... read early, used late; any scratch
// use costs every wave of the launch its scratch set-up).
template <typename Tp>
__device__ __forceinline__ const Tp &read_late(const Tp &x) {
  typedef const Tp __attribute__((address_space(3))) *LdsPtr;
  unsigned off = (unsigned)(size_t)(LdsPtr)&x;  // the 32-bit LDS address goes through the asm, not a pointer
  asm volatile("" : "+v"(off));
  return *(const Tp *)(LdsPtr)(size_t)off;
}

// The kernel-argument block lives in (device) memory that is not cached like ordinary data: every
// s_load of it that the compiler places in the middle of the service wave's dependent chain is a
// memory round trip (the argument structs are ~700 B, far more than the SGPR file keeps live, so
// the compiler re-reads them ~30 times).  The service wave therefore copies the block into LDS
// once, with a few wave-wide vector loads, and reads fields from LDS afterwards.
struct ArgBlock {
  EbcParams p;
  DevState s;
  StepIO io;
};
template <typename Tp>
__device__ __forceinline__ void stage_args(Tp *dst_lds, const Tp &src, int lane) {
  const uint32_t *from = reinterpret_cast<const uint32_t *>(&src);
  uint32_t *to = reinterpret_cast<uint32_t *>(dst_lds);
  for (int w = lane; w < (int)(sizeof(Tp) / 4); w += EBC_WAVE) to[w] = from[w];
}

struct LaneMap {
  int el, i, n, ns;
  size_t ee, k;
  bool env_ok, active, leader;
};
struct HumanRegs {
  double px, py, vx, vy, gx, gy, rad, vpref, arrival;
  int type;
};

__device__ __forceinline__ LaneMap lane_map(const DevState &s, int e0, int epb, int lane) {
  LaneMap m;
  const int N = s.N;
  m.el = lane / N;
  m.i = lane - m.el * N;
  const int e = e0 + m.el;
  m.env_ok = m.el < epb && e < s.E;
  m.ee = m.env_ok ? (size_t)e : 0;
  m.k = m.ee * N + m.i;
  m.n = m.env_ok ? s.n_humans[m.ee] : 0;
  m.ns = (m.env_ok && s.S) ? s.n_static[m.ee] : 0;
  m.active = m.env_ok && m.i < m.n;
  m.leader = m.env_ok && m.i == 0;
  return m;
}

__device__ __forceinline__ HumanRegs load_human(const DevState &s, const LaneMap &m) {
  HumanRegs h = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  if (m.env_ok) {  // padded slots hold zeros; `active` decides what is used
    h.px = s.px[m.k];
    h.py = s.py[m.k];
    h.vx = s.vx[m.k];
    h.vy = s.vy[m.k];
    h.gx = s.gx[m.k];
    h.gy = s.gy[m.k];
    h.rad = s.radius[m.k];
    h.vpref = s.v_pref[m.k];
    h.type = s.type[m.k];
    h.arrival = s.arrival[m.k];
  }
  return h;  // the callers pin() these after issuing every other load
}

// Everything service_commit reads from HBM, loaded BEFORE the first store of the step: vector
// memory operations retire in issue order, so a load issued behind stores waits for them
// (the profile of the first fused kernel showed 73 % of the service wave's cycles in s_waitcnt).
#define EBC_MAXT 4  // observation rows per lane that are preloaded (R <= 4 N); more fall back to late loads
struct CommitPre {
  int cursor;  // pool slot this env restarts from (auto-reset)
  double sx[EBC_MAXT], sy[EBC_MAXT], sr[EBC_MAXT];  // static rows this lane will emit
  // the restart scene (second-stage loads behind `cursor`; issued at the start of the kernel so
  // that they are back long before the restore, and pinned with everything else)
  double px, py, vx, vy, gx, gy, rad, vpref, spx, spy, srad, robot[9];
  int type, n_humans, n_static;
};

template <bool WITH_ROBOT = true>
__device__ __forceinline__ CommitPre preload_commit(const DevState &s, const StepIO &io, const LaneMap &m) {
  CommitPre c;
#pragma unroll
  for (int t = 0; t < EBC_MAXT; ++t) {
    const int r = m.i + t * s.N;
    const bool st = m.env_ok && (io.ob || io.obs_rotated) && r >= m.n && r - m.n < m.ns;
    const size_t q = m.ee * s.S + (st ? r - m.n : 0);
    c.sx[t] = st ? s.spx[q] : 0.0;
    c.sy[t] = st ? s.spy[q] : 0.0;
    c.sr[t] = st ? s.sradius[q] : 0.0;
  }
  c.cursor = 0;
  c.px = c.py = c.vx = c.vy = c.gx = c.gy = c.rad = c.vpref = c.spx = c.spy = c.srad = 0;
  c.type = c.n_humans = c.n_static = 0;
#pragma unroll
  for (int q = 0; q < 9; ++q) c.robot[q] = 0;
  if (io.auto_reset && m.env_ok) {
    const ScenePool &P = s.pool;
    // without a custom pool an env restarts from its own slot: no load in front of the scene's
    c.cursor = P.P > 0 ? P.cursor[m.ee] : (int)m.ee;
    const size_t src = (size_t)c.cursor * s.N + m.i;
    c.n_humans = P.n_humans[c.cursor];
    c.px = P.px[src]; c.py = P.py[src]; c.vx = P.vx[src]; c.vy = P.vy[src];
    c.gx = P.gx[src]; c.gy = P.gy[src]; c.rad = P.radius[src]; c.vpref = P.v_pref[src];
    c.type = P.type[src];
    if (s.S) {
      c.n_static = P.n_static[c.cursor];
      if (m.i < s.S) {  // static rows: lane i carries row i (rows past N - 1 are copied late)
        const size_t q = (size_t)c.cursor * s.S + m.i;
        c.spx = P.spx[q]; c.spy = P.spy[q]; c.srad = P.sradius[q];
      }
    }
    if (WITH_ROBOT && m.leader) {
#pragma unroll
      for (int q = 0; q < 9; ++q) c.robot[q] = P.robot[(size_t)c.cursor * 9 + q];
    }
  }
  return c;
}

__device__ __forceinline__ void pin_loads(HumanRegs &h, CommitPre &c, double (&rb)[9], double &gtime) {
  pin(h.px); pin(h.py); pin(h.vx); pin(h.vy); pin(h.gx); pin(h.gy); pin(h.rad); pin(h.vpref);
  pin(h.arrival); pin(h.type);
  pin(c.cursor);
#pragma unroll
  for (int q = 0; q < 9; ++q) pin(rb[q]);
#pragma unroll
  for (int t = 0; t < EBC_MAXT; ++t) { pin(c.sx[t]); pin(c.sy[t]); pin(c.sr[t]); }
  pin(gtime);
}

// The restart scene is loaded behind `cursor` (one more memory round trip than the state).  It is
// pinned later than the rest: just before service_commit's first store, after the arithmetic of
// the commit, so the extra round trip overlaps that arithmetic instead of heading the chain.
__device__ __forceinline__ void pin_pool(CommitPre &c) {
  pin(c.px); pin(c.py); pin(c.vx); pin(c.vy); pin(c.gx); pin(c.gy); pin(c.rad); pin(c.vpref);
  pin(c.spx); pin(c.spy); pin(c.srad); pin(c.type); pin(c.n_humans); pin(c.n_static);
#pragma unroll
  for (int q = 0; q < 9; ++q) pin(c.robot[q]);
}

// Where service_commit stores, as VGPR addresses fixed at the start of the kernel.
struct CommitAddr {
  double *px, *py, *vx, *vy, *arrival, *robot, *time, *human_action, *ob;
  float4 *tile;
  float *obs;
};
template <bool PIN = true>
__device__ __forceinline__ CommitAddr commit_addr(const DevState &s, const StepIO &io, const LaneMap &m) {
  CommitAddr a;
  a.px = s.px + m.k; a.py = s.py + m.k; a.vx = s.vx + m.k; a.vy = s.vy + m.k;
  a.arrival = s.arrival + m.k;
  a.robot = s.robot + m.ee * 9;
  a.time = s.time + m.ee;
  a.human_action = io.human_action ? io.human_action + m.k * 2 : nullptr;
  a.ob = io.ob ? io.ob + m.ee * (size_t)(s.N + s.S) * 5 : nullptr;
  a.obs = io.obs_rotated;  // row offset depends on T: added by the caller
  a.tile = s.tile + 2 * m.k;
  if (PIN) {
    pin(a.px); pin(a.py); pin(a.vx); pin(a.vy); pin(a.arrival); pin(a.robot); pin(a.time);
    pin(a.human_action); pin(a.ob); pin(a.obs);
    pin(a.tile);
  }
  return a;
}

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// 16-byte device-scope (sc1: L2-served, never from this CU's L1) load / store of a mailbox granule.  The load
// waits for its data inside the statement: the compiler does not track loads issued by asm.
#ifndef EBC_VEL_STORE_SCOPE
#define EBC_VEL_STORE_SCOPE "sc1"
#endif
__device__ __forceinline__ u32x4 load16_device(const uint4 *box) {
  u32x4 v;
#ifdef EBC_VEL_8B  // experiment: the granule as two 8-byte device-scope atomics
  const unsigned long long a = __hip_atomic_load((const unsigned long long *)box, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned long long b = __hip_atomic_load((const unsigned long long *)box + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  v.x = (unsigned)a; v.y = (unsigned)(a >> 32); v.z = (unsigned)b; v.w = (unsigned)(b >> 32);
#else
  asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(box) : "memory");
#endif
  return v;
}
__device__ __forceinline__ void store16_device(uint4 *box, u32x4 v) {
#ifdef EBC_VEL_8B
  __hip_atomic_store((unsigned long long *)box, ((unsigned long long)v.y << 32) | v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store((unsigned long long *)box + 1, ((unsigned long long)v.w << 32) | v.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
  // s_nop: a store of more than 64 bits reads its data registers over several cycles, and a vector instruction that
  // rewrites them right behind it corrupts the lanes read last (12-15 of every 16: seen).  The compiler pads that
  // hazard for stores it emits itself; it cannot see inside an asm statement.
  asm volatile("global_store_dwordx4 %0, %1, off " EBC_VEL_STORE_SCOPE "\n\ts_nop 2" ::"v"(box), "v"(v) : "memory");
#endif
}
// wave-wide: returns once every lane with `need` has found its velocity word tagged with this launch's
// epoch in both halves (or gave up)
__device__ __forceinline__ u32x4 velocity_wait(const uint4 *box, bool need, unsigned epoch, unsigned *fault) {
  u32x4 v = {0u, 0u, 0u, 0u};
  bool waiting = need;
  for (unsigned spins = 0;; ++spins) {
    if (waiting) {
      v = load16_device(box);
      waiting = v.y != epoch || v.w != epoch;
    }
    if (!__any(waiting)) break;
    if (spins > EBC_SPIN_LIMIT) {
      if (waiting) atomicOr(fault, 1u);
      break;
    }
    __builtin_amdgcn_s_sleep(EBC_POLL_SLEEP);
  }
  return v;
}
// the same for a box that holds the launch's epoch when ready
__device__ __forceinline__ void mailbox_wait_epoch(unsigned *box, bool need, unsigned epoch, unsigned *fault) {
  bool waiting = need;
  for (unsigned spins = 0;; ++spins) {
    if (waiting) waiting = __hip_atomic_load(box, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch;
    if (!__any(waiting)) break;
    if (spins > EBC_SPIN_LIMIT) {
      if (waiting) atomicOr(fault, 1u);
      break;
    }
    __builtin_amdgcn_s_sleep(EBC_POLL_SLEEP);
  }
}
template <typename W>
__device__ __forceinline__ void mailbox_put(W *box, W v) {
  __hip_atomic_store(box, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void unpack_velocity(u32x4 v, double &ax, double &ay) {
  ax = (double)__uint_as_float(v.x);  // getAgentVelocity -> Python float
  ay = (double)__uint_as_float(v.z);
}

// The robot's action for this step (its policy, or the caller's) — env.py:388-392.
__device__ __forceinline__ void robot_action(const StepIO &io, size_t ee, const double *rb, double &a0, double &a1) {
  if (io.robot_policy == EBC_ROBOT_LINEAR) {
    linear_policy(rb[0], rb[1], rb[5], rb[6], rb[7], a0, a1);
  } else {
    a0 = io.robot_action[2 * ee];
    a1 = io.robot_action[2 * ee + 1];
  }
}
// Agent.step for the robot (agent.py:202-228): rb becomes the next state.  It depends on the action
// alone, so every role that needs the robot's next state works it out itself instead of waiting
// for the ENV role.
__device__ __forceinline__ void robot_advance(const EbcParams &p, double *rb, double a0, double a1) {
  double nx, ny;
  robot_next_position(rb, p.robot_kinematics, a0, a1, p.time_step, nx, ny);
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
}

// Leader lanes return the step's done flag and leave the robot's NEXT state in rb.
// `epoch` != 0 (fused ORCA step): the leader publishes the robot's next state in s.robot_n as soon
// as the action is known — the ROWS role builds the observation frame from it long before the
// collision / reward work below is done.
#define EBC_ROBOT_STAGE 21  // robots (9 doubles each) the distance scratch stages at once: 3 * 64 / 9
struct EnvScratch {  // LDS of one service_env wave
  double cand[3][EBC_WAVE];  // per collision class: this lane's distance, if it counts
  double ract[EBC_WAVE][2];  // the envs' robot actions
  double gtime[EBC_WAVE];    // per env: global time, parked from its early load to the reward at the end
  int n[EBC_WAVE];           // per env: its number of humans, parked for the leader's walk over them
  // second form of the step (env2_role): more of what only an env's leader lane needs at the end
  double goal[EBC_WAVE][2];
  int slot[EBC_WAVE], cursor[EBC_WAVE];
};
// humans(): the lane's human (position, velocity, radius, type) — asked for AFTER the robot's action has been
// worked out and published.  The fused step's ENV role loads its humans only then: double-precision
// atan2 / sin / cos beside ten live human registers did not fit its 80-register budget (spills), and the
// ENV waves are not what the launch waits for; the one-wave step kernel hands over preloaded registers.
template <typename Humans>
__device__ __forceinline__ int service_env(const EbcParams &p_early, const DevState &s_early, const StepIO &io_early,
                                           const LaneMap &m, Humans humans, double rb[9],
                                           double gtime, int lane, EnvScratch &X, unsigned epoch = 0, int e0 = 0) {
  int coll_t[3], grid_slot;
  {
  const EbcParams &p = p_early;
  const DevState &s = s_early;
  const StepIO &io = io_early;
  double (&sh_cand)[3][EBC_WAVE] = X.cand;
  double (&sh_ract)[EBC_WAVE][2] = X.ract;
  grid_slot = (m.leader && s.pool.grid) ? s.grid_scene[m.ee] : 0;
  pin(grid_slot);
  const int epb_env = EBC_WAVE / s.N;                     // envs per wave
  if (m.leader) {
    X.gtime[m.el] = gtime;
    X.n[m.el] = m.n;
  }
  // which lanes carry a human, as a wave mask in scalar registers: m.n need not stay in a vector register
  const unsigned long long active_mask = __ballot(m.active);
  if (m.leader) {
    double a0, a1;
    if (io.robot_policy == EBC_ROBOT_LINEAR) {
      linear_policy(rb[0], rb[1], rb[5], rb[6], rb[7], a0, a1);
    } else {
      a0 = io.robot_action[2 * m.ee];
      a1 = io.robot_action[2 * m.ee + 1];
    }
    EBC_MARK(1);
    sh_ract[m.el][0] = a0;
    sh_ract[m.el][1] = a1;
    if (epoch) {
      double rn[9];
#pragma unroll
      for (int c = 0; c < 9; ++c) rn[c] = rb[c];
      robot_advance(p, rn, a0, a1);
      // The robots' next states leave as ONE run of device-scope stores for the whole wave (lane = (env, field):
      // the envs of a wave are consecutive, so the run is contiguous) instead of nine 8-byte fabric writes
      // per env from the leader lanes: staged through the distance scratch, which is not in use yet.
      // robot_n is what the ROWS role builds this step's observation frame from (after robot_ready) and
      // what the next step reads; a restart overwrites it later in this launch — only after ROWS has it
      // (rows_loaded).  (An agent-scope release FENCE instead of device-scope stores writes back the whole
      // L2 of the XCD on this part: it doubled the step time.)
      if (m.el < EBC_ROBOT_STAGE) {
#pragma unroll
        for (int c = 0; c < 9; ++c) (&sh_cand[0][0])[m.el * 9 + c] = rn[c];
      } else {  // N <= 2: more envs per wave than the scratch stages at once (rare shape)
        double *o = s.robot_n + m.ee * 9;
#pragma unroll
        for (int c = 0; c < 9; ++c) __hip_atomic_store(o + c, rn[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
  wave_sync();
  if (epoch) {
    const int envs_here = min(min(epb_env, EBC_ROBOT_STAGE), s.E - e0);
    for (int q = lane; q < 9 * envs_here; q += EBC_WAVE)
      __hip_atomic_store(s.robot_n + (size_t)e0 * 9 + q, (&sh_cand[0][0])[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    wave_sync();  // the scratch is the distance scratch again
  }
  const HumanRegs h = humans();
  const double a0 = sh_ract[m.env_ok ? m.el : 0][0], a1 = sh_ract[m.env_ok ? m.el : 0][1];
  double rvx, rvy;
  if (p.robot_kinematics == EBC_HOLONOMIC) {
    rvx = a0;
    rvy = a1;
  } else {
    rvx = a0 * cos(a1 + rb[8]);
    rvy = a0 * sin(a1 + rb[8]);
  }
  // ordered per-type reduction with break at the first hit (env.py:303-313), without the serial
  // walk: per type, the first colliding human of the env comes out of a ballot; a human counts
  // for the type's minimum distance when it lies before that one
  const bool active = (active_mask >> lane) & 1;
  const double dt_now = read_late(p_early).time_step;
  const double d = active ? closest_dist(h.px, h.py, h.vx, h.vy, h.rad, rb[0], rb[1], rb[4], rvx, rvy, dt_now) : 0.0;
  const bool hit = active && d < 0;
  const int base = lane - m.i;  // first lane of this env
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    const unsigned long long hm = (__ballot(hit && h.type == t) >> base) & (s.N >= 64 ? ~0ull : ((1ull << s.N) - 1));
    const int first = hm ? __ffsll((long long)hm) - 1 : s.N;
    coll_t[t] = hm != 0;
    sh_cand[t][lane] = (active && h.type == t && !hit && m.i < first) ? d : INFINITY;
  }
  wave_sync();
  EBC_MARK(2);
  if (epoch) {  // the stores of robot_n went out a distance computation ago: this wait is short
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (m.leader) mailbox_put(s.robot_ready + m.ee, epoch);
  }
  }
  if (!m.leader) return 0;
  const double a0 = X.ract[m.el][0], a1 = X.ract[m.el][1];
  const double dt = read_late(p_early).time_step;
  double (&sh_cand)[3][EBC_WAVE] = X.cand;
  // the leader's tail reads its arguments HERE (read_late), not at the top of the wave
  const EbcParams &p = read_late(p_early);
  const DevState &s = read_late(s_early);
  const StepIO &io = read_late(io_early);
  gtime = read_late(X).gtime[m.el];
  const int n_env = read_late(X).n[m.el];
  double dm0 = INFINITY, dm1 = INFINITY, dm2 = INFINITY;
  const int c0 = coll_t[0], c1 = coll_t[1], c2 = coll_t[2];
  for (int q = 0; q < n_env; ++q) {  // independent loads: no branch between them
    const double d0 = sh_cand[0][lane + q], d1 = sh_cand[1][lane + q], d2 = sh_cand[2][lane + q];
    dm0 = d0 < dm0 ? d0 : dm0;
    dm1 = d1 < dm1 ? d1 : dm1;
    dm2 = d2 < dm2 ? d2 : dm2;
  }
  const double dmin[3] = {dm0, dm1, dm2};
  EBC_MARK(3);
  double nx, ny;
  robot_next_position(rb, p.robot_kinematics, a0, a1, dt, nx, ny);
  int coll[4] = {c0, c1, c2, 0};
  coll[3] = grid_collision(s.pool.grid ? s.pool.grid + (size_t)grid_slot * s.G * 2 : nullptr, s.G, p.map_size_m,
                           p.map_resolution, nx, ny, rb[4], io.has_border ? io.border : nullptr);
  EBC_MARK(4);
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
  s.done[m.ee] = (uint8_t)ro.done;
  if (io.reward) io.reward[m.ee] = ro.reward;
  if (io.done) io.done[m.ee] = (uint8_t)ro.done;
  if (io.info) io.info[m.ee] = (uint8_t)ro.info;
  if (io.dmin) {
    io.dmin[3 * m.ee] = dm0;
    io.dmin[3 * m.ee + 1] = dm1;
    io.dmin[3 * m.ee + 2] = dm2;
  }
  if (io.dist_to_goal) io.dist_to_goal[m.ee] = ro.dist_to_goal;
  if (io.robot_action_out) {
    io.robot_action_out[2 * m.ee] = a0;
    io.robot_action_out[2 * m.ee + 1] = a1;
  }
  return ro.done;
}

// The state the next step will see: the moved humans and their float tile entries, or — after a
// terminal step under auto-reset — the env's next scene.
__device__ __forceinline__ void commit_state_tail(const EbcParams &p, const DevState &s, const LaneMap &m,
                                                  const HumanRegs &h, const CommitPre &pre,
                                                  const CommitAddr &A, bool restore) {
  const int N = s.N, S = s.S;
  // the humans: the moved state ...
  if (m.active && !restore) {
    *A.px = h.px;
    *A.py = h.py;
    *A.vx = h.vx;
    *A.vy = h.vy;
    *A.arrival = h.arrival;
    float prefx, prefy;  // the float tile entry (store_tile), through the pinned addresses
    orca_pref_velocity(h.px, h.py, h.gx, h.gy, prefx, prefy);
    A.tile[0] = make_float4((float)h.px, (float)h.py, (float)h.vx, (float)h.vy);
    A.tile[1] = make_float4((float)(h.rad + 0.01 + p.orca_safety_space), (float)h.vpref, prefx, prefy);
  }
  // ... or, after a terminal step under auto-reset, the env's next scene from the pool: every
  // per-scene field (ragged human count, goals, radii, static rows, map), time 0.  Rare path: its
  // loads sit behind the step's stores on purpose.
  if (restore) {
    const ScenePool &P = s.pool;
    const int n_new = pre.n_humans;
    const bool live = m.i < n_new;
    const double npx = live ? pre.px : 0.0, npy = live ? pre.py : 0.0;
    const double nvx = live ? pre.vx : 0.0, nvy = live ? pre.vy : 0.0;
    const double ngx = live ? pre.gx : 0.0, ngy = live ? pre.gy : 0.0;
    const double nrad = live ? pre.rad : 0.0, nvp = live ? pre.vpref : 0.0;
    *A.px = npx; *A.py = npy; *A.vx = nvx; *A.vy = nvy; *A.arrival = 0.0;
    s.gx[m.k] = ngx; s.gy[m.k] = ngy; s.radius[m.k] = nrad; s.v_pref[m.k] = nvp;
    s.type[m.k] = live ? (uint8_t)pre.type : (uint8_t)0;
    if (live) {
      store_tile(p, s, m.k, npx, npy, nvx, nvy, ngx, ngy, nrad, nvp);
    } else {
      A.tile[0] = make_float4(0, 0, 0, 0);
      A.tile[1] = make_float4(0, 0, 0, 0);
    }
    if (m.i < S) {
      s.spx[m.ee * S + m.i] = pre.spx;
      s.spy[m.ee * S + m.i] = pre.spy;
      s.sradius[m.ee * S + m.i] = pre.srad;
    }
    for (int q = m.i + N; q < S; q += N) {  // more static rows than humans: late loads (rare)
      const size_t c = (size_t)pre.cursor * S + q;
      s.spx[m.ee * S + q] = P.spx[c];
      s.spy[m.ee * S + q] = P.spy[c];
      s.sradius[m.ee * S + q] = P.sradius[c];
    }
    if (m.leader) {
      s.n_humans[m.ee] = n_new;
      if (S) s.n_static[m.ee] = pre.n_static;
      s.grid_scene[m.ee] = pre.cursor;
#pragma unroll
      for (int q = 0; q < 9; ++q) A.robot[q] = pre.robot[q];
      *A.time = 0.0;
      if (P.P > 0) {  // walk the custom pool; without one the env keeps restarting from its own slot
        int nxt = pre.cursor - s.E + P.stride;
        P.cursor[m.ee] = s.E + (nxt >= P.P ? nxt % P.P : nxt);
      }
    }
  }
}

// rbn: the robot's next state (in registers); (ax, ay): this human's velocity.
template <int T>
__device__ __forceinline__ void service_commit(const EbcParams &p, const DevState &s, const StepIO &io,
                                               const LaneMap &m, HumanRegs h, CommitPre pre,
                                               const CommitAddr &A, const double (&rbn)[9],
                                               bool restore, double tnew, double ax, double ay) {
  const int N = s.N, S = s.S, R = N + S;
  const double dt = p.time_step;
  if (m.active) {  // Agent.step (agent.py:202-211), first arrival (env.py:365-378)
    h.px = h.px + ax * dt;
    h.py = h.py + ay * dt;
    h.vx = ax;
    h.vy = ay;
    if (h.arrival == 0 && norm2(h.px - h.gx, h.py - h.gy) < h.rad) h.arrival = tnew;
  } else {
    ax = 0;
    ay = 0;
  }
  const RotFrame f = rot_frame(rbn, p.rotate_unicycle);
  pin_pool(pre);  // first store of the commit below
  if (m.leader && !restore) {  // the robot: the moved state (the restart scene: restore path)
#pragma unroll
    for (int c = 0; c < 9; ++c) A.robot[c] = rbn[c];
    *A.time = tnew;
  }
  if (A.human_action) {
    A.human_action[0] = ax;
    A.human_action[1] = ay;
  }
  // returned observation: humans then static rows, raw and in the robot frame
  if (io.ob || io.obs_rotated) {
    int t = 0;
    for (int r = m.i; r < R; r += N, ++t) {
      double opx = 0, opy = 0, ovx = 0, ovy = 0, orad = 0;
      int otype = 0;
      bool valid = false;
      if (r < m.n) {  // r == i: this lane's own human
        opx = h.px; opy = h.py; ovx = h.vx; ovy = h.vy; orad = h.rad; otype = h.type;
        valid = true;
      } else if (r - m.n < m.ns) {
        if (t < EBC_MAXT) {
          opx = t == 0 ? pre.sx[0] : t == 1 ? pre.sx[1] : t == 2 ? pre.sx[2] : pre.sx[3];
          opy = t == 0 ? pre.sy[0] : t == 1 ? pre.sy[1] : t == 2 ? pre.sy[2] : pre.sy[3];
          orad = t == 0 ? pre.sr[0] : t == 1 ? pre.sr[1] : t == 2 ? pre.sr[2] : pre.sr[3];
        } else {
          const size_t q = m.ee * S + (r - m.n);
          opx = s.spx[q]; opy = s.spy[q]; orad = s.sradius[q];
        }
        otype = EBC_ADULT_STATIC;
        valid = true;
      }
      if (A.ob) {
        double *o = A.ob + (size_t)r * 5;
        o[0] = opx; o[1] = opy; o[2] = ovx; o[3] = ovy; o[4] = orad;
      }
      if (A.obs) {
        float out[T];
        if (valid) {
          rotate_row<T>(f, opx, opy, ovx, ovy, orad, otype, out);
        } else {
#pragma unroll
          for (int c = 0; c < T; ++c) out[c] = 0.0f;
        }
        float *o = A.obs + (m.ee * R + r) * T;
#pragma unroll
        for (int c = 0; c < T; ++c) o[c] = out[c];
      }
    }
  }
  commit_state_tail(p, s, m, h, pre, A, restore);
}

__device__ __forceinline__ void load_robot(const DevState &s, const LaneMap &m, double rb[9]) {
  const double *rb_in = s.robot + m.ee * 9;
#pragma unroll
  for (int c = 0; c < 9; ++c) rb[c] = m.env_ok ? rb_in[c] : 0.0;
}

// Service-only step (humans on the linear policy, or velocities supplied / cached in hact): ONE
// launch of one-wave workgroups, service_env then service_commit in the same wave.
template <int POLICY, int T>
__global__ __launch_bounds__(EBC_WAVE) void step_kernel(EbcParams p_in, DevState s_in, StepIO io_in) {
  const WaveTrace wt(1);
  const int lane = threadIdx.x;
  __shared__ ArgBlock args;
  __shared__ EnvScratch env_scratch;
  __shared__ double sh_rbn[EBC_WAVE][9];
  stage_args(&args.p, p_in, lane);
  stage_args(&args.s, s_in, lane);
  stage_args(&args.io, io_in, lane);
  wave_sync();
  const EbcParams &p = args.p;
  const DevState &s = args.s;
  const StepIO &io = args.io;
  const int epb = EBC_WAVE / s.N;
  const LaneMap m = lane_map(s, blockIdx.x * epb, epb, lane);
  double rb[9];
  load_robot(s, m, rb);
  double gtime = m.env_ok ? s.time[m.ee] : 0.0;
  HumanRegs h = load_human(s, m);
  CommitPre pre = preload_commit(s, io, m);
  const CommitAddr A = commit_addr(s, io, m);
  double ax = 0, ay = 0;
  if (POLICY == EBC_HUMAN_LINEAR) {
    if (m.active) linear_policy(h.px, h.py, h.gx, h.gy, h.vpref, ax, ay);  // pre-step state (env.py:393-405)
  } else if (m.env_ok) {
    ax = s.hact[m.k * 2];
    ay = s.hact[m.k * 2 + 1];
  }
  pin_loads(h, pre, rb, gtime);  // every load of the step is back before its first store
  pin(ax);
  pin(ay);
  int done_flag = service_env(p, s, io, m, [&]() { return h; }, rb, gtime, lane, env_scratch);
  if (m.leader) {
#pragma unroll
    for (int c = 0; c < 9; ++c) sh_rbn[m.el][c] = rb[c];
  }
  // every lane of an env learns whether the step was terminal (leader = first lane of the env)
  done_flag = __shfl(done_flag, lane - m.i, EBC_WAVE);
  wave_sync();  // sh_rbn
  if (!m.env_ok) return;
  double rbn[9];
#pragma unroll
  for (int c = 0; c < 9; ++c) rbn[c] = sh_rbn[m.el][c];
  service_commit<T>(p, s, io, m, h, pre, A, rbn, io.auto_reset && done_flag, gtime + p.time_step, ax, ay);
}

// ORCA step: ONE launch of one-wave workgroups in four roles (orca_step_kernel).
//
//   ENV    service_env (6 envs per wave at N = 10): the robot side of the step, reward / done /
//          info.  The longest dependent chain, so these blocks come first.    -> env_done
//   ORCA   64 / GS humans per wave                                            -> vel
//   ROWS   lane = observation row slot: reads the pre-step state, tells STATE so (rows_loaded),
//          takes the robot's next state (robot_n, after robot_ready) and the human's velocity (vel),
//          emits the raw and the rotated row
//   STATE  lane = human slot: takes vel and env_done, moves the humans, writes the state
//          and float tile of the next step (or the restart scene)
//
// Everything in a step is local to an env, so the roles meet through per-env / per-human mailbox
// words in HBM instead of a second launch: an env's rows and state are written as soon as ITS
// humans and ITS robot-side work are done, while the slowest ORCA waves of the launch (the
// humans deep in linearProgram3) are still running.  Consumers (ROWS, STATE) only wait for
// workgroups of LOWER index; the dispatcher starts workgroups in index order and producers wait
// for nothing, so every wait ends.  A poll that exceeds EBC_SPIN_LIMIT gives up and raises
// DevState::fault rather than hang the device.
struct StepGrid {
  unsigned env_blocks, orca_blocks, rows_blocks;  // then env_blocks STATE blocks
  unsigned rows_epw;                               // envs per ROWS wave (1 when N + S > 64)
  unsigned epoch;                                  // launch counter, never 0
  unsigned total;                                  // one-wave "blocks" in all (EBC_STEP_WPB of them per workgroup)
};

#define EBC_RBN_ENVS 16  // envs per STATE wave whose restart robot is parked in LDS (more: late loads)
struct RoleLds {  // after the arguments, what the role of the wave needs
  ArgBlock args;
  union {
    double restart_robot[EBC_RBN_ENVS][9];            // STATE
    EnvScratch env;                                   // ENV
  };
};

// ---- ENV
__device__ __forceinline__ void env_role(const EbcParams &p_in, const DevState &s_in, const StepIO &io_in,
                                         RoleLds &L, int block, unsigned epoch, int lane) {
  stage_args(&L.args.p, p_in, lane);
  stage_args(&L.args.s, s_in, lane);
  stage_args(&L.args.io, io_in, lane);
  wave_sync();
  const EbcParams &p = L.args.p;
  const DevState &s = L.args.s;
  const StepIO &io = L.args.io;
  const int epb = EBC_WAVE / s.N;
  const LaneMap m = lane_map(s, block * epb, epb, lane);
  double rb[9];
  load_robot(s, m, rb);
  double gtime = m.env_ok ? s.time[m.ee] : 0.0;
  pin(gtime);
#pragma unroll
  for (int q = 0; q < 9; ++q) pin(rb[q]);
  EBC_MARK(0);
  auto humans = [&]() {
    HumanRegs h = load_human(read_late(s), m);
    pin(h.px); pin(h.py); pin(h.vx); pin(h.vy); pin(h.rad); pin(h.type);
    return h;
  };
  const int done = service_env(p, s, io, m, humans, rb, gtime, lane, L.env, epoch, __builtin_amdgcn_readfirstlane(block * epb));
  if (m.leader) mailbox_put(read_late(s).env_done + m.ee, ((unsigned long long)epoch << 32) | (1u + (unsigned)done));
}

// Have a kernel argument in a scalar register HERE.  The compiler otherwise loads each field where
// it is first used, inside the branch that uses it: four or five scalar-memory round trips in a
// row before an ORCA wave issues its first vector load.
template <typename Tp>
__device__ __forceinline__ void sreg(const Tp &x) {
  asm volatile("" ::"s"(x));
}

// ---- ORCA
template <int GS>
__device__ __forceinline__ void orca_role(const EbcParams &p, const DevState &s, const OrcaHot &hot, uint4 *vel, unsigned epoch,
                                          unsigned char *scratch, unsigned block, int lane) {
  // `hot`, vel, epoch: preloaded kernel arguments; the rest is asked for now, all at once
  sreg(s.robot); sreg(s.range_sq); sreg(s.inv_time_horizon); sreg(s.inv_time_step);
  sreg(p.robot_visible); sreg(p.orca_max_neighbors); sreg(p.orca_safety_space);
  constexpr int HPW = EBC_WAVE / GS;
  const int group = lane / GS, j = lane - group * GS;
  // 32-bit index math (E * N < 2^31 is checked at create)
  const unsigned hh = block * HPW + group;
  const bool h_ok = group < HPW && hh < (unsigned)hot.E * (unsigned)hot.N;  // lanes past HPW * GS idle
  int e, i;
  split_human(hot, hh, h_ok, e, i);
  float ox, oy;
  bool human_ok;
  orca_wave<GS>(p, s, hot, h_ok, e, i, scratch, ox, oy, human_ok);
#ifdef EBC_WAVE_TRACE  // measurement / test build only: one producer "forgets" its hand-off (tests the give-up path)
  if (h_ok && (int)hh == g_withhold_human) return;
#endif
  if (h_ok && j == 0) {  // one 16-byte device-scope store: the velocity, tagged with this launch's epoch in both halves
    const u32x4 w = {human_ok ? __float_as_uint(ox) : 0u, epoch, human_ok ? __float_as_uint(oy) : 0u, epoch};
    store16_device(vel + hh, w);
  }
}

// ---- ROWS: lane = row slot of an env.  Slots [0, N) are the human slots, [N, N + S) the static
// slots, so no load waits for n_humans; the ROW a slot emits is: human i < n -> i, static j < ns ->
// n + j, and the padding rows (zeros) are shared out over the unused slots: human slot i >= n ->
// ns + i, static slot j >= ns -> N + j.
template <int T>
__device__ __forceinline__ void rows_role(const EbcParams &p_in, const DevState &s_in, const StepIO &io_in,
                                          RoleLds &L, unsigned block, unsigned epw, unsigned epoch, int lane) {
  stage_args(&L.args.p, p_in, lane);
  stage_args(&L.args.s, s_in, lane);
  stage_args(&L.args.io, io_in, lane);
  wave_sync();
  const EbcParams &p = L.args.p;
  const DevState &s = L.args.s;
  const StepIO &io = L.args.io;
  const int N = s.N, S = s.S, R = N + S;
  const int el = R <= EBC_WAVE ? lane / R : 0;
  const int first = lane - el * R;
  const int e = (int)(block * epw) + el;
  const bool env_ok = el < (int)epw && e < s.E;
  const size_t ee = env_ok ? (size_t)e : 0;
  const int stride = R <= EBC_WAVE ? R : EBC_WAVE;  // one pass unless an env has more rows than lanes
  RotFrame f = {};
  int n = env_ok ? s.n_humans[ee] : 0;
  int ns = (env_ok && S) ? s.n_static[ee] : 0;
  for (int slot = first; slot < R; slot += stride) {
    const bool human = slot < N;
    const size_t k = ee * N + (human ? slot : 0), q = ee * (size_t)(S ? S : 1) + (human ? 0 : slot - N);
    double opx = 0, opy = 0, ovx = 0, ovy = 0, orad = 0;
    int otype = 0;
    if (env_ok && human) {
      opx = s.px[k]; opy = s.py[k]; orad = s.radius[k]; otype = s.type[k];
    } else if (env_ok) {
      opx = s.spx[q]; opy = s.spy[q]; orad = s.sradius[q]; otype = EBC_ADULT_STATIC;
    }
    const uint4 *vbox = s.vel + k;
    pin(opx); pin(opy); pin(orad); pin(otype); pin(vbox); pin(n); pin(ns);
    if (slot == first) {
      // the robot's next state: the ENV role publishes it early in its run (service_env)
      mailbox_wait_epoch(s.robot_ready + ee, env_ok, epoch, s.fault);
      double rb[9];  // device-scope loads, issued after the flag was seen: they do not come from a stale cache line
#pragma unroll
      for (int c = 0; c < 9; ++c)
        rb[c] = env_ok ? __hip_atomic_load(s.robot_n + ee * 9 + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
      f = rot_frame(rb, p.rotate_unicycle);
      pin(f.px); pin(f.py); pin(f.dg);
    }
    // every pre-step value this env's rows need AND the robot's next state are in registers: STATE may
    // overwrite the state, and (restart) robot_n
    if (env_ok && slot + stride >= R && first == 0) mailbox_put(s.rows_loaded + ee, epoch);
    const bool valid = human ? slot < n : slot - N < ns;
    const int row = human ? (valid ? slot : ns + slot) : (valid ? n + slot - N : slot);
    auto emit = [&]() {
      if (!valid) opx = opy = ovx = ovy = orad = 0.0;
      if (io.ob) {
        double *o = io.ob + (ee * R + row) * 5;
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
        // streaming stores: nobody on the device reads the rows, and what is not left dirty in the
        // L2s does not have to be written back before the next launch may start (-0.2 us per step)
        // (a lane's T floats are consecutive: the compiler stores them as 16-byte vectors)
        float *o = io.obs_rotated + (ee * R + row) * T;
#pragma unroll
        for (int c = 0; c < T; ++c) __builtin_nontemporal_store(out[c], o + c);
      }
    };
    // Static rows and padding need nothing from the ORCA waves: they leave NOW, long before the humans' velocities
    // arrive — 8 of the 18 rows of the bench workload that are then not part of the burst of writes at the end of the
    // launch, behind which the last stores of the launch queue (-0.22 us per step, profiles/r02_early_rows_ab.txt).
    const bool moving = human && valid;
    if (env_ok && !moving) emit();
    const u32x4 v = velocity_wait(vbox, env_ok && human, epoch, s.fault);
    if (env_ok && moving) {  // Agent.step (agent.py:202-211)
      double ax, ay;
      unpack_velocity(v, ax, ay);
      opx = opx + ax * p.time_step;
      opy = opy + ay * p.time_step;
      ovx = ax;
      ovy = ay;
      emit();
    }
  }
}

// ---- STATE
__device__ __forceinline__ void state_role(const EbcParams &p_in, const DevState &s_in, const StepIO &io_in,
                                           RoleLds &L, int block, bool wait_rows, unsigned epoch, int lane) {
  stage_args(&L.args.p, p_in, lane);
  stage_args(&L.args.s, s_in, lane);
  stage_args(&L.args.io, io_in, lane);
  wave_sync();
  const EbcParams &p = L.args.p;
  const DevState &s = L.args.s;
  const StepIO &io = L.args.io;
  const int epb = EBC_WAVE / s.N;
  const LaneMap m = lane_map(s, block * epb, epb, lane);
  double gtime = m.env_ok ? s.time[m.ee] : 0.0;
  HumanRegs h = load_human(s, m);
  StepIO io_state = io;  // no observation rows here: CommitPre's static-row preloads stay empty
  io_state.ob = nullptr;
  io_state.obs_rotated = nullptr;
  const uint4 *vbox = s.vel + m.k;
  unsigned long long *done_box = s.env_done + m.ee;
  unsigned *loaded_box = s.rows_loaded + m.ee;
  // The restart scene while there is nothing else to do (behind the cursor when a custom pool is
  // installed), its preferred velocity (worked out when the pool was uploaded) and, parked in LDS,
  // the restart robot.
  CommitPre pre = preload_commit<false>(s, io_state, m);
  float rprefx = 0, rprefy = 0;
  const bool parked = m.el < EBC_RBN_ENVS;
  if (io.auto_reset && m.env_ok) {
    const float2 rp = s.pool.pref[(size_t)pre.cursor * s.N + m.i];
    rprefx = rp.x;
    rprefy = rp.y;
    if (m.leader && parked) {
      const double *src = s.pool.robot + (size_t)pre.cursor * 9;
#pragma unroll
      for (int q = 0; q < 9; ++q) L.restart_robot[m.el][q] = src[q];
    }
  }
  const float tile_rad = (float)(h.rad + 0.01 + p.orca_safety_space), tile_max = (float)h.vpref;  // orca.py:116-126
  pin(vbox); pin(done_box); pin(loaded_box);
  pin(h.px); pin(h.py); pin(h.gx); pin(h.gy); pin(h.rad); pin(h.arrival); pin(gtime);
  pin(pre.cursor); pin(pre.px); pin(pre.py); pin(pre.vx); pin(pre.vy); pin(pre.gx); pin(pre.gy); pin(pre.rad);
  pin(pre.vpref); pin(pre.spx); pin(pre.spy); pin(pre.srad); pin(pre.type); pin(pre.n_humans); pin(pre.n_static);
  pin(rprefx); pin(rprefy);
  wave_sync();  // restart_robot
  EBC_MARK(0);
  // wait for this wave's humans (every ORCA group of these envs has then read the tile and the
  // robot), for the envs' robot-side result, and for the ROWS waves to have read the old state
  // (one loop for the three boxes: a poll is a memory round trip, three in a row were 2 us of the step)
  u32x4 v = {0u, 0u, 0u, 0u};
  unsigned d = 0;
  {
    bool wv = m.env_ok, wd = m.env_ok, wr = m.env_ok && wait_rows;
    for (unsigned spins = 0;; ++spins) {
      unsigned long long dw = 0;
      unsigned r = epoch;
      // the two small loads first; the 16-byte load's wait (inside load16_device) then covers all three
      if (wd) dw = __hip_atomic_load(done_box, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (wr) r = __hip_atomic_load(loaded_box, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (wv) v = load16_device(vbox);
      wv = wv && (v.y != epoch || v.w != epoch);
      if (wd && (unsigned)(dw >> 32) == epoch) {
        d = (unsigned)dw;
        wd = false;
      }
      wr = wr && r != epoch;
      if (!__any(wv || wd || wr)) break;
      if (spins > EBC_SPIN_LIMIT) {
        if (wv || wd || wr) atomicOr(s.fault, 1u);
        break;
      }
      __builtin_amdgcn_s_sleep(EBC_POLL_SLEEP);
    }
  }
  EBC_MARK(1);
  if (!m.env_ok) return;
  const bool restore = io.auto_reset && d == 2u;
  const double tnew = gtime + p.time_step;
  const int N = s.N, S = s.S;
  float4 *tile = s.tile + 2 * m.k;
  double ax = 0, ay = 0;
  if (m.active) unpack_velocity(v, ax, ay);
  if (io.human_action) {
    io.human_action[m.k * 2] = ax;
    io.human_action[m.k * 2 + 1] = ay;
  }
  if (!restore) {
    if (m.active) {  // Agent.step (agent.py:202-211), first arrival (env.py:365-378)
      h.px = h.px + ax * p.time_step;
      h.py = h.py + ay * p.time_step;
      // one norm serves the arrival test and the preferred velocity of the float tile
      // (orca.py:136-140): |position - goal| == |goal - position| bit for bit
      const double dx = h.gx - h.px, dy = h.gy - h.py;
      const double dist = norm2(dx, dy);
      const bool arrived_now = h.arrival == 0 && dist < h.rad;
      if (arrived_now) h.arrival = tnew;
      float prefx, prefy;
      orca_pref_from(dx, dy, dist, prefx, prefy);
      typedef float v4f __attribute__((ext_vector_type(4)));
      const v4f t0 = {(float)h.px, (float)h.py, (float)ax, (float)ay}, t1 = {tile_rad, tile_max, prefx, prefy};
      // streaming stores: nothing reads these again in this launch, and what is not left dirty in the L2s is not
      // waiting for the write-back at the end of the launch (-0.2 us per step, profiles/r02_state_nt_ab.txt)
      __builtin_nontemporal_store(h.px, s.px + m.k);
      __builtin_nontemporal_store(h.py, s.py + m.k);
      __builtin_nontemporal_store(ax, s.vx + m.k);
      __builtin_nontemporal_store(ay, s.vy + m.k);
      if (arrived_now) __builtin_nontemporal_store(h.arrival, s.arrival + m.k);  // once per episode, not every step
      __builtin_nontemporal_store(t0, reinterpret_cast<v4f *>(tile));
      __builtin_nontemporal_store(t1, reinterpret_cast<v4f *>(tile + 1));
    }
    if (m.leader) s.time[m.ee] = tnew;  // the robot's next state is in robot_n already (ENV)
  } else {
    // a terminal env under auto-reset takes its next scene: every per-scene field (ragged human
    // count, goals, radii, static rows, map), time 0
    const ScenePool &P = s.pool;
    const bool live = m.i < pre.n_humans;
    s.px[m.k] = live ? pre.px : 0.0;
    s.py[m.k] = live ? pre.py : 0.0;
    s.vx[m.k] = live ? pre.vx : 0.0;
    s.vy[m.k] = live ? pre.vy : 0.0;
    s.arrival[m.k] = 0.0;
    s.gx[m.k] = live ? pre.gx : 0.0;
    s.gy[m.k] = live ? pre.gy : 0.0;
    s.radius[m.k] = live ? pre.rad : 0.0;
    s.v_pref[m.k] = live ? pre.vpref : 0.0;
    s.type[m.k] = live ? (uint8_t)pre.type : (uint8_t)0;
    if (live) {
      tile[0] = make_float4((float)pre.px, (float)pre.py, (float)pre.vx, (float)pre.vy);
      tile[1] = make_float4((float)(pre.rad + 0.01 + p.orca_safety_space), (float)pre.vpref, rprefx, rprefy);
    } else {
      tile[0] = make_float4(0, 0, 0, 0);
      tile[1] = make_float4(0, 0, 0, 0);
    }
    if (m.i < S) {
      s.spx[m.ee * S + m.i] = pre.spx;
      s.spy[m.ee * S + m.i] = pre.spy;
      s.sradius[m.ee * S + m.i] = pre.srad;
    }
    for (int q = m.i + N; q < S; q += N) {  // more static rows than humans: late loads (rare)
      const size_t c = (size_t)pre.cursor * S + q;
      s.spx[m.ee * S + q] = P.spx[c];
      s.spy[m.ee * S + q] = P.spy[c];
      s.sradius[m.ee * S + q] = P.sradius[c];
    }
    if (m.leader) {
      s.n_humans[m.ee] = pre.n_humans;
      if (S) s.n_static[m.ee] = pre.n_static;
      s.grid_scene[m.ee] = pre.cursor;
      double *rdst = s.robot_n + m.ee * 9;  // where the next step reads the robot: overwrites ENV's result
      if (parked) {
#pragma unroll
        for (int q = 0; q < 9; ++q) rdst[q] = L.restart_robot[m.el][q];
      } else {
        const double *src = P.robot + (size_t)pre.cursor * 9;
        for (int q = 0; q < 9; ++q) rdst[q] = src[q];
      }
      s.time[m.ee] = 0.0;
      if (P.P > 0) {  // walk the custom pool; without one the env keeps restarting from its own slot
        int nxt = pre.cursor - s.E + P.stride;
        P.cursor[m.ee] = s.E + (nxt >= P.P ? nxt % P.P : nxt);
      }
    }
  }
  EBC_MARK(2);
}

// Waves per SIMD the register budget is set for.  6 (80 VGPRs) holds every role without spilling
// up to 9-lane groups; wider groups unroll a longer ranking loop and get 5 (96): a spilled register
// costs every wave of the launch its scratch set-up, one wave per SIMD less costs the late starters.
#ifndef EBC_STEP_WAVES
#define EBC_STEP_WAVES(GS) ((GS) <= 9 ? 6 : 5)
#endif
// Waves per workgroup of the step launch.  Every wave is a "block" of its own (roles by index, no
// workgroup barrier, a private LDS slice); packing several into one workgroup only changes how fast
// the dispatcher starts them.
#ifndef EBC_STEP_WPB
#define EBC_STEP_WPB 1
#endif
__device__ __forceinline__ void env_lane_role(const EbcParams &p_in, const DevState &s_in, const StepIO &io_in, RoleLds &L,
                                              unsigned block, unsigned epoch, int lane);

template <int GS, int T>
__global__ __launch_bounds__(EBC_WAVE * EBC_STEP_WPB, EBC_STEP_WAVES(GS)) void orca_step_kernel(
    // 14 dwords the wave finds in scalar registers when it starts (kernarg preload): its role and, for
    // an ORCA wave, everything up to its first vector load
    unsigned env_blocks, unsigned orca_blocks, int hot_E, int hot_N, unsigned hot_magic, unsigned hot_shift,
    const float4 *hot_tile, const int *hot_n_humans, uint4 *hot_vel, unsigned hot_epoch, unsigned hot_pad,
    EbcParams p_in, DevState s_in, StepIO io_in, StepGrid g) {
  const WaveTrace wt(2);
  constexpr size_t LDS = ((sizeof(RoleLds) > (size_t)OrcaLds<GS>::BYTES ? sizeof(RoleLds) : (size_t)OrcaLds<GS>::BYTES) + 15) / 16 * 16;
  __shared__ __align__(16) unsigned char lds_all[EBC_STEP_WPB][LDS];
  unsigned char *lds = lds_all[threadIdx.x / EBC_WAVE];
  RoleLds &L = *reinterpret_cast<RoleLds *>(lds);
  const int lane = threadIdx.x & (EBC_WAVE - 1);
  unsigned b = blockIdx.x * EBC_STEP_WPB + threadIdx.x / EBC_WAVE;
  if (EBC_STEP_WPB > 1 && b >= g.total) return;
#ifndef EBC_ROLE_MASK  // register-budget experiments: compile a subset of the roles
#define EBC_ROLE_MASK 15
#endif
#ifdef EBC_ORCA_FIRST
  // Experiment (measured slower: 18.4 against 17.4 us per step, profiles/r02_block_order_ab.txt).  ENV + ORCA waves
  // (683 + 5852 at 4096 x 10) are more than the chip keeps resident (6144 at six waves per SIMD): whichever
  // role comes second has waves that start only when slots free up (~3.5-6 us).  With ENV first the late
  // starters are ORCA waves and the launch ends with them; with ORCA first they are ENV waves — and their
  // chain (9 us) then ends later than the late ORCA waves did.
  if (b < orca_blocks) {
    const OrcaHot hot{hot_E, hot_N, hot_magic, hot_shift, hot_tile, hot_n_humans};
    if (EBC_ROLE_MASK & 2) orca_role<GS>(p_in, s_in, hot, hot_vel, hot_epoch, lds, b, lane);
    return;
  }
  b -= orca_blocks;
  if (b < env_blocks) {
    __builtin_amdgcn_s_setprio(EBC_ENV_PRIO);  // the long dependent chain of the launch
    if (EBC_ROLE_MASK & 1) env_role(p_in, s_in, io_in, L, (int)b, g.epoch, lane);
    return;
  }
  b -= env_blocks;
#else
  if (b < env_blocks) {
    __builtin_amdgcn_s_setprio(EBC_ENV_PRIO);  // the long dependent chain of the launch
    if (hot_pad) {  // lane = env (round 3)
      if (EBC_ROLE_MASK & 1) env_lane_role(p_in, s_in, io_in, L, b, g.epoch, lane);
    } else {
      if (EBC_ROLE_MASK & 1) env_role(p_in, s_in, io_in, L, (int)b, g.epoch, lane);
    }
    return;
  }
  b -= env_blocks;
  if (b < orca_blocks) {
    // ahead of the ROWS / STATE waves that poll on the same SIMD, behind the ENV chain (priority 3):
    // -0.08 us per step (profiles/r02_early_rows_ab.txt)
    __builtin_amdgcn_s_setprio(EBC_ORCA_PRIO);
    const OrcaHot hot{hot_E, hot_N, hot_magic, hot_shift, hot_tile, hot_n_humans};
    if (EBC_ROLE_MASK & 2) orca_role<GS>(p_in, s_in, hot, hot_vel, hot_epoch, lds, b, lane);
    return;
  }
  b -= orca_blocks;
#endif
  if (b < g.rows_blocks) {
    if (EBC_ROLE_MASK & 4) rows_role<T>(p_in, s_in, io_in, L, b, g.rows_epw, g.epoch, lane);
    return;
  }
  b -= g.rows_blocks;
  if (EBC_ROLE_MASK & 8) state_role(p_in, s_in, io_in, L, (int)b, g.rows_blocks != 0, g.epoch, lane);
}

// ---- ENV with EBC_ENV_LANES lanes per ENV (round 3).  The ENV role above maps a lane to a human slot: 6 envs per wave
// at N = 10, so the robot's policy (double-precision atan2 / sin / cos), the grid window and the reward run on 6 lanes
// of 64 — 683 waves that, with the 5 852 ORCA waves, exceed the 6 144 resident slots: 391 ORCA waves start 5 us late
// and the launch ends with them (profiles/r02_wave_timeline_marks.txt).  Here an env has 4 lanes (16 envs per wave,
// 256 waves at 4096 envs: ENV + ORCA waves all fit at once): each walks a contiguous quarter of the env's humans —
// the serial form of env.py:303-313 with its break per type — and the quarters are combined in order.  (A lane per
// env, 64 waves, was tried first: its 10-human serial walk made the role 14.8 us long and everything waited for
// env_done, profiles/r03_env_lane_timeline.txt.)  Same mailboxes as ENV: robot_n + robot_ready early, env_done last.
#define EBC_ENV_LANES 4
__device__ __forceinline__ void env_lane_role(const EbcParams &p_in, const DevState &s_in, const StepIO &io_in, RoleLds &L,
                                              unsigned block, unsigned epoch, int lane) {
  stage_args(&L.args.p, p_in, lane);
  stage_args(&L.args.s, s_in, lane);
  stage_args(&L.args.io, io_in, lane);
  wave_sync();
  const EbcParams &p = L.args.p;
  const DevState &s = L.args.s;
  const StepIO &io = L.args.io;
  constexpr int Q = EBC_ENV_LANES, EPW = EBC_WAVE / Q;
  const int el = lane / Q, part = lane - el * Q;
  const int e0 = (int)block * EPW;
  const int e = e0 + el;
  const bool ok = e < s.E;
  const size_t ee = ok ? (size_t)e : 0;
  const int N = s.N;
  double rb[9];
#pragma unroll
  for (int c = 0; c < 9; ++c) rb[c] = ok ? s.robot[ee * 9 + c] : 0.0;
  const int n = ok ? s.n_humans[ee] : 0;
  double gtime = ok ? s.time[ee] : 0.0;
  int grid_slot = (ok && s.pool.grid) ? s.grid_scene[ee] : 0;
  const int chunk = (N + Q - 1) / Q;
  const int i0 = part * chunk;
  pin(gtime); pin(grid_slot);
#pragma unroll
  for (int c = 0; c < 9; ++c) pin(rb[c]);
  double a0 = 0, a1 = 0;
  if (ok) robot_action(io, ee, rb, a0, a1);
  EBC_MARK(1);
  {  // the robot's next state, published early: the ROWS role builds the observation frame from it
    double rn[9];
#pragma unroll
    for (int c = 0; c < 9; ++c) rn[c] = rb[c];
    robot_advance(p, rn, a0, a1);
    double *stage = &L.env.cand[0][0];  // 3 x 64 doubles >= 16 robots
    if (part == 0) {
#pragma unroll
      for (int c = 0; c < 9; ++c) stage[el * 9 + c] = rn[c];
      // what only the env's first lane needs at the end: parked, out of the registers of the distance work
      L.env.gtime[el] = gtime;
      L.env.slot[el] = grid_slot;
      L.env.goal[el][0] = rb[5];
      L.env.goal[el][1] = rb[6];
      L.env.ract[el][0] = rn[0];  // the next position (Agent.compute_position): robot_advance's
      L.env.ract[el][1] = rn[1];
    }
    wave_sync();
    const int here = min(EPW, s.E - e0);
    for (int q = lane; q < 9 * here; q += EBC_WAVE)
      __hip_atomic_store(s.robot_n + (size_t)e0 * 9 + q, stage[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // this lane's quarter of the humans, asked for only now: they do not fit the 80-register budget beside the
  // double-precision atan2 / sin / cos of the robot's policy (60 spilled registers when loaded up front)
  constexpr int PRE = 2;  // humans per lane that are loaded together (N <= 8: all of them; a third comes as it is reached)
  double hpx[PRE], hpy[PRE], hvx[PRE], hvy[PRE], hr[PRE];
  int ht[PRE];
  {
    const DevState &sl = read_late(s);
#pragma unroll
    for (int q = 0; q < PRE; ++q) {
      const int i = i0 + q;
      const bool have = ok && q < chunk && i < N;
      const size_t k = ee * N + (have ? i : 0);
      hpx[q] = have ? sl.px[k] : 0.0; hpy[q] = have ? sl.py[k] : 0.0; hvx[q] = have ? sl.vx[k] : 0.0; hvy[q] = have ? sl.vy[k] : 0.0;
      hr[q] = have ? sl.radius[k] : 0.0;
      ht[q] = have ? (int)sl.type[k] : 0;
    }
  }
  double rvx, rvy;
  if (p.robot_kinematics == EBC_HOLONOMIC) {
    rvx = a0;
    rvy = a1;
  } else {
    rvx = a0 * cos(a1 + rb[8]);
    rvy = a0 * sin(a1 + rb[8]);
  }
  // compute_collisions (env.py:303-338) over this lane's humans: per type in index order, break at the first hit
  double dmin[3] = {INFINITY, INFINITY, INFINITY};
  int coll[4] = {0, 0, 0, 0};
  auto visit = [&](double px, double py, double vx, double vy, double r, int t) {
    const double d = closest_dist(px, py, vx, vy, r, rb[0], rb[1], rb[4], rvx, rvy, p.time_step);
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      if (t == q && !coll[q]) {
        if (d < 0) coll[q] = 1;
        else if (d < dmin[q]) dmin[q] = d;
      }
    }
  };
#pragma unroll
  for (int q = 0; q < PRE; ++q)
    if (q < chunk && i0 + q < n) visit(hpx[q], hpy[q], hvx[q], hvy[q], hr[q], ht[q]);
  for (int q = PRE; q < chunk; ++q) {  // N > 12: the rest of the quarter, loaded as it is reached
    const int i = i0 + q;
    if (!__any(i < n)) break;
    if (i < n) {
      const size_t k = ee * N + i;
      visit(s.px[k], s.py[k], s.vx[k], s.vy[k], s.radius[k], (int)s.type[k]);
    }
  }
  // the robot's stores went out a distance computation ago: this wait is short
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (ok && part == 0) mailbox_put(s.robot_ready + ee, epoch);
  // the quarters in order: a later quarter counts for a type only while no earlier one has hit it
  double fm[3] = {INFINITY, INFINITY, INFINITY};
  int fc[3] = {0, 0, 0};
  const int base = lane - part;
#pragma unroll
  for (int src = 0; src < Q; ++src) {
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const double od = __shfl(dmin[t], base + src, EBC_WAVE);
      const int oc = __shfl(coll[t], base + src, EBC_WAVE);
      if (!fc[t]) {
        fm[t] = od < fm[t] ? od : fm[t];
        fc[t] = oc;
      }
    }
  }
  EBC_MARK(2);
  if (ok && part == 0) {
    const EnvScratch &XL = read_late(L.env);
    const double nx = XL.ract[el][0], ny = XL.ract[el][1];
    // (the grid window worked out earlier, beside the humans' loads, measured 0.5 us SLOWER per step: r03_step_forms_ab.txt)
    int c4[4] = {fc[0], fc[1], fc[2], 0};
    c4[3] = grid_collision(s.pool.grid ? s.pool.grid + (size_t)XL.slot[el] * s.G * 2 : nullptr, s.G, p.map_size_m,
                           p.map_resolution, nx, ny, rb[4], io.has_border ? io.border : nullptr);
    EBC_MARK(3);
    const RewardOut ro = reward_compute(p, nx, ny, XL.goal[el][0], XL.goal[el][1], rb[4], a1, XL.gtime[el], fm, c4);
    s.done[ee] = (uint8_t)ro.done;
    if (io.reward) io.reward[ee] = ro.reward;
    if (io.done) io.done[ee] = (uint8_t)ro.done;
    if (io.info) io.info[ee] = (uint8_t)ro.info;
    if (io.dmin) {
      io.dmin[3 * ee] = fm[0];
      io.dmin[3 * ee + 1] = fm[1];
      io.dmin[3 * ee + 2] = fm[2];
    }
    if (io.dist_to_goal) io.dist_to_goal[ee] = ro.dist_to_goal;
    if (io.robot_action_out) {
      io.robot_action_out[2 * ee] = a0;
      io.robot_action_out[2 * ee + 1] = a1;
    }
    mailbox_put(s.env_done + ee, ((unsigned long long)epoch << 32) | (1u + (unsigned)ro.done));
  }
  EBC_MARK(4);
}

// =========================================================================== ORCA step, second form (round 3)
// What the wave timeline of the four-role launch showed (profiles/r02_wave_timeline_marks.txt): the launch ends with
// its CONSUMER roles.  A STATE wave starts at 8 us (last in dispatch order, when slots free up), needs 5.5 us for three
// dependent memory round trips under the load of 2 000 polling waves, and stores 2.5 us after its mailboxes are full;
// the last ORCA wave ends 2.8 us before the launch does; ENV + ORCA waves (6 535) exceed the 6 144 resident slots, so
// 391 ORCA waves start 5 us late.  Here the hand-offs that sat on the tail are gone:
//
//   ENV1  lane = ENV (64 envs per wave, 64 waves at 4096 envs): the robot's action and next state (the double-precision
//         atan2 / sin / cos of the linear policy at full lane occupancy instead of 6 lanes of 64), robot_n, and two
//         epoch-tagged records per env: `ract` (action + next position, for ENV2) and `frame` (the robot-centric frame
//         of rotate(), for every row writer).  Dispatched first; waits for nothing.
//   ORCA  the lane group of a human, as before — and then lane 0 of the group COMMITS its human: next position, first
//         arrival, the next float tile record (into the second buffers: other groups still read the current ones),
//         the human's velocity and its raw / rotated observation row.  No velocity mailbox, no consumer wave, no hop
//         between the LP and the state: the launch ends when its last ORCA wave does.  ENV1 + ORCA = 5 916 waves: all
//         resident at once, none starts late.
//   ENV2  lane = human slot (6 envs per wave): swept distances, ordered per-type reduce, grid window, reward / done /
//         info, the rows of static obstacles and padding, time — nothing here waits for ORCA.  A terminal env under
//         auto-reset (rare) waits for its humans' `committed` flags and writes the restart scene over their commits.
//         Dispatched last: its waves start as ORCA waves finish, and are done before the slow ORCA waves are.
//
// Ordering of conflicting stores across XCDs (their L2s are not coherent): a commit that a restart may overwrite
// leaves with write-through (sc1) stores and is drained (s_waitcnt vmcnt(0)) before the `committed` flag; the
// restart's plain stores are written back at the end of the launch, after it.
struct Step2Grid {
  unsigned env1_blocks, orca_blocks, env2_blocks;
  unsigned epoch;
  unsigned total;
};

template <int GS>
struct Step2Lds {
  ArgBlock args;
  union {
    EnvScratch env;              // ENV2
    double robots[EBC_WAVE][9];  // ENV1: the robots' next states, staged for one coalesced run of stores
    struct {                     // ORCA: the LP's scratch, and what lane 0 of each group needs after the LP
      unsigned char lp[(OrcaLds<GS>::BYTES + 15) / 16 * 16];
      double v[(EBC_WAVE + GS - 1) / GS][7];  // px, py, gx, gy, radius, arrival, global time
      int type[(EBC_WAVE + GS - 1) / GS];
    } orca;
  };
};

__device__ __forceinline__ void store16_sc1(void *dst, u32x4 v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 2" ::"v"(dst), "v"(v) : "memory");  // s_nop: see store16_device
}
__device__ __forceinline__ void store8_sc1(double *dst, double v) {
  asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(dst), "v"(v) : "memory");
}
__device__ __forceinline__ u32x4 tag3(float a, float b, float c, unsigned epoch) {
  return u32x4{__float_as_uint(a), __float_as_uint(b), __float_as_uint(c), epoch};
}
__device__ __forceinline__ u32x4 tag_double(double v, unsigned epoch) {
  const unsigned long long u = (unsigned long long)__double_as_longlong(v);
  return u32x4{(unsigned)u, (unsigned)(u >> 32), epoch, epoch};
}
__device__ __forceinline__ double untag_double(u32x4 v) {
  return __longlong_as_double((long long)(((unsigned long long)v.y << 32) | v.x));
}

// The frame record of env `ee` (4 granules written by ENV1): polled until every granule carries this launch's epoch.
__device__ __forceinline__ RotFrame frame_wait(const uint4 *frame, size_t ee, bool need, unsigned epoch, unsigned *fault) {
  u32x4 g0 = {0u, 0u, 0u, 0u}, g1 = g0, g2 = g0, g3 = g0;
  bool waiting = need;
  for (unsigned spins = 0;; ++spins) {
    if (waiting) {
      const uint4 *b = frame + ee * 4;
      asm volatile("global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %4, off offset:16 sc1\n\t"
                   "global_load_dwordx4 %2, %4, off offset:32 sc1\n\tglobal_load_dwordx4 %3, %4, off offset:48 sc1\n\t"
                   "s_waitcnt vmcnt(0)"
                   : "=&v"(g0), "=&v"(g1), "=&v"(g2), "=&v"(g3) : "v"(b) : "memory");
      waiting = g0.w != epoch || g1.w != epoch || g2.w != epoch || g3.w != epoch;
    }
    if (!__any(waiting)) break;
    if (spins > EBC_SPIN_LIMIT) {
      if (waiting) atomicOr(fault, 1u);
      break;
    }
    __builtin_amdgcn_s_sleep(EBC_POLL_SLEEP);
  }
  RotFrame f;
  f.px = __uint_as_float(g0.x); f.py = __uint_as_float(g0.y); f.r = __uint_as_float(g0.z);
  f.vpref = __uint_as_float(g1.x); f.c = __uint_as_float(g1.y); f.s = __uint_as_float(g1.z);
  f.dg = __uint_as_float(g2.x); f.vx = __uint_as_float(g2.y); f.vy = __uint_as_float(g2.z);
  f.theta = __uint_as_float(g3.x);
  return f;
}

template <int T>
__device__ __forceinline__ void emit_row(const StepIO &io, size_t row_index, const RotFrame &f, bool valid, double opx, double opy,
                                         double ovx, double ovy, double orad, int otype) {
  if (io.ob) {
    double *o = io.ob + row_index * 5;
    o[0] = valid ? opx : 0.0; o[1] = valid ? opy : 0.0; o[2] = valid ? ovx : 0.0; o[3] = valid ? ovy : 0.0; o[4] = valid ? orad : 0.0;
  }
  if (io.obs_rotated) {
    float out[T];
    if (valid) {
      rotate_row<T>(f, opx, opy, ovx, ovy, orad, otype, out);
    } else {
#pragma unroll
      for (int c = 0; c < T; ++c) out[c] = 0.0f;
    }
    float *o = io.obs_rotated + row_index * T;  // nobody on the device reads the rows: streaming stores
#pragma unroll
    for (int c = 0; c < T; ++c) __builtin_nontemporal_store(out[c], o + c);
  }
}

// ---- ENV1: lane = env
template <typename Lds>
__device__ __forceinline__ void env1_role(const EbcParams &p_in, const DevState &s_in, const StepIO &io_in, Lds &L,
                                          unsigned block, unsigned epoch, int lane) {
  stage_args(&L.args.p, p_in, lane);
  stage_args(&L.args.s, s_in, lane);
  stage_args(&L.args.io, io_in, lane);
  wave_sync();
  const EbcParams &p = L.args.p;
  const DevState &s = L.args.s;
  const StepIO &io = L.args.io;
  const int e0 = (int)block * EBC_WAVE;
  const int e = e0 + lane;
  const bool ok = e < s.E;
  const size_t ee = ok ? (size_t)e : 0;
  double rb[9];
#pragma unroll
  for (int c = 0; c < 9; ++c) rb[c] = ok ? s.robot[ee * 9 + c] : 0.0;
  double a0 = 0, a1 = 0;
  if (ok) robot_action(io, ee, rb, a0, a1);
  robot_advance(p, rb, a0, a1);  // rb = the robot's next state (agent.py:202-228)
  const RotFrame f = rot_frame(rb, p.rotate_unicycle);
  bool publish = ok;
#ifdef EBC_WAVE_TRACE  // test build only: the env of the withheld human never gets its records (tests the give-up path)
  if (g_withhold_human >= 0 && e == g_withhold_human / s.N) publish = false;
#endif
  if (publish) {
    uint4 *fr = s.frame + ee * 4, *ra = s.ract + ee * 4;
    store16_sc1(fr, tag3(f.px, f.py, f.r, epoch));
    store16_sc1(fr + 1, tag3(f.vpref, f.c, f.s, epoch));
    store16_sc1(fr + 2, tag3(f.dg, f.vx, f.vy, epoch));
    store16_sc1(fr + 3, tag3(f.theta, 0.0f, 0.0f, epoch));
    store16_sc1(ra, tag_double(a0, epoch));
    store16_sc1(ra + 1, tag_double(a1, epoch));
    store16_sc1(ra + 2, tag_double(rb[0], epoch));
    store16_sc1(ra + 3, tag_double(rb[1], epoch));
    if (io.robot_action_out) {
      io.robot_action_out[2 * ee] = a0;
      io.robot_action_out[2 * ee + 1] = a1;
    }
  }
  // robot_n: what the next step reads (a restart overwrites it later in this launch: write-through, so that the
  // restart's write-back at the end of the launch comes after it).  One coalesced run for the wave's 64 envs.
#pragma unroll
  for (int c = 0; c < 9; ++c) L.robots[lane][c] = rb[c];
  wave_sync();
  const int envs_here = min(EBC_WAVE, s.E - e0);
  for (int q = lane; q < 9 * envs_here; q += EBC_WAVE) store8_sc1(s.robot_n + (size_t)e0 * 9 + q, (&L.robots[0][0])[q]);
}

// ---- ORCA: the LP of v1, then lane 0 of each group commits its human and writes its observation row
template <int GS, int T>
__device__ __forceinline__ void orca_role2(const EbcParams &p_in, const DevState &s_in, const StepIO &io_in, const OrcaHot &hot,
                                           unsigned epoch, Step2Lds<GS> &L, unsigned block, int lane) {
  constexpr int HPW = EBC_WAVE / GS;
  const int group = lane / GS, j = lane - group * GS;
  const unsigned hh = block * HPW + group;
  const bool h_ok = group < HPW && hh < (unsigned)hot.E * (unsigned)hot.N;  // lanes past HPW * GS idle
  int e, i;
  split_human(hot, hh, h_ok, e, i);
  // the tile loads go out first (their addresses come from preloaded arguments); the argument block follows them
  // into LDS while they are in flight
  const OrcaTile tile = orca_tile_load<GS>(hot, h_ok, e, i);
  stage_args(&L.args.p, p_in, lane);
  stage_args(&L.args.s, s_in, lane);
  stage_args(&L.args.io, io_in, lane);
  wave_sync();
  const bool mine = h_ok && j == 0;
  // what the commit needs of the human, asked for now (lane 0 of the group) and parked in LDS after the ranking:
  // fifteen registers that must not live through the LP
  double cpx = 0, cpy = 0, cgx = 0, cgy = 0, crad = 0, carr = 0, ctime = 0;
  int ctype = 0;
  if (mine) {
    const DevState &s = L.args.s;
    const size_t k = hh;
    cpx = s.px[k]; cpy = s.py[k]; cgx = s.gx[k]; cgy = s.gy[k]; crad = s.radius[k]; carr = s.arrival[k];
    ctype = s.type[k];
    ctime = s.time[e];  // read NOW: ENV2 advances it while this launch runs
  }
  auto park = [&]() {
    if (mine) {
      double *v = L.orca.v[group];
      v[0] = cpx; v[1] = cpy; v[2] = cgx; v[3] = cgy; v[4] = crad; v[5] = carr; v[6] = ctime;
      L.orca.type[group] = ctype;
    }
  };
  float ox, oy;
  bool human_ok;
  orca_wave_compute<GS>(L.args.p, L.args.s, hot, tile, h_ok, e, i, L.orca.lp, ox, oy, human_ok, park);
  if (!mine) return;
  const EbcParams &p = read_late(L.args.p);
  const DevState &s = read_late(L.args.s);
  const StepIO &io = read_late(L.args.io);
  const size_t k = hh, ee = (size_t)e;
  const int R = s.N + s.S;
  const double *v = read_late(L.orca.v[group]);
  double px = v[0], py = v[1];
  const double gx = v[2], gy = v[3], rad = v[4], arrival = v[5], gtime = v[6];
  const int type = read_late(L.orca.type[group]);
  double ax = 0, ay = 0;
  typedef float v4f __attribute__((ext_vector_type(4)));
  v4f t0 = {0, 0, 0, 0}, t1 = {0, 0, 0, 0};
  if (human_ok) {  // Agent.step (agent.py:202-211), first arrival (env.py:365-378), the next tile record (orca.py:110-140)
    ax = (double)ox;  // getAgentVelocity -> Python float
    ay = (double)oy;
    px = px + ax * p.time_step;
    py = py + ay * p.time_step;
    const double dx = gx - px, dy = gy - py;
    const double dist = norm2(dx, dy);
    if (arrival == 0 && dist < rad) store8_sc1(s.arrival + k, gtime + p.time_step);
    float prefx, prefy;
    orca_pref_from(dx, dy, dist, prefx, prefy);
    t0 = v4f{(float)px, (float)py, ox, oy};
    t1 = v4f{tile.b.x, tile.b.y, prefx, prefy};  // radius + 0.01 + safety and maxSpeed do not change within an episode
  } else {
    px = py = 0.0;
  }
  // write-through: a restart of this env (ENV2, on another XCD) may overwrite these before the launch ends
  store8_sc1(s.px_n + k, px);
  store8_sc1(s.py_n + k, py);
  store8_sc1(s.vx_n + k, ax);
  store8_sc1(s.vy_n + k, ay);
  {
    u32x4 w0, w1;
    __builtin_memcpy(&w0, &t0, 16);
    __builtin_memcpy(&w1, &t1, 16);
    store16_sc1(s.tile_n + 2 * k, w0);
    store16_sc1(s.tile_n + 2 * k + 1, w1);
  }
  if (io.human_action) {
    io.human_action[k * 2] = ax;
    io.human_action[k * 2 + 1] = ay;
  }
  EBC_MARK(5);
  if (io.ob || io.obs_rotated) {
    // the human's row: row i, or — a padded slot — one of the zero rows behind the static ones (rows_role's map)
    const RotFrame f = frame_wait(s.frame, ee, true, epoch, s.fault);
    const int ns = (human_ok || !s.S) ? 0 : s.n_static[ee];
    const int row = human_ok ? i : ns + i;
    emit_row<T>(io, ee * R + row, f, human_ok, px, py, ax, ay, rad, type);
  }
  if (io.auto_reset) {  // the flag a restart waits for: behind the commit's stores, and behind this lane's last read of
                        // anything a restart rewrites (n_static)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    mailbox_put(s.committed + k, epoch);
  }
}

// ---- ENV2: lane = human slot
template <int T, typename Lds>
__device__ __forceinline__ void env2_role(const EbcParams &p_in, const DevState &s_in, const StepIO &io_in, Lds &L,
                                          unsigned block, unsigned epoch, int lane) {
  stage_args(&L.args.p, p_in, lane);
  stage_args(&L.args.s, s_in, lane);
  stage_args(&L.args.io, io_in, lane);
  wave_sync();
  const EbcParams &p = L.args.p;
  const DevState &s = L.args.s;
  const StepIO &io = L.args.io;
  EnvScratch &X = L.env;
  const int N = s.N, S = s.S, R = N + S;
  const int epb = EBC_WAVE / N;
  const LaneMap m = lane_map(s, (int)block * epb, epb, lane);
  // pre-step state: nothing of it is overwritten while this launch runs (commits go to the second buffers)
  double rb[9];
  load_robot(s, m, rb);
  double gtime = m.env_ok ? s.time[m.ee] : 0.0;
  HumanRegs h = load_human(s, m);
  int grid_slot = (m.leader && s.pool.grid) ? s.grid_scene[m.ee] : 0;
  int cursor = 0;
  if (io.auto_reset && m.leader) cursor = s.pool.P > 0 ? s.pool.cursor[m.ee] : (int)m.ee;
  pin(gtime); pin(grid_slot); pin(cursor);
  pin(h.px); pin(h.py); pin(h.vx); pin(h.vy); pin(h.rad); pin(h.type);
  if (m.leader) {  // what only the leader's tail needs: parked in LDS, out of the registers of the distance work
    X.gtime[m.el] = gtime;
    X.n[m.el] = m.n;
    X.slot[m.el] = grid_slot;
    X.cursor[m.el] = cursor;
    X.goal[m.el][0] = rb[5];
    X.goal[m.el][1] = rb[6];
  }
  // the robot's action and next position (ENV1)
  double a0, a1, nx, ny;
  {
    u32x4 g0 = {0u, 0u, 0u, 0u}, g1 = g0, g2 = g0, g3 = g0;
    bool waiting = m.env_ok;
    for (unsigned spins = 0;; ++spins) {
      if (waiting) {
        const uint4 *b = s.ract + m.ee * 4;
        asm volatile("global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %4, off offset:16 sc1\n\t"
                     "global_load_dwordx4 %2, %4, off offset:32 sc1\n\tglobal_load_dwordx4 %3, %4, off offset:48 sc1\n\t"
                     "s_waitcnt vmcnt(0)"
                     : "=&v"(g0), "=&v"(g1), "=&v"(g2), "=&v"(g3) : "v"(b) : "memory");
        waiting = g0.z != epoch || g0.w != epoch || g1.z != epoch || g1.w != epoch || g2.z != epoch || g2.w != epoch ||
                  g3.z != epoch || g3.w != epoch;
      }
      if (!__any(waiting)) break;
      if (spins > EBC_SPIN_LIMIT) {
        if (waiting) atomicOr(s.fault, 1u);
        break;
      }
      __builtin_amdgcn_s_sleep(EBC_POLL_SLEEP);
    }
    a0 = untag_double(g0); a1 = untag_double(g1); nx = untag_double(g2); ny = untag_double(g3);
  }
  if (m.leader) {
    X.ract[m.el][0] = nx;
    X.ract[m.el][1] = ny;
  }
  EBC_MARK(1);
  double rvx, rvy;
  if (p.robot_kinematics == EBC_HOLONOMIC) {
    rvx = a0;
    rvy = a1;
  } else {
    rvx = a0 * cos(a1 + rb[8]);
    rvy = a0 * sin(a1 + rb[8]);
  }
  // ordered per-type reduction with break at the first hit (env.py:303-313): service_env's form
  const double d = m.active ? closest_dist(h.px, h.py, h.vx, h.vy, h.rad, rb[0], rb[1], rb[4], rvx, rvy, p.time_step) : 0.0;
  const bool hit = m.active && d < 0;
  const int base = lane - m.i;
  int coll[4] = {0, 0, 0, 0};
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    const unsigned long long hm = (__ballot(hit && h.type == t) >> base) & (N >= 64 ? ~0ull : ((1ull << N) - 1));
    const int first = hm ? __ffsll((long long)hm) - 1 : N;
    coll[t] = hm != 0;
    X.cand[t][lane] = (m.active && h.type == t && !hit && m.i < first) ? d : INFINITY;
  }
  wave_sync();
  EBC_MARK(2);
  int done_flag = 0;
  if (m.leader) {
    const EnvScratch &XL = read_late(X);
    const double nx = XL.ract[m.el][0], ny = XL.ract[m.el][1], gtime = XL.gtime[m.el];
    const int n_env = XL.n[m.el];
    double dm0 = INFINITY, dm1 = INFINITY, dm2 = INFINITY;
    for (int q = 0; q < n_env; ++q) {
      const double d0 = X.cand[0][lane + q], d1 = X.cand[1][lane + q], d2 = X.cand[2][lane + q];
      dm0 = d0 < dm0 ? d0 : dm0;
      dm1 = d1 < dm1 ? d1 : dm1;
      dm2 = d2 < dm2 ? d2 : dm2;
    }
    const double dmin[3] = {dm0, dm1, dm2};
    coll[3] = grid_collision(s.pool.grid ? s.pool.grid + (size_t)XL.slot[m.el] * s.G * 2 : nullptr, s.G, p.map_size_m,
                             p.map_resolution, nx, ny, rb[4], io.has_border ? io.border : nullptr);
    const RewardOut ro = reward_compute(p, nx, ny, XL.goal[m.el][0], XL.goal[m.el][1], rb[4], a1, gtime, dmin, coll);
    done_flag = ro.done;
    if (!(io.auto_reset && ro.done)) s.time[m.ee] = gtime + p.time_step;
    s.done[m.ee] = (uint8_t)ro.done;
    if (io.reward) io.reward[m.ee] = ro.reward;
    if (io.done) io.done[m.ee] = (uint8_t)ro.done;
    if (io.info) io.info[m.ee] = (uint8_t)ro.info;
    if (io.dmin) {
      io.dmin[3 * m.ee] = dm0;
      io.dmin[3 * m.ee + 1] = dm1;
      io.dmin[3 * m.ee + 2] = dm2;
    }
    if (io.dist_to_goal) io.dist_to_goal[m.ee] = ro.dist_to_goal;
  }
  EBC_MARK(3);
  done_flag = __shfl(done_flag, base, EBC_WAVE);
  const bool restore = m.env_ok && io.auto_reset && done_flag;
  // the rows of static obstacles and the padding rows (rows_role's map: static j < ns -> n + j, else N + j)
  if ((io.ob || io.obs_rotated) && S > 0) {
    const RotFrame f = frame_wait(s.frame, m.ee, m.env_ok, epoch, s.fault);
    for (int q = m.i; q < S; q += N) {
      if (!m.env_ok) break;
      const size_t c = m.ee * S + q;  // loaded here: ENV2 is not what the launch waits for
      const double sx = s.spx[c], sy = s.spy[c], sr = s.sradius[c];
      const bool valid = q < m.ns;
      emit_row<T>(io, m.ee * R + (valid ? m.n + q : N + q), f, valid, sx, sy, 0.0, 0.0, sr, EBC_ADULT_STATIC);
    }
  }
  EBC_MARK(4);
  if (!__any(restore)) return;
  // ---- a terminal env under auto-reset takes its next scene (rare): behind its humans' commits
  mailbox_wait_epoch(s.committed + m.k, restore, epoch, s.fault);
  if (!restore) return;
  const ScenePool &P = s.pool;
  cursor = read_late(X).cursor[m.el];
  const size_t src = (size_t)cursor * N + m.i;
  const int n_new = P.n_humans[cursor];
  const bool live = m.i < n_new;
  const double npx = live ? P.px[src] : 0.0, npy = live ? P.py[src] : 0.0, nvx = live ? P.vx[src] : 0.0, nvy = live ? P.vy[src] : 0.0;
  const double nrad = live ? P.radius[src] : 0.0, nvp = live ? P.v_pref[src] : 0.0;
  s.px_n[m.k] = npx; s.py_n[m.k] = npy; s.vx_n[m.k] = nvx; s.vy_n[m.k] = nvy;
  s.arrival[m.k] = 0.0;
  s.gx[m.k] = live ? P.gx[src] : 0.0;
  s.gy[m.k] = live ? P.gy[src] : 0.0;
  s.radius[m.k] = nrad;
  s.v_pref[m.k] = nvp;
  s.type[m.k] = live ? P.type[src] : (uint8_t)0;
  if (live) {
    const float2 rp = P.pref[src];
    s.tile_n[2 * m.k] = make_float4((float)npx, (float)npy, (float)nvx, (float)nvy);
    s.tile_n[2 * m.k + 1] = make_float4((float)(nrad + 0.01 + p.orca_safety_space), (float)nvp, rp.x, rp.y);
  } else {
    s.tile_n[2 * m.k] = make_float4(0, 0, 0, 0);
    s.tile_n[2 * m.k + 1] = make_float4(0, 0, 0, 0);
  }
  for (int q = m.i; q < S; q += N) {
    const size_t c = (size_t)cursor * S + q;
    s.spx[m.ee * S + q] = P.spx[c];
    s.spy[m.ee * S + q] = P.spy[c];
    s.sradius[m.ee * S + q] = P.sradius[c];
  }
  if (m.leader) {
    s.n_humans[m.ee] = n_new;
    if (S) s.n_static[m.ee] = P.n_static[cursor];
    s.grid_scene[m.ee] = cursor;
    for (int q = 0; q < 9; ++q) s.robot_n[m.ee * 9 + q] = P.robot[(size_t)cursor * 9 + q];
    s.time[m.ee] = 0.0;
    if (P.P > 0) {  // walk the custom pool; without one the env keeps restarting from its own slot
      int nxt = cursor - s.E + P.stride;
      P.cursor[m.ee] = s.E + (nxt >= P.P ? nxt % P.P : nxt);
    }
  }
}

template <int GS, int T>
__global__ __launch_bounds__(EBC_WAVE, EBC_STEP_WAVES(GS)) void orca_step2_kernel(
    unsigned env1_blocks, unsigned orca_blocks, int hot_E, int hot_N, unsigned hot_magic, unsigned hot_shift,
    const float4 *hot_tile, const int *hot_n_humans, unsigned hot_epoch, unsigned hot_pad0, unsigned hot_pad1, unsigned hot_pad2,
    EbcParams p_in, DevState s_in, StepIO io_in, Step2Grid g) {
  const WaveTrace wt(2);
  __shared__ __align__(16) Step2Lds<GS> L;
  const int lane = threadIdx.x;
  unsigned b = blockIdx.x;
#ifndef EBC_STEP2_ROLES  // register-budget experiments: compile a subset of the roles
#define EBC_STEP2_ROLES 7
#endif
  if (b < env1_blocks) {
    __builtin_amdgcn_s_setprio(EBC_ENV_PRIO);  // everything else in the launch waits for these few waves
    if (EBC_STEP2_ROLES & 1) env1_role(p_in, s_in, io_in, L, b, hot_epoch, lane);
    return;
  }
  b -= env1_blocks;
  if (b < orca_blocks) {
    const OrcaHot hot{hot_E, hot_N, hot_magic, hot_shift, hot_tile, hot_n_humans};
    if (EBC_STEP2_ROLES & 2) orca_role2<GS, T>(p_in, s_in, io_in, hot, hot_epoch, L, b, lane);
    return;
  }
  b -= orca_blocks;
  if (EBC_STEP2_ROLES & 4) env2_role<T>(p_in, s_in, io_in, L, b, g.epoch, lane);
}

// SceneGenerator.generate_random_scene for n seeds, one lane per scene (ebc_scene_gen.h): `d` holds the base of
// arrays shaped like an EbcScene's, `mt` n MT19937 states word-major (lanes of a wave touch consecutive words).
__global__ __launch_bounds__(64) void scene_gen_kernel(EbcSceneGen c, const uint32_t *seeds, uint32_t seed0, int n, int N, int S,
                                                        int G, SceneRow d, uint32_t *mt, int *status) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const size_t k = (size_t)r * N, q = (size_t)r * (S ? S : 1);
  const SceneRow o = {d.n_humans + r, d.px + k, d.py + k, d.vx + k, d.vy + k, d.gx + k, d.gy + k, d.radius + k, d.v_pref + k,
                      d.type + k, d.n_static + r, d.spx + q, d.spy + q, d.sradius + q,
                      d.grid ? d.grid + (size_t)r * G * 2 : nullptr, d.robot + (size_t)r * 9};
  const int st = generate_scene_row(c, seeds ? seeds[r] : seed0 + (uint32_t)r, mt + r, (size_t)n, o, N, S, G);
  if (st) atomicOr(status, st);
}

// Discounted returns of a rollout window, one thread per env walking its K steps backwards (explorer.py:159-170,
// :82-92): see ebc_il_targets in include/ebcsim.h.
__global__ __launch_bounds__(256) void il_targets_kernel(const double *reward, const uint8_t *done, const uint8_t *info, int K,
                                                         int E, double gamma_bar, double *values, uint8_t *keep) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  double run = 0.0;
  uint8_t closed = 0;
  for (int t = K - 1; t >= 0; --t) {
    const size_t q = (size_t)t * E + e;
    const double r = reward[q];
    if (done[q]) {
      const uint8_t c = info[q];
      run = r;
      closed = (c == EBC_INFO_REACH_GOAL || c == EBC_INFO_COLLISION_OBSTACLE || c == EBC_INFO_COLLISION_ADULT ||
                c == EBC_INFO_COLLISION_BICYCLE || c == EBC_INFO_COLLISION_CHILD) ? 1 : 0;
    } else {
      run = r + gamma_bar * run;
    }
    values[q] = run;
    keep[q] = closed;
  }
}

// Observation rows that exist per env (humans + static obstacles as pedestrians, env.py:381-382, :457-458):
// what the value network's per-pair kernels mask with.  Read from the device state, so it follows restarts
// from a ragged scene pool.
__global__ __launch_bounds__(256) void row_counts_kernel(DevState s, long long *out) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e < s.E) out[e] = (long long)s.n_humans[e] + (s.S ? s.n_static[e] : 0);
}

// ------------------------------------------------------------------------- observe
// Rows of the current state (env.py:188-193), raw and rotated; thread = (env, row).
template <int T>
__global__ __launch_bounds__(256) void observe_kernel(EbcParams p, DevState s, double *ob, float *obs) {
  const int R = s.N + s.S;
  const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= (size_t)s.E * R) return;
  const size_t e = g / R;
  const int r = (int)(g - e * R);
  const int n = s.n_humans[e];
  const int ns = s.S ? s.n_static[e] : 0;
  double opx = 0, opy = 0, ovx = 0, ovy = 0, orad = 0;
  int otype = 0;
  bool valid = false;
  if (r < n) {
    const size_t k = e * s.N + r;
    opx = s.px[k]; opy = s.py[k]; ovx = s.vx[k]; ovy = s.vy[k]; orad = s.radius[k]; otype = s.type[k];
    valid = true;
  } else if (r - n < ns) {
    const size_t q = e * s.S + (r - n);
    opx = s.spx[q]; opy = s.spy[q]; orad = s.sradius[q]; otype = EBC_ADULT_STATIC;
    valid = true;
  }
  if (ob) {
    double *o = ob + g * 5;
    o[0] = opx; o[1] = opy; o[2] = ovx; o[3] = ovy; o[4] = orad;
  }
  if (obs) {
    float out[T];
    if (valid) {
      const RotFrame f = rot_frame(s.robot + e * 9, p.rotate_unicycle);
      rotate_row<T>(f, opx, opy, ovx, ovy, orad, otype, out);
    } else {
#pragma unroll
      for (int c = 0; c < T; ++c) out[c] = 0.0f;
    }
    float *o = obs + g * T;
#pragma unroll
    for (int c = 0; c < T; ++c) o[c] = out[c];
  }
}

// ------------------------------------------------------------------------- look-ahead
// The |A|-way onestep_lookahead sweep of MultiHumanRL.predict (multi_human_rl.py:38-61), one
// 256-thread workgroup per env.  Human velocities come from s.hact (ORCA kernel,
// ebc_set_human_actions) or the linear policy, once for all actions.
//   phase A  threads over rows: next observable rows (agent.py:80-93; env.py:457-458) -> LDS
//   phase B1 threads over (action, human) pairs: swept distance (collisions.py:29-57) -> LDS
//   phase B2 threads over actions: ordered per-type reduction, grid window, reward, and the
//            robot-frame terms of rotate() for the propagated robot (cadrl.py:118-165)
//   phase C  rows_rotated[e][a][r][:]: 256 rows at a time are rotated into an LDS tile and
//            leave as 16-byte stores in flat order (the 4 T bytes of a row are not a
//            multiple of 16: writing row by row costs T dword stores per lane and splits every
//            128-B line; this is the one output of the path that is HBM-bound: 4 T A R bytes
//            per env)
// Dynamic LDS: A * n_max doubles (pair distances), then 256 * T floats (row tile).
#define EBC_LA_THREADS 256
#define EBC_LA_MAX_ROWS 128
#define EBC_LA_MAX_ACTIONS 128
template <int POLICY, int T>
__global__ __launch_bounds__(EBC_LA_THREADS) void lookahead_kernel(EbcParams p, DevState s, LookIO io) {
  extern __shared__ __align__(16) unsigned char la_lds[];
  __shared__ double hpx[EBC_WAVE], hpy[EBC_WAVE], hvx[EBC_WAVE], hvy[EBC_WAVE], hrad[EBC_WAVE];
  __shared__ uint8_t htype[EBC_WAVE];
  __shared__ double row[EBC_LA_MAX_ROWS][5];
  __shared__ uint8_t row_type[EBC_LA_MAX_ROWS];
  __shared__ RotFrame frames[EBC_LA_MAX_ACTIONS];
  const int N = s.N, S = s.S, R = N + S, A = io.A;
  const int tid = threadIdx.x;
  const int e = blockIdx.x;
  const int n = s.n_humans[e];
  const int ns = S ? s.n_static[e] : 0;
  const double dt = p.time_step;
  const double *rbp = s.robot + (size_t)e * 9;
  double rb[9];
#pragma unroll
  for (int c = 0; c < 9; ++c) rb[c] = rbp[c];
  double *dist = reinterpret_cast<double *>(la_lds);                 // [A][N]
  float *tile = reinterpret_cast<float *>(la_lds + (size_t)A * N * 8);  // [256][T]

  for (int r = tid; r < R; r += EBC_LA_THREADS) {
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

  // phase B1: every (action, human) pair in parallel
  for (int q = tid; q < A * n; q += EBC_LA_THREADS) {
    const int a = q / n, j = q - a * n;
    const double a0 = io.actions[2 * a], a1 = io.actions[2 * a + 1];
    double rvx, rvy;
    if (p.robot_kinematics == EBC_HOLONOMIC) {
      rvx = a0;
      rvy = a1;
    } else {
      rvx = a0 * cos(a1 + rb[8]);
      rvy = a0 * sin(a1 + rb[8]);
    }
    dist[a * N + j] = closest_dist(hpx[j], hpy[j], hvx[j], hvy[j], hrad[j], rb[0], rb[1], rb[4], rvx, rvy, dt);
  }
  __syncthreads();

  // phase B2
  const double gtime = s.time[e];
  const int grid_slot = s.pool.grid ? s.grid_scene[e] : 0;
  for (int a = tid; a < A; a += EBC_LA_THREADS) {
    const double a0 = io.actions[2 * a], a1 = io.actions[2 * a + 1];
    double dm0 = INFINITY, dm1 = INFINITY, dm2 = INFINITY;
    int c0 = 0, c1 = 0, c2 = 0;
    for (int j = 0; j < n; ++j) {  // index order, break at the first hit per type (env.py:303-313)
      const int t = htype[j];
      const double d = dist[a * N + j];
      const bool hit = d < 0;
      if (t == 0 && !c0) { c0 = hit; dm0 = (!hit && d < dm0) ? d : dm0; }
      if (t == 1 && !c1) { c1 = hit; dm1 = (!hit && d < dm1) ? d : dm1; }
      if (t == 2 && !c2) { c2 = hit; dm2 = (!hit && d < dm2) ? d : dm2; }
    }
    const double dmin[3] = {dm0, dm1, dm2};
    double nx, ny;
    robot_next_position(rb, p.robot_kinematics, a0, a1, dt, nx, ny);
    int coll[4] = {c0, c1, c2, 0};
    coll[3] = grid_collision(s.pool.grid ? s.pool.grid + (size_t)grid_slot * s.G * 2 : nullptr, s.G, p.map_size_m,
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

  // phase C
  const int total_rows = A * R;
  const size_t env_base = (size_t)e * total_rows * T;  // in floats
  for (int c0 = 0; c0 < total_rows; c0 += EBC_LA_THREADS) {
    const int idx = c0 + tid;
    if (idx < total_rows) {
      const int a = idx / R, r = idx - a * R;
      float out[T];
      if (r < n + ns) {
        rotate_row<T>(frames[a], row[r][0], row[r][1], row[r][2], row[r][3], row[r][4], row_type[r], out);
      } else {
#pragma unroll
        for (int c = 0; c < T; ++c) out[c] = 0.0f;
      }
#pragma unroll
      for (int c = 0; c < T; ++c) tile[tid * T + c] = out[c];  // stride T is odd: conflict-free
    }
    __syncthreads();
    // flat copy of the chunk: floats [g0, g0 + cnt) of the output
    const int rows_here = min(EBC_LA_THREADS, total_rows - c0);
    const int cnt = rows_here * T;
    const size_t g0 = env_base + (size_t)c0 * T;
    const int head = (int)((4 - (g0 & 3)) & 3);  // floats before the first 16-byte boundary
    float *dst = io.rows + g0;
    if (tid < head && tid < cnt) dst[tid] = tile[tid];
    const int body4 = cnt > head ? (cnt - head) / 4 : 0;
    for (int v = tid; v < body4; v += EBC_LA_THREADS) {
      const int f = head + 4 * v;
      typedef float f32x4 __attribute__((ext_vector_type(4)));
      const f32x4 w = {tile[f], tile[f + 1], tile[f + 2], tile[f + 3]};
      __builtin_nontemporal_store(w, reinterpret_cast<f32x4 *>(dst + f));  // 400 MB that nothing on the device reads twice
    }
    const int tail0 = head + 4 * body4;
    if (tail0 + tid < cnt && tid < 4) dst[tail0 + tid] = tile[tail0 + tid];
    __syncthreads();
  }
}

}  // namespace ebc
