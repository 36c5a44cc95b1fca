/*
 * ebcsim.h — C ABI of libebcsim.so, the MI355X-native batched crowd-navigation
 * simulator that replaces the simulation hot path of kolomeytsev/EB-CADRL.
 *
 * The reference has no FFI: its boundary is the duck-typed Python surface of
 * simulator/env.py (EntityBasedCollisionAvoidance).  Every entry point below
 * names the reference interface (file:line under the reference tree) it stands
 * in for.  The binding a maintainer of the reference would add is the ctypes
 * stub shown in INTEGRATION.md.
 *
 * Conventions
 *   - plain C, no torch types; every function returns EBC_OK (0) or a negative
 *     EbcStatus; the message of the last failure on the calling thread is
 *     ebc_last_error().
 *   - the caller owns every buffer it passes; the library owns the opaque
 *     handle and the device allocations behind it.
 *   - buffers are either all host or all device per call (`location`).  Host
 *     buffers are staged through the handle's own device buffers and the call
 *     returns after the copy-back; device buffers are used in place and the
 *     call only enqueues work on the handle's stream (ebc_synchronize waits).
 *   - batched layout is struct-of-arrays.  E = n_envs, N = max_humans (row
 *     stride of every per-human array, humans in the reference's type-major
 *     order adults, bicycles, children: simulator/env.py:393), S = max_static,
 *     R = N + S observation rows, T = 13 (+4 with agent type) rotated width.
 *   - simulator state is IEEE double (the reference keeps Python floats:
 *     simulator/agents/agent.py:164-228); ORCA runs in float exactly where the
 *     third-party rvo2 library does; rotated observations are float where the
 *     reference builds them with torch.Tensor (rl/policy/cadrl.py:236-337).
 *   - there is no CPU fallback: without a HIP device ebc_create fails.
 */
#ifndef EBCSIM_H
#define EBCSIM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EBC_ABI_VERSION 1

typedef enum EbcStatus {
  EBC_OK = 0,
  EBC_ERR_INVALID = -1,     /* bad argument (NULL, size, range)            */
  EBC_ERR_UNSUPPORTED = -2, /* valid in the reference, not built here      */
  EBC_ERR_DEVICE = -3,      /* HIP runtime error or no device              */
  EBC_ERR_STATE = -4        /* call order (e.g. step before reset)         */
} EbcStatus;

/* simulator/utils/utils.py:9-14 (AgentType) */
enum { EBC_ADULT = 0, EBC_BICYCLE = 1, EBC_CHILD = 2, EBC_ADULT_STATIC = 3, EBC_ROBOT = 4 };

/* Info subclass identity, simulator/utils/info.py, in the priority order of
 * Reward.compute (simulator/utils/reward.py:103-179). */
enum {
  EBC_INFO_NOTHING = 0,
  EBC_INFO_DANGER = 1,
  EBC_INFO_REACH_GOAL = 2,
  EBC_INFO_COLLISION_OBSTACLE = 3,
  EBC_INFO_COLLISION_ADULT = 4,
  EBC_INFO_COLLISION_BICYCLE = 5,
  EBC_INFO_COLLISION_CHILD = 6,
  EBC_INFO_TIMEOUT = 7
};

/* robot kinematics: Agent.kinematics (simulator/agents/agent.py:21, :164-188).
 * HOLONOMIC takes ActionXY(vx, vy); UNICYCLE takes ActionRot(v, r).  ActionXYRot
 * cannot pass compute_collision_agent_with_robot (simulator/utils/collisions.py
 * :41-42 reads action.v) and is therefore not part of the path. */
enum { EBC_HOLONOMIC = 0, EBC_UNICYCLE = 1 };

enum { EBC_HOST = 0, EBC_DEVICE = 1 };

/* who moves the humans this step (simulator/policy/policy_factory.py:10-14) */
enum {
  EBC_HUMAN_EXTERNAL = 0, /* velocities given by ebc_set_human_actions (BASELINE config 2) */
  EBC_HUMAN_LINEAR = 1,   /* simulator/policy/linear.py:17-23 */
  EBC_HUMAN_ORCA = 2,     /* simulator/policy/orca.py:85-157 (+ rvo2) */
  EBC_HUMAN_CACHED = 3    /* re-use the velocities the last ebc_lookahead computed for the same state */
};

enum {
  EBC_ROBOT_EXTERNAL = 0, /* robot_action[E][2] supplied */
  EBC_ROBOT_LINEAR = 1,   /* simulator/policy/linear.py:17-23 applied to the robot, on device */
  EBC_ROBOT_ORCA = 2      /* ebc_step_k only: the robot on ORCA (ebc_robot_orca), the imitation-learning demonstrator */
};

/* flags */
enum {
  EBC_FLAG_AUTO_RESET = 1, /* a terminal step's outputs describe the terminal transition; the env's
                              state is then put back to its reset() scene (time 0) for the next step */
  EBC_FLAG_BORDER = 2      /* border[4] valid: simulator/env.py:264-271 */
};

/* Everything env.configure() / Reward.__init__ / ORCA.__init__ read from the INI
 * files (simulator/env.py:58-87, simulator/utils/reward.py:18-75,
 * simulator/policy/orca.py:58-69) plus the two policy-side switches rotate()
 * consults (rl/policy/cadrl.py:261, :304).  A missing optional key of the
 * reference (value None) is NaN here. */
typedef struct EbcParams {
  uint32_t struct_size; /* = sizeof(EbcParams), checked */
  int32_t robot_kinematics;
  int32_t robot_visible;     /* [robot] visible: humans see the robot (env.py:401-402) */
  int32_t new_reward;        /* reward.py:19 */
  double time_step;          /* [env] time_step */
  double time_limit;         /* [env] time_limit (getint) */
  double map_size_m;         /* [map] map_size_m */
  double map_resolution;     /* [map] map_resolution */
  double time_max;           /* reward.py:20 (NaN when absent) */
  double time_good;          /* reward.py:25 */
  double max_goal_distance;  /* reward.py:21 (NaN when absent: goal_reward None) */
  double success_reward;     /* reward.py:26 */
  double collision_penalty[4]; /* adult, bicycle, child, obstacle: reward.py:27-38 */
  double discomfort_dist[3];   /* adult, bicycle, child: reward.py:40-50 */
  double discomfort_factor[3]; /* adult, bicycle, child: reward.py:52-68 */
  double rotation_penalty_factor; /* reward.py:70 */
  double orca_safety_space;  /* orca.py:63 */
  float orca_neighbor_dist;  /* orca.py:64 */
  float orca_time_horizon;   /* orca.py:66 */
  int32_t orca_max_neighbors; /* orca.py:65 */
  int32_t with_agent_type;   /* sarl.py:103-110: T = 17 instead of 13 */
  int32_t rotate_unicycle;   /* policy kinematics == "unicycle": cadrl.py:261 */
  int32_t reserved;
} EbcParams;

/* Scene rows for ebc_reset: the outputs of SceneGenerator (simulator/scene/
 * scene_generator.py:330-378, :807-863) for n envs, all host pointers.
 * Per-human arrays are [n][N], per-static arrays [n][S], the grid is [n][G][2]
 * 64-bit words (G = round(map_size_m / map_resolution) <= 128): bit y of row x
 * set <=> scene.map[x, y] == 0 (occupied).  grid may be NULL (free map). */
typedef struct EbcScene {
  uint32_t struct_size;
  int32_t n;
  const int32_t *n_humans;   /* [n] */
  const double *px, *py, *vx, *vy, *gx, *gy, *radius, *v_pref; /* [n][N] */
  const uint8_t *type;       /* [n][N] AgentType */
  const int32_t *n_static;   /* [n] (may be NULL when S == 0) */
  const double *spx, *spy, *sradius; /* [n][S] static_obstacles_as_pedestrians */
  const uint64_t *grid;      /* [n][G][2] or NULL */
  const double *robot;       /* [n][9] px,py,vx,vy,radius,gx,gy,v_pref,theta (FullState order, state.py:1-12) */
} EbcScene;

/* One env.step(action, update=True) for every env (simulator/env.py:388-466). */
typedef struct EbcStepArgs {
  uint32_t struct_size;
  int32_t location;     /* of every pointer below */
  int32_t human_policy;
  int32_t robot_policy;
  int32_t flags;
  int32_t reserved;
  const double *robot_action; /* [E][2] ActionXY(vx,vy) or ActionRot(v,r); NULL with EBC_ROBOT_LINEAR */
  const double *border;       /* host [4] = x_lo, x_hi, y_lo, y_hi (always host) */
  /* outputs, each may be NULL */
  double *reward;       /* [E] */
  uint8_t *done;        /* [E] */
  uint8_t *info;        /* [E] EBC_INFO_* */
  double *dmin;         /* [E][3] dmin_adult, dmin_bicycle, dmin_child (inf when none) */
  double *dist_to_goal; /* [E] */
  double *robot_action_out; /* [E][2] the action applied (for EBC_ROBOT_LINEAR) */
  double *human_action; /* [E][N][2] velocities the humans chose this step */
  double *ob;           /* [E][R][5] px,py,vx,vy,radius of the returned ob list (env.py:381-382, :457-458) */
  float *obs_rotated;   /* [E][R][T] rotate(robot full state + ob row), cadrl.py:236-337 */
} EbcStepArgs;

/* The |A|-way env.onestep_lookahead sweep of MultiHumanRL.predict
 * (rl/policy/multi_human_rl.py:38-61) with humans evaluated once. */
typedef struct EbcLookaheadArgs {
  uint32_t struct_size;
  int32_t location;
  int32_t human_policy; /* LINEAR, ORCA or EXTERNAL */
  int32_t n_actions;    /* A */
  int32_t flags;
  int32_t reserved;
  const double *actions; /* [A][2], shared by all envs (cadrl.py:91-116) */
  const double *border;  /* host [4] or NULL */
  double *reward;        /* [E][A] */
  uint8_t *done;         /* [E][A] */
  uint8_t *info;         /* [E][A] */
  double *dmin;          /* [E][A][3] or NULL */
  double *next_ob;       /* [E][R][5] get_next_observable_state rows (env.py:449-458) or NULL */
  float *rows_rotated;   /* [E][A][R][T] rotate(propagate(robot, a) + next row) or NULL */
} EbcLookaheadArgs;

/* Read-back of the simulator state (env.states / Agent.get_full_state). */
typedef struct EbcStateView {
  uint32_t struct_size;
  int32_t location;
  double *px, *py, *vx, *vy, *gx, *gy, *radius, *v_pref; /* [E][N], each may be NULL */
  uint8_t *type;         /* [E][N] */
  int32_t *n_humans;     /* [E] */
  double *robot;         /* [E][9] FullState order */
  double *global_time;   /* [E] env.global_time */
  double *arrival_time;  /* [E][N] adult_times / bicycle_times / children_times (env.py:365-378) */
  uint8_t *done;         /* [E] latched terminal flag */
} EbcStateView;

int ebc_abi_version(void);
const char *ebc_last_error(void);

/* Defaults of the reference (ORCA constants orca.py:58-69, time_good reward.py:25). */
int ebc_params_default(EbcParams *p);

/* gym.make + env.configure + env.set_robot: simulator/__init__.py:4-7,
 * simulator/env.py:58-92.  Allocates device state for E envs on HIP device
 * `device_id`. */
int ebc_create(int device_id, int n_envs, int max_humans, int max_static,
               const EbcParams *params, void **handle_out);
int ebc_destroy(void *handle);

/* Run the handle's work on an existing HIP stream (e.g. torch's current
 * stream) so the caller's events order with it.  NULL is the HIP null stream
 * (torch's default stream).  Until this is called the handle uses a private
 * non-blocking stream created by ebc_create. */
int ebc_set_stream(void *handle, void *hip_stream);
int ebc_synchronize(void *handle);

/* env.reset for the listed envs (simulator/env.py:128-205): uploads the scene
 * rows, zeroes global_time and arrival times (env.py:149-151) and keeps a copy
 * for EBC_FLAG_AUTO_RESET.  env_ids NULL = envs 0..n-1. */
int ebc_reset(void *handle, const int32_t *env_ids, const EbcScene *scene);

/* Device-resident scene pool for EBC_FLAG_AUTO_RESET (the reference generates a fresh scene at
 * every env.reset on the host, simulator/scene/scene_generator.py:330-378; with thousands of envs
 * resets would dominate).  `pool` holds P host-generated scenes (same layout as for ebc_reset,
 * scene.n = P).  Env e restarts from scene cursor[e] = e mod P, and the cursor then advances by
 * `stride` (mod P), entirely on the device.  Without this call the pool is the envs' own
 * ebc_reset scenes (P = E, stride 0).  May be called again while episodes run (a smaller, larger
 * or regenerated pool): running episodes are not touched — an env that is in the middle of an
 * episode on a scene of the OLD pool keeps that scene's occupancy grid (copied into the env's own
 * slot before the old pool is freed) and its static rows until its next restart; the cursors are
 * set back to e mod P. */
int ebc_set_scene_pool(void *handle, const EbcScene *pool, int stride);

/* ---- SceneGenerator.generate_random_scene on the device (simulator/scene/scene_generator.py:330-378) ----
 * The keys SceneGenerator.__init__ reads (:20-72) and the per-type agent sections (agents/agent.py:16-35), with
 * the phase already resolved by the caller: count[t] / rule[t] are the number of humans of AgentType t and its
 * crossing rule for THIS phase (test_sim_* or train_val_sim_*; 1 human per type when neither `test` nor
 * multiagent_training, :343-355). */
enum EbcCrossingRule {
  EBC_RULE_CIRCLE_CROSSING = 0,     /* scene_generator.py:593-648 (adults, bicycles) */
  EBC_RULE_SQUARE_CROSSING = 1,     /* :650-712 */
  EBC_RULE_SQUARE_CROSSING_OLD = 2  /* :714-761 (bicycles only) */
};
#define EBC_GEN_STATIC_OVERFLOW 1 /* a generated map has more observation rows than max_static */

typedef struct EbcSceneGen {
  uint32_t struct_size;
  int32_t randomize_attributes;      /* [env] randomize_attributes */
  int32_t count[3];                  /* adults, bicycles, children */
  int32_t rule[3];                   /* EbcCrossingRule per type */
  double radius[3], v_pref[3];       /* fixed attributes ([adults] / [bicycles] / [children]) */
  double radius_min[3], radius_max[3], v_pref_min[3], v_pref_max[3]; /* sample_random_attributes, agent.py:48-56 */
  double square_width, circle_radius; /* [sim] */
  double discomfort_dist;            /* [reward] discomfort_dist (the rejection tests' margin) */
  double robot_radius, robot_v_pref; /* [robot] */
  double map_resolution, map_size_m; /* [map] */
  int32_t min_wall_length, max_wall_length, num_circles, num_walls;
} EbcSceneGen;

/* Arrays a generated batch is copied out to (ebc_generate_scenes): an EbcScene's, writable. */
typedef struct EbcSceneOut {
  uint32_t struct_size;
  int32_t location;          /* EBC_HOST or EBC_DEVICE, of every pointer below */
  int32_t *n_humans;
  double *px, *py, *vx, *vy, *gx, *gy, *radius, *v_pref;
  uint8_t *type;
  int32_t *n_static;         /* NULL allowed when max_static == 0 */
  double *spx, *spy, *sradius;
  uint64_t *grid;            /* [n][G][2] or NULL */
  double *robot;
} EbcSceneOut;

/* n scenes generated on the device, one per seed: scene i is what the reference's generate_random_scene draws after
 * np.random.seed(seeds[i]) (seeds NULL: seed0 + i; env.reset seeds with counter_offset[phase] + case,
 * simulator/env.py:153-169) — numpy's legacy MT19937 stream, draw for draw, so every value is the reference's bit
 * for bit, except the cos / sin of circle_crossing (<= 1 ulp: numpy's own results there depend on the CPU).
 * Shapes follow the handle (max_humans, max_static, grid width); sum(count) must fit max_humans; a map with more
 * observation rows than max_static is EBC_ERR_INVALID.
 *   ebc_generate_scenes: copy the batch out (tests, inspection).
 *   ebc_generate_reset:  env.reset of envs first..first+n-1 from it, nothing crossing PCIe but the seeds.
 *   ebc_generate_pool:   install it as the EBC_FLAG_AUTO_RESET pool (as ebc_set_scene_pool, P = n). */
int ebc_generate_scenes(void *handle, const EbcSceneGen *gen, uint32_t seed0, const uint32_t *seeds, int n, EbcSceneOut *out);
int ebc_generate_reset(void *handle, const EbcSceneGen *gen, uint32_t seed0, const uint32_t *seeds, int first, int n);
int ebc_generate_pool(void *handle, const EbcSceneGen *gen, uint32_t seed0, const uint32_t *seeds, int n, int stride);

/* Humans moved by the host (BASELINE config 2): act[E][N][2]. */
int ebc_set_human_actions(void *handle, int location, const double *act);

/* The observation of the CURRENT state, without stepping: what env.reset returns
 * (simulator/env.py:188-193) and what MultiHumanRL.transform builds from it
 * (rl/policy/multi_human_rl.py:128-149).  ob [E][R][5], obs_rotated [E][R][T]; either may be NULL. */
int ebc_observe(void *handle, int location, double *ob, float *obs_rotated);

/* ORCA.predict with the ROBOT as the agent, for every env (simulator/policy/orca.py:85-157 called
 * through Robot.act, simulator/agents/robot.py:16-25): the demonstrator of the imitation-learning
 * stage (rl/train.py:99-143, which sets the policy's safety_space itself; rl/utils/explorer.py:33-45).
 * The others are the rows of the observation in their order: humans, then the static obstacles as
 * pedestrians.  action [E][2] = ActionXY; feed it to ebc_step (EBC_ROBOT_EXTERNAL).  Holonomic robots,
 * N + S <= 32. */
int ebc_robot_orca(void *handle, double safety_space, int location, double *action);

/* The demonstrator's PERSISTENT rvo2 simulator (simulator/policy/orca.py:96-133).  The reference's ORCA policy object
 * keeps one simulator for all its predict() calls and rebuilds it only when the number of agents changes
 * (orca.py:96-101); otherwise it updates positions and velocities only (:129-133), so the rows' radii
 * (+ 0.01 + safety_space), the robot's radius and its maxSpeed stay those of the call that BUILT the simulator — also
 * across env.reset(): rl/train.py:130-133 makes ONE il_policy for every imitation-learning episode, and with
 * randomize_attributes = true episodes 2... run on the radii of the episode that built it.
 * enable != 0: from now on ebc_robot_orca and EBC_ROBOT_ORCA (ebc_step_k) of this handle behave like that, one
 * simulator per env (an env = one policy object playing its episodes one after the other), all of them "not built
 * yet" (= a fresh policy object; call it again for the next one).  enable == 0 (the default): every call works from
 * the current state alone (= a fresh policy object per call).  ebc_reset and restarts do not touch the simulators. */
int ebc_robot_orca_sim(void *handle, int enable);
/* Read (set == 0) or replace (set != 0) the simulators: rows [E] (number of agents besides the robot, -1 = not built),
 * radius [E][N + S] (float, as rvo2 holds them: row j of the observation), self [E][2] = {radius, maxSpeed}.  The
 * single-env facade keeps a policy object's simulator in the object and hands it to whichever handle serves it. */
int ebc_robot_orca_sim_state(void *handle, int location, int set, int32_t *rows, float *radius, float *self);

/* One env.step for every env (simulator/env.py:388-466), enqueued on the handle's stream.  With
 * EBC_HUMAN_ORCA it is ONE kernel launch whose arguments change from call to call (the robot state
 * is double-buffered and a launch counter travels with the launch): call it once per step; do not
 * capture it in a HIP graph and replay it (a call on a stream under capture returns EBC_ERR_UNSUPPORTED and
 * records nothing).  A broken hand-off inside that launch (never seen)
 * surfaces as EBC_ERR_DEVICE — not as a hang — from ebc_synchronize or from the next host-location
 * call (whose copy-back carries the fault word); it is reported once, the handle then refuses
 * steps (EBC_ERR_STATE) until ebc_reset has re-armed it. */
int ebc_step(void *handle, const EbcStepArgs *args);
int ebc_lookahead(void *handle, const EbcLookaheadArgs *args);

/* K consecutive env.step calls for every env with a robot policy that lives on the device, in ONE call: the
 * episode loop of Explorer.run_one_episode (rl/utils/explorer.py:33-45: act -> step -> keep state, action,
 * reward) without a host round trip per step — with EBC_ROBOT_ORCA the imitation-learning rollouts of
 * rl/train.py:124-133.  Per step k the library enqueues: the rotated joint state the policy sees
 * (MultiHumanRL.transform, rl/policy/multi_human_rl.py:128-149 -> state_rotated[k]), the robot's action
 * (EBC_ROBOT_ORCA: ebc_robot_orca with robot_safety_space; EBC_ROBOT_LINEAR; EBC_ROBOT_EXTERNAL:
 * robot_action[k]), and ebc_step with every output written at index k.  Same arithmetic, launch for launch,
 * as K calls of ebc_observe / ebc_robot_orca / ebc_step (tests hold it equal to K oracle steps); what it
 * removes is the per-step host work.  EBC_FLAG_AUTO_RESET keeps every env in an episode. */
typedef struct EbcStepKArgs {
  uint32_t struct_size;
  int32_t location;      /* of every pointer below */
  int32_t K;             /* steps, >= 1 */
  int32_t human_policy;  /* EBC_HUMAN_ORCA, _LINEAR or _EXTERNAL (the same supplied velocities every step) */
  int32_t robot_policy;  /* EBC_ROBOT_ORCA, _LINEAR or _EXTERNAL */
  int32_t flags;
  double robot_safety_space;  /* EBC_ROBOT_ORCA: the policy's safety_space (rl/train.py:127-129) */
  const double *robot_action; /* [K][E][2], EBC_ROBOT_EXTERNAL only */
  /* outputs, each may be NULL */
  float *state_rotated;     /* [K][E][R][T] rotated joint state BEFORE step k (policy.last_state, explorer.py:43) */
  long long *n_rows;        /* [K][E] observation rows that exist before step k (ebc_row_counts) */
  double *robot_action_out; /* [K][E][2] the action taken at step k */
  double *reward;           /* [K][E] */
  uint8_t *done;            /* [K][E] */
  uint8_t *info;            /* [K][E] */
  double *dmin;             /* [K][E][3] */
  double *dist_to_goal;     /* [K][E] */
  float *obs_rotated;       /* [K][E][R][T] rotated observation returned by step k */
} EbcStepKArgs;
int ebc_step_k(void *handle, const EbcStepKArgs *args);
int ebc_get_state(void *handle, const EbcStateView *view);

/* Rows of the observation that exist, per env: len(ob) of the list env.step returns (humans, then the
 * static obstacles as pedestrians: simulator/env.py:381-382, :457-458) — the first dimension of the
 * state tensor MultiHumanRL.transform builds (rl/policy/multi_human_rl.py:128-149).  n_rows [E] int64
 * (what ebc_pair_mean / ebc_pair_attend take as n_valid).  Read from the device state, so it follows
 * auto-reset restarts from a scene pool whose scenes differ in size. */
int ebc_row_counts(void *handle, int location, long long *n_rows);

/* Imitation-learning value targets of a rollout window (rl/utils/explorer.py:159-170 over the episodes of
 * ebc_step_k's outputs): value[t][e] = sum over the rest of ITS episode of gamma_bar^(t' - t) reward[t'][e], walked
 * backwards per env; keep[t][e] = 1 when that episode ended inside the window in ReachGoal or a collision (the
 * reference stores whole episodes, successes and collisions only: explorer.py:82-92).  reward float64, done / info
 * uint8 [K][E] (device), values float64 [K][E], keep uint8 [K][E]; stream: hipStream_t. */
int ebc_il_targets(void *stream, const double *reward, const uint8_t *done, const uint8_t *info, int K, int E,
                   double gamma_bar, double *values, uint8_t *keep);

/* Geometry of the handle (E, N, S, T, G). */
int ebc_dims(void *handle, int32_t out[5]);

/* Timing aid for bench.py: average device milliseconds per ebc_step kernel
 * launch since the last call with reset != 0, measured with HIP events on the
 * stream the kernel runs on.  Enabled by ebc_timing(handle, 1). */
int ebc_timing(void *handle, int enable);
int ebc_timing_read(void *handle, int reset, double *avg_ms, int64_t *launches);

/* ---- value-network layers (the consumer of ebc_lookahead's rows) -------------------------------
 * A two-layer block of the reference's mlp() helper (rl/policy/cadrl.py:13-21: Linear + ReLU stacks;
 * rl/policy/sarl.py:25-35 builds mlp1 / mlp2 / attention / mlp3 from it):
 *     y = [relu] (W2 relu(W1 x + b1) + b2),   x [M][K0] -> y [M][O]   (float32, device)
 * on the bf16 matrix cores with every float32 operand split in two bf16 numbers (three products kept):
 * float32-grade results (tests/test_value_net.py: 4e-5 relative; 1.5e-5 on the value of the reference's
 * decision runs) at several times the float32 GEMM rate, the hidden layer never leaving the registers.
 * w1 [H][K0], b1 [H], w2 [O][H], b2 [O]: host pointers, torch.nn.Linear layout.  K0, O <= 224.
 * Optional third layer with ONE output (the attention score): w3 [O], b3 [1] (NULL = none); forward
 * then writes y [M] = w3 . relu(W2 ... + b2) + b3 and the [M][O] activations never reach memory. */
int ebc_mlp2_create(int device_id, int K0, int H, int O, const float *w1, const float *b1, const float *w2,
                    const float *b2, const float *w3, const float *b3, void **mlp_out);
/* x, y: device pointers; stream: hipStream_t (NULL = the null stream); relu_out: ReLU on y.
 * row_bias (device, NULL = none) [M / group_rows][H]: added to the hidden pre-activation of every row of
 * a group of group_rows consecutive rows — ValueNetwork's attention input cat([h, mean(h)]) (sarl.py:56-62)
 * without the concatenation: the mean's half of the first attention layer acts once per pair. */
int ebc_mlp2_forward(void *mlp, void *stream, const float *x, int M, int relu_out, const float *row_bias,
                     int group_rows, float *y);
int ebc_mlp2_destroy(void *mlp);

/* The per-pair glue of ValueNetwork.forward between those blocks (rl/policy/sarl.py:52-78), one pass over
 * the activations each.  A pair = one joint state = R consecutive rows, the first n_valid[b] of which exist
 * (device int64 [B]; NULL = all R).  Device pointers, 16-byte aligned; H and F multiples of 4, F <= 256.
 *   ebc_pair_mean:   g [B][H] = mean over the pair's rows of h [B*R][H]            (sarl.py:56-58)
 *   ebc_pair_attend: out [B][F] = sum_r softmax'(scores)[b][r] feat [B*R][F], softmax' = exp(s) (s != 0)
 *                    normalised over the pair's rows                                 (sarl.py:69-76) */
int ebc_pair_mean(void *stream, const float *h, const long long *n_valid, int B, int R, int H, float *g);
int ebc_pair_attend(void *stream, const float *scores, const float *feat, const long long *n_valid, int B, int R,
                    int F, float *out);

/* Activations handed from block to block already split and in fragment order (round 3).  A block created with
 * EBC_MLP_IN_FRAGMENTS takes its input as the tensor another block wrote with `frag_out`: per 32-row tile and 32-column
 * tile, the two k-steps' hi / lo bf16 fragments as the matrix cores take them (16 bytes per lane and piece; the same 4
 * bytes per element as float32 rows, rows and columns padded to 32) — the consumer's input phase is 16-byte loads,
 * no transposition through LDS and no per-lane splitting.  Both sides tile the rows from row 0 in steps of 32.
 * ebc_mlp2_forward_ex: every form of the forward in one call: x (rows) or frag_in; y (rows, may be NULL), partial /
 * seg_rows / row_weight as in ebc_mlp2_forward_reduce, frag_out [ceil(M / 32)][ceil(O / 32)][2][2][64] x 16 bytes. */
#define EBC_MLP_IN_FRAGMENTS 1
#define EBC_MLP_GENERAL_KERNEL 1
typedef struct EbcMlpArgs {
  uint32_t struct_size;
  int32_t M, relu_out, group_rows, seg_rows;
  int32_t flags; /* EBC_MLP_GENERAL_KERNEL: the general block even where a specialised kernel exists (the streamed
                    attention block of ebc_vn_stream.h takes fragment input + row_bias + a one-output third layer at
                    7 x 7 x 7 tiles): same values bit for bit — for measurements and for the test that says so */
  const float *x;
  const void *frag_in;
  const float *row_bias;
  const float *row_weight;
  float *y;
  double *partial;
  void *frag_out;
} EbcMlpArgs;
int ebc_mlp2_create_ex(int device_id, int K0, int H, int O, const float *w1, const float *b1, const float *w2,
                       const float *b2, const float *w3, const float *b3, int flags, void **mlp_out);
int ebc_mlp2_forward_ex(void *mlp, void *stream, const EbcMlpArgs *args);

/* Refresh an existing block from weights in DEVICE memory (torch Linear layout, the shapes it was created with),
 * enqueued on `stream`: the packed split-bf16 fragments, the biases and the float32 copies are rebuilt by a kernel,
 * bit-equal to what ebc_mlp2_create packs on the host.  The training loop of rl/train.py:239-259 changes the network
 * every round; its rollouts are decisions like any others and run on the matrix-core blocks after one such call per
 * block and round.  w3 / b3: the one-output third layer, for blocks created with one (b3 is read back: one sync). */
int ebc_mlp2_update(void *mlp, void *stream, const float *w1, const float *b1, const float *w2, const float *b2,
                    const float *w3, const float *b3);

/* The same block in plain float32 on the vector ALUs (every product an fmaf into a float32 sum, like a float32 GEMM):
 * for the few rows whose value decides an argmax — SarlValueNet.action_values re-evaluates the candidates that lie
 * within the split-bf16 error of the best one (rl/policy/multi_human_rl.py:61-80 takes the maximum of float32 values).
 * Same arguments and outputs as ebc_mlp2_forward. */
int ebc_mlp2_forward_f32(void *mlp, void *stream, const float *x, int M, int relu_out, const float *row_bias,
                         int group_rows, float *y);

/* The same two reductions folded into the block that PRODUCES the activations, so that no second pass reads them
 * (and the features the attention weights multiply are never written at all):
 *   ebc_mlp2_forward_reduce: ebc_mlp2_forward, and per 32-row tile the row-weighted sums of y over each group of
 *     seg_rows (16 <= seg_rows, = R) consecutive rows the tile touches: partial [ceil(M / 32)][3][O] (device, float64).
 *     row_weight [M] (device; NULL = 1).  y NULL: the activations are not stored.  O a multiple of 4.
 *   ebc_pair_weights: w [B][R] = softmax'(scores) of ebc_pair_attend — the row weights of the feature block.
 *   ebc_pair_mask:    w [B][R] = 1 for rows < n_valid[b], else 0 — the row weights of a masked mean.
 *   ebc_pair_combine: out [B][O] = the sum of pair b's partials (two or three tiles), / n_valid[b] when mean != 0:
 *     with row weights 1 (or the mask) the pair mean of sarl.py:56-58, with ebc_pair_weights the weighted feature
 *     sum of sarl.py:73-76.  The sums are carried in float64 (float32 x float32 products are exact there) and rounded
 *     to float32 once: the result does not depend on where in the batch a pair's rows lie, and equals the one-pass
 *     kernels' to float32 rounding. */
int ebc_mlp2_forward_reduce(void *mlp, void *stream, const float *x, int M, int relu_out, const float *row_bias,
                            int group_rows, float *y, int seg_rows, const float *row_weight, double *partial);
int ebc_pair_weights(void *stream, const float *scores, const long long *n_valid, int B, int R, float *w);

/* The argmax side of MultiHumanRL.predict for a batch (rl/policy/multi_human_rl.py:72-80), one launch:
 *   values [E][A] (float64) = reward + discount * v      (v [E][A]: the value network's float32 outputs)
 *   order  [E][A] (int32)   = each env's actions by value, best first (equal values: the lower action index first)
 *   count  [E]    (int32)   = how many of them lie within `bound` of the env's best: the prefix of `order` that can
 *                             hold the float32 network's best action when the values carry an error of bound / 2
 * Device pointers; A <= 1024.  (What ebcsim.sarl.SarlValueNet.action_values selects its re-evaluated candidates from.) */
int ebc_decision_rank(void *stream, const float *v, const double *reward, double discount, double bound, int E, int A,
                      double *values, int32_t *order, int32_t *count);
/* The re-evaluated candidates back into the values, one launch: for i < n, values[env[i]][act[i]] = reward[env[i]][act[i]] +
 * discount * exact[i] (exact: the float32 network's values of the n candidates; env / act: int64 indices), and
 * *worst (float, zeroed by the caller) = max_i |exact[i] - v[env[i]][act[i]]| — the error of the matrix-core values
 * where it matters, which the caller holds against its bound.  Device pointers. */
int ebc_decision_apply(void *stream, const float *exact, const float *v, const long long *env, const long long *act,
                       const double *reward, double discount, int A, int n, double *values, float *worst);
int ebc_pair_mask(void *stream, const long long *n_valid, int B, int R, float *w);
int ebc_pair_combine(void *stream, const double *partial, const long long *n_valid, int B, int R, int O, int mean,
                     float *out);

#ifdef __cplusplus
}
#endif
#endif /* EBCSIM_H */
