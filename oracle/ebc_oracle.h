/*
 * ebc_oracle.h — TEST INFRASTRUCTURE, NOT PRODUCT.
 *
 * Scalar CPU restatement of the EB-CADRL simulation hot path, used only as the
 * checker in tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
 * Nothing under eb-cadrl_amd/ may include, link or load it.
 *
 * Parity status: every function except orc_orca() is pinned against golden
 * vectors produced by importing the reference itself (tests/golden/, script
 * tests/golden/make_golden.py).  orc_orca() restates the published RVO2 v2
 * algorithm (third-party `rvo2`, un-vendored and un-pinned by the reference,
 * absent from this image): ORCA float parity is UNPINNED; it is anchored only
 * on the reference's call site (simulator/policy/orca.py:85-157) and on the
 * terminal-outcome classes of tests/test_collisions_simulation.py:12-32.
 */
#ifndef EBC_ORACLE_H
#define EBC_ORACLE_H

#include "../include/ebcsim.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Batched simulator state, same SoA layout as the product (include/ebcsim.h). */
typedef struct OrcState {
  int32_t E, N, S, G;
  int32_t *n_humans;                                       /* [E] */
  double *px, *py, *vx, *vy, *gx, *gy, *radius, *v_pref;   /* [E][N] */
  uint8_t *type;                                           /* [E][N] */
  int32_t *n_static;                                       /* [E] */
  double *spx, *spy, *sradius;                             /* [E][S] */
  int32_t *grid_scene;                                     /* [E] pool slot whose grid env e uses */
  double *robot;                                           /* [E][9] */
  double *global_time;                                     /* [E] */
  double *arrival_time;                                    /* [E][N] */
  uint8_t *done;                                           /* [E] */
  double *human_action;                                    /* [E][N][2] external / cached */
  /* scene pool for EBC_FLAG_AUTO_RESET, same layout: slots [0, E) = the envs' reset() scenes,
   * slots [E, E + P) = the pool of ebc_set_scene_pool.  Env e restarts from slot cursor[e]: itself
   * when P == 0, else it walks E + (e mod P), + stride, ... (mod P) */
  int32_t P, stride;
  int32_t *cursor;                                         /* [E] */
  int32_t *p_n_humans;                                     /* [E + P] */
  double *p_px, *p_py, *p_vx, *p_vy, *p_gx, *p_gy, *p_radius, *p_v_pref; /* [P][N] */
  uint8_t *p_type;                                         /* [P][N] */
  int32_t *p_n_static;                                     /* [P] */
  double *p_spx, *p_spy, *p_sradius;                       /* [P][S] */
  uint64_t *p_grid;                                        /* [P][G][2] or NULL */
  double *p_robot;                                         /* [P][9] */
} OrcState;

double orc_point_to_segment_dist(double x1, double y1, double x2, double y2, double x3, double y3);

/* returns collision flag, lowers *dmin like the reference */
int orc_collision_agent_robot(double hpx, double hpy, double hvx, double hvy, double hr,
                              double rpx, double rpy, double rtheta, double rr,
                              int kinematics, double a0, double a1, double dt, double *dmin);

int orc_grid_collision(const uint64_t *grid, int G, double map_size_m, double map_resolution,
                       double px, double py, double robot_radius, const double *border);

void orc_robot_next_position(const double *robot, int kinematics, double a0, double a1, double dt,
                             double *nx, double *ny);

void orc_reward(const EbcParams *p, const double *robot, double a0, double a1, double global_time,
                const double dmin[3], const int coll[4], double *reward, uint8_t *done,
                uint8_t *info, double *dist_to_goal);

void orc_linear(double px, double py, double gx, double gy, double v_pref, double *vx, double *vy);

/* self + n_others (ob order); others' arrays hold px,py,vx,vy,radius */
void orc_orca(const EbcParams *p, double px, double py, double vx, double vy, double radius,
              double gx, double gy, double v_pref, int n_others, const double *opx,
              const double *opy, const double *ovx, const double *ovy, const double *oradius,
              double *out_vx, double *out_vy);

/* agent 0 of one rvo2 doStep on raw float state (others = agents 1..n) */
void orc_rvo2_agent0(float timeStep, float neighborDist, int maxNeighbors, float timeHorizon,
                     const float pos0[2], const float vel0[2], float radius0, float maxSpeed0,
                     const float pref0[2], int n_others, const float *opos, const float *ovel,
                     const float *oradius, float out[2]);

/* in[15] = px,py,vx,vy,r,gx,gy,v_pref,theta | px1,py1,vx1,vy1,r1,type ; out[T] */
void orc_rotate_row(const double in[15], int with_agent_type, int rotate_unicycle, float *out);

void orc_propagate_robot(const double *robot, int kinematics, double a0, double a1, double dt,
                         double next[9]);

int orc_observe(const EbcParams *p, const OrcState *s, double *ob, float *obs_rotated);
int orc_step(const EbcParams *p, OrcState *s, const EbcStepArgs *a);
int orc_lookahead(const EbcParams *p, OrcState *s, const EbcLookaheadArgs *a);
/* ORCA.predict with the robot as the agent: stateless, and with the policy object's persistent rvo2 simulator
 * (sim_rows [E] (-1 = none yet), sim_radius [E][N + S], sim_self [E][2]) */
int orc_robot_orca(const EbcParams *p, const OrcState *s, double safety_space, double *action);
int orc_robot_orca_sim(const EbcParams *p, const OrcState *s, double safety_space, int32_t *sim_rows,
                       float *sim_radius, float *sim_self, double *action);

#ifdef __cplusplus
}
#endif
#endif
