// ebc_scene_gen.h — SceneGenerator.generate_random_scene on the device, one lane per scene.
//
// What the reference does on the host at every env.reset (simulator/scene/scene_generator.py:330-378):
// seed numpy's legacy MT19937 stream with the scene's number, place the humans type by type under the
// configured crossing rule with rejection sampling (circle_crossing :593-648, square_crossing :650-712,
// square_crossing_old :714-761, attributes agents/agent.py:48-56), draw the static map (circles and walls,
// :109-328), rasterise it (place_obstacles_on_map :888-922) and turn the obstacles into observation rows
// (create_observation_from_static_obstacles :380-422).  Here a lane owns one scene: its MT19937 state
// lives in a strided global scratch array, every draw is numpy's legacy draw (random_sample = 53 bits of
// two words, uniform = low + (high - low) * u, randint = masked rejection on 32-bit words, choice = randint
// or the cumulative-probability search), every rejection test the reference's, in its order.  Seed s here
// is therefore scene s there: all of it bit for bit except cos/sin of circle_crossing, which numpy
// evaluates with a CPU-dependent SIMD routine (<= 1 ulp); this file's sincos_dd is correctly rounded in
// all but ~2^-12 of the cases, so circle positions agree with a given host to <= 1 ulp (tests state it).
//
// The file is plain C++ with the host/device qualifier as a macro: tests/ compile it with g++ and check
// it against ebcsim/scene.py on the CPU; the product compiles it with hipcc into scene_gen_kernel.
#pragma once

#include <stdint.h>
#include <math.h>

#include "../../include/ebcsim.h"

#ifdef __HIPCC__
#define EBC_HD __host__ __device__ __forceinline__
#else
#define EBC_HD inline
#endif

namespace ebc {

// numpy.random.RandomState(seed) for an integer seed: init_genrand (Knuth's multiplier), state word i of
// scene r at mt[i * stride + r]
struct Mt19937 {
  uint32_t *mt;
  size_t stride;
  int mti;

  EBC_HD uint32_t &at(int i) { return mt[(size_t)i * stride]; }
  EBC_HD void seed(uint32_t s) {
    at(0) = s;
    for (int i = 1; i < 624; ++i) {
      s = 1812433253u * (s ^ (s >> 30)) + (uint32_t)i;
      at(i) = s;
    }
    mti = 624;
  }
  EBC_HD void twist() {
    for (int k = 0; k < 624; ++k) {
      const uint32_t y = (at(k) & 0x80000000u) | (at(k + 1 < 624 ? k + 1 : 0) & 0x7fffffffu);
      const int m = k + 397 < 624 ? k + 397 : k + 397 - 624;
      at(k) = at(m) ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    mti = 0;
  }
  EBC_HD uint32_t next32() {
    if (mti >= 624) twist();
    uint32_t y = at(mti++);
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
  }
  // RandomState.random_sample
  EBC_HD double next_double() {
    const uint32_t a = next32() >> 5, b = next32() >> 6;
    return (a * 67108864.0 + b) / 9007199254740992.0;
  }
  // RandomState.uniform(low, high)
  EBC_HD double uniform(double low, double high) { return low + (high - low) * next_double(); }
  // RandomState.randint(low, high) with the default dtype, high - 1 - low < 2^32: masked rejection on 32-bit words
  EBC_HD long randint(long low, long high) {
    const uint32_t rng = (uint32_t)(high - 1 - low);
    if (rng == 0) return low;
    uint32_t mask = rng;
    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
    uint32_t v;
    do {
      v = next32() & mask;
    } while (v > rng);
    return low + (long)v;
  }
};

// ---- cos and sin of one angle in [0, 2 pi], in double-double, rounded once at the end -----------------------
struct DD {
  double hi, lo;
};
EBC_HD DD dd_two_sum(double a, double b) {
  const double s = a + b, bb = s - a;
  return DD{s, (a - (s - bb)) + (b - bb)};
}
EBC_HD DD dd_fast_two_sum(double a, double b) {
  const double s = a + b;
  return DD{s, b - (s - a)};
}
EBC_HD DD dd_add(DD a, DD b) {
  DD s = dd_two_sum(a.hi, b.hi);
  const DD t = dd_two_sum(a.lo, b.lo);
  s.lo += t.hi;
  s = dd_fast_two_sum(s.hi, s.lo);
  s.lo += t.lo;
  return dd_fast_two_sum(s.hi, s.lo);
}
EBC_HD DD dd_mul(DD a, DD b) {
  const double p = a.hi * b.hi;
  const double e = fma(a.hi, b.hi, -p) + (a.hi * b.lo + a.lo * b.hi);
  return dd_fast_two_sum(p, e);
}
EBC_HD DD dd_mul_d(DD a, double b) {
  const double p = a.hi * b;
  const double e = fma(a.hi, b, -p) + a.lo * b;
  return dd_fast_two_sum(p, e);
}
// 1/n! for the odd and even Taylor terms as double-double would be overkill past the first few terms:
// the terms are divided in turn instead (t_k = t_(k-1) * r^2 / ((2k)(2k+1))), each division in double-double
EBC_HD DD dd_div_d(DD a, double b) {
  const double q1 = a.hi / b;
  const double p = q1 * b;
  const double e = fma(q1, b, -p);
  const double q2 = ((a.hi - p) - e + a.lo) / b;
  return dd_fast_two_sum(q1, q2);
}
EBC_HD void sincos_dd(double angle, double &c_out, double &s_out) {
  // pi/2 to 107 bits
  const double P_HI = 1.5707963267948966, P_LO = 6.123233995736766e-17;
  const int k = (int)nearbyint(angle / P_HI);  // 0..4
  // r = angle - k pi/2: k * P_HI is not exact for k = 3, so its rounding error is subtracted too, then k * P_LO
  const double p = (double)k * P_HI, pe = fma((double)k, P_HI, -p);
  DD r = dd_two_sum(angle, -p);
  r = dd_add(r, DD{-pe, 0.0});
  r = dd_add(r, dd_mul_d(DD{P_LO, 0.0}, -(double)k));
  const DD r2 = dd_mul(r, r);
  // sin r = r - r^3/3! + ..., cos r = 1 - r^2/2! + ...; |r| <= pi/4 + eps: 14 terms reach 2^-100
  DD ts = r, tc = DD{1.0, 0.0}, ss = r, cs = DD{1.0, 0.0};
  for (int j = 1; j <= 14; ++j) {
    tc = dd_div_d(dd_mul(tc, r2), -(double)((2 * j - 1) * (2 * j)));
    ts = dd_div_d(dd_mul(ts, r2), -(double)((2 * j) * (2 * j + 1)));
    cs = dd_add(cs, tc);
    ss = dd_add(ss, ts);
  }
  const double C = cs.hi + cs.lo, S = ss.hi + ss.lo;
  switch (k & 3) {
    case 0: c_out = C; s_out = S; break;
    case 1: c_out = -S; s_out = C; break;
    case 2: c_out = -C; s_out = -S; break;
    default: c_out = S; s_out = -C; break;
  }
}

// np.linalg.norm((x, y)) as the reference's rejection tests evaluate it: a two-element BLAS dot (the second
// product fused into the sum) and a square root — the same form as ebc::norm2 (ebc_device.h)
EBC_HD double gen_hyp(double x, double y) { return sqrt(fma(y, y, x * x)); }
// Python's round() on a float: half to even
EBC_HD long py_round(double x) { return (long)nearbyint(x); }

#define EBC_GEN_MAX_TRIES 100000  // scene_generator.py:11

// Destination rows of one scene (row r of arrays shaped like an EbcScene's)
struct SceneRow {
  int *n_humans;
  double *px, *py, *vx, *vy, *gx, *gy, *radius, *v_pref;
  uint8_t *type;
  int *n_static;
  double *spx, *spy, *sradius;
  uint64_t *grid;  // [G][2] of this scene, or nullptr
  double *robot;   // [9]
};

struct SceneGenCore {
  const EbcSceneGen &c;
  Mt19937 &rs;
  const SceneRow &o;
  int N, S, G;
  double rpx, rpy, rgx, rgy;  // env.reset: robot.set(0, -R, 0, R, 0, 0, pi/2), simulator/env.py:159-161

  // rejection tests see the robot and the humans of the same type placed so far (`first..count`)
  EBC_HD bool clash_pos(double px, double py, double rad, int first, int count, bool with_goal) const {
    for (int q = -1; q < count; ++q) {
      const double opx = q < 0 ? rpx : o.px[first + q], opy = q < 0 ? rpy : o.py[first + q];
      const double orad = q < 0 ? c.robot_radius : o.radius[first + q];
      const double lim = rad + orad + c.discomfort_dist;
      if (gen_hyp(px - opx, py - opy) < lim) return true;
      if (with_goal) {
        const double ogx = q < 0 ? rgx : o.gx[first + q], ogy = q < 0 ? rgy : o.gy[first + q];
        if (gen_hyp(px - ogx, py - ogy) < lim) return true;
      }
    }
    return false;
  }

  EBC_HD void place(int slot, int kind, int first, int count) {
    double radius = c.radius[kind], v_pref = c.v_pref[kind];
    if (c.randomize_attributes) {  // Agent.sample_random_attributes, agent.py:48-56
      v_pref = rs.uniform(c.v_pref_min[kind], c.v_pref_max[kind]);
      radius = rs.uniform(c.radius_min[kind], c.radius_max[kind]);
    }
    double px = 0, py = 0, gx = 0, gy = 0;
    const int rule = c.rule[kind];
    if (rule == EBC_RULE_CIRCLE_CROSSING) {  // scene_generator.py:593-648
      for (int t = 0; t < EBC_GEN_MAX_TRIES; ++t) {
        const double angle = rs.next_double() * 3.141592653589793 * 2;
        double cs, sn;
        sincos_dd(angle, cs, sn);
        px = c.circle_radius * cs + 0;
        py = c.circle_radius * sn + 0;
        if (!clash_pos(px, py, radius, first, count, true)) break;
      }
      gx = -px;
      gy = -py;
    } else if (rule == EBC_RULE_SQUARE_CROSSING) {  // :650-712
      const double hw = c.square_width / 2;
      for (int t = 0; t < EBC_GEN_MAX_TRIES; ++t) {
        const long side = rs.randint(0, 4);  // choice(["top", "bottom", "left", "right"])
        const double u = rs.uniform(-hw, hw);
        px = side == 0 || side == 1 ? u : (side == 2 ? -hw : hw);
        py = side == 0 ? hw : (side == 1 ? -hw : u);
        if (clash_pos(px, py, radius, first, count, false) && t != EBC_GEN_MAX_TRIES - 1) continue;
        const double g = rs.uniform(-hw, hw);  // the goal lies on the opposite side
        gx = side == 0 || side == 1 ? g : (side == 2 ? hw : -hw);
        gy = side == 0 ? -hw : (side == 1 ? hw : g);
        break;
      }
    } else {  // square_crossing_old, :714-761
      const double sign = rs.next_double() < 0.5 ? 1.0 : -1.0;  // choice([1, -1], p=[0.5, 0.5])
      for (int t = 0; t < EBC_GEN_MAX_TRIES; ++t) {
        px = rs.next_double() * c.square_width * 0.5 * sign;
        py = c.square_width * 0.5;
        if (rs.next_double() > 0.5) {
          const double tmp = px;
          px = py;
          py = tmp;
        }
        if (clash_pos(px, py, radius, first, count, false) && t != EBC_GEN_MAX_TRIES - 1) continue;
        const long v = rs.randint(0, 3);  // [(-1, 1), (1, -1), (-1, -1)][randint(3)]
        gx = px * (v == 1 ? 1.0 : -1.0);
        gy = py * (v == 0 ? 1.0 : -1.0);
        bool clash = false;
        if (t != EBC_GEN_MAX_TRIES - 1)
          clash = gen_hyp(gx - rgx, gy - rgy) < radius + c.robot_radius + c.discomfort_dist;
        if (!clash) break;
      }
    }
    o.px[slot] = px; o.py[slot] = py; o.gx[slot] = gx; o.gy[slot] = gy;
    o.vx[slot] = 0; o.vy[slot] = 0;
    o.radius[slot] = radius; o.v_pref[slot] = v_pref;
    o.type[slot] = (uint8_t)kind;
  }

  EBC_HD void clear_cell(long x, long y) const { o.grid[x * 2 + (y >> 6)] |= 1ull << (y & 63); }

  // place_obstacles_on_map, :888-922 (bit y of row x set <=> map[x, y] == 0)
  EBC_HD void rasterize(long lx, long ly, long dx, long dy) const {
    if (!o.grid) return;
    if (dx / 2.0 < lx && lx < G - dx / 2.0 && dy / 2.0 < ly && ly < G - dy / 2.0) {
      const long sx = py_round(lx - dx / 2.0), sy = py_round(ly - dy / 2.0);
      for (long x = sx; x < sx + dx && x < G; ++x)
        for (long y = sy; y < sy + dy && y < G; ++y)
          if (x >= 0 && y >= 0) clear_cell(x, y);
    } else {
      for (long ix = 0; ix < dx; ++ix)
        for (long iy = 0; iy < dy; ++iy) {
          const long x = py_round(lx + (ix - dx / 2.0)), y = py_round(ly + (iy - dy / 2.0));
          if (0 < x && x < G && 0 < y && y < G) clear_cell(x, y);
        }
    }
  }

  // one obstacle's observation rows, :380-422; returns false when the handle has too few static rows
  EBC_HD bool rows_of(long d0, long d1, double xm, double ym, double hx, double hy, int &ns) const {
    const double SQRT2 = 1.4142135623730951;
    // vertices 0..3: (xm + hx, ym + hy), (xm - hx, ym + hy), (xm - hx, ym - hy), (xm + hx, ym - hy)
    const double v0x = xm + hx, v0y = ym + hy, v1x = xm - hx, v2x = xm - hx, v2y = ym - hy;
    if (d0 == d1) {
      const double px = (v0x + v2x) / 2.0, py = (v0y + v2y) / 2.0;
      if (ns >= S) return false;
      o.spx[ns] = px; o.spy[ns] = py; o.sradius[ns] = (v0x - px) * SQRT2;
      ++ns;
    } else if (d0 > d1) {
      const double py = (v0y + v2y) / 2.0, rad = (v0y - py) * SQRT2;
      for (double px = v1x + rad; px < v0x; px = px + 2 * rad) {
        if (ns >= S) return false;
        o.spx[ns] = px; o.spy[ns] = py; o.sradius[ns] = rad;
        ++ns;
      }
    } else {
      const double px = (v0x + v2x) / 2.0, rad = (v0x - px) * SQRT2;
      for (double py = v2y + rad; py < v0y; py = py + 2 * rad) {
        if (ns >= S) return false;
        o.spx[ns] = px; o.spy[ns] = py; o.sradius[ns] = rad;
        ++ns;
      }
    }
    return true;
  }

  // returns 0, or EBC_GEN_STATIC_OVERFLOW
  EBC_HD int run() {
    int slot = 0;
    for (int kind = 0; kind < 3; ++kind) {
      const int first = slot;
      for (int q = 0; q < c.count[kind]; ++q, ++slot) place(slot, kind, first, q);
    }
    *o.n_humans = slot;
    for (; slot < N; ++slot) {
      o.px[slot] = o.py[slot] = o.gx[slot] = o.gy[slot] = o.vx[slot] = o.vy[slot] = 0;
      o.radius[slot] = o.v_pref[slot] = 0;
      o.type[slot] = 0;
    }
    o.robot[0] = rpx; o.robot[1] = rpy; o.robot[2] = 0; o.robot[3] = 0; o.robot[4] = c.robot_radius;
    o.robot[5] = rgx; o.robot[6] = rgy; o.robot[7] = c.robot_v_pref; o.robot[8] = 3.141592653589793 / 2;
    // ---- static map, :109-328
    if (o.grid)
      for (int q = 0; q < G * 2; ++q) o.grid[q] = 0;
    const double res = c.map_resolution, reach = c.robot_radius + c.discomfort_dist;
    const long lo = (long)(-G / 2.0), hi = (long)(G / 2.0);  // randint(-max_loc / 2.0, max_loc / 2.0): int() truncates
    int ns = 0, status = 0;
    for (int q = 0; q < c.num_circles; ++q) {
      long lx = 0, ly = 0;
      double rad = 0, xm = 0, ym = 0;
      for (int t = 0; t < EBC_GEN_MAX_TRIES; ++t) {
        lx = rs.randint(lo, hi);
        ly = rs.randint(lo, hi);
        rad = (rs.next_double() + 0.5) * 0.7;
        xm = lx * res;
        ym = ly * res;
        if (!(gen_hyp(xm - rpx, ym - rpy) < rad + reach || gen_hyp(xm - rgx, ym - rgy) < rad + reach)) break;
      }
      const long d = py_round(2 * rad / res);
      rasterize(py_round(lx + G / 2.0), py_round(ly + G / 2.0), d, d);
      if (!rows_of(d, d, xm, ym, rad, rad, ns)) status = EBC_GEN_STATIC_OVERFLOW;
    }
    for (int q = 0; q < c.num_walls; ++q) {
      long lx = 0, ly = 0, xd = 1, yd = 1;
      double xm = 0, ym = 0;
      for (int t = 0; t < EBC_GEN_MAX_TRIES; ++t) {
        lx = rs.randint(lo, hi);
        ly = rs.randint(lo, hi);
        if (rs.next_double() > 0.5) {
          xd = rs.randint(c.min_wall_length, c.max_wall_length + 1);
          yd = 1;
        } else {
          yd = rs.randint(c.min_wall_length, c.max_wall_length + 1);
          xd = 1;
        }
        xm = lx * res;
        ym = ly * res;
        const bool near_start = fabs(xm - rpx) < xd / 2.0 + reach && fabs(ym - rpy) < yd / 2.0 + reach;
        const bool near_goal = fabs(xm - rgx) < xd / 2.0 + reach && fabs(ym - rgy) < yd / 2.0 + reach;
        if (!(near_start || near_goal)) break;
      }
      const long d0 = py_round(xd / res), d1 = py_round(yd / res);
      rasterize(py_round(lx + G / 2.0), py_round(ly + G / 2.0), d0, d1);
      if (!rows_of(d0, d1, xm, ym, xd / 2.0, yd / 2.0, ns)) status = EBC_GEN_STATIC_OVERFLOW;
    }
    if (o.n_static) *o.n_static = ns;
    for (int q = ns; q < S; ++q) o.spx[q] = o.spy[q] = o.sradius[q] = 0;
    return status;
  }
};

// One scene: seed the stream, run the generator into row `o`.
EBC_HD int generate_scene_row(const EbcSceneGen &c, uint32_t seed, uint32_t *mt, size_t mt_stride, const SceneRow &o,
                              int N, int S, int G) {
  Mt19937 rs{mt, mt_stride, 624};
  rs.seed(seed);
  SceneGenCore g{c, rs, o, N, S, G, 0.0, -c.circle_radius, 0.0, c.circle_radius};
  return g.run();
}

}  // namespace ebc
