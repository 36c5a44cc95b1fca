"""Host-side scene construction: what `env.reset()` hands to the kernels.

Mirrors the *outputs* of the reference's SceneGenerator
(simulator/scene/scene_generator.py): initial human rows in type-major order,
the occupancy grid, the static-obstacle observation rows.  Random scenes draw
from numpy's legacy MT19937 stream in the reference's order, so seed s here is
scene s there (scene_generator.py:356-370).  Stays on the host by design
(north_star: "Python host code keeps the config/scene loaders").
"""
import json
import math
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

from . import _abi

MAX_TRIES = 100000  # scene_generator.py:11


@dataclass
class AgentSpec:
    """[adults] / [bicycles] / [children] / [robot] section (agents/agent.py:16-35)."""
    radius: Optional[float] = None
    v_pref: Optional[float] = None
    v_pref_min: Optional[float] = None
    v_pref_max: Optional[float] = None
    radius_min: Optional[float] = None
    radius_max: Optional[float] = None
    visible: bool = True
    policy: str = "orca"

    @classmethod
    def from_config(cls, cfg, section):
        g = lambda k: cfg.getfloat(section, k, fallback=None)  # noqa: E731
        return cls(radius=g("radius"), v_pref=g("v_pref"), v_pref_min=g("v_pref_min"),
                   v_pref_max=g("v_pref_max"), radius_min=g("radius_min"),
                   radius_max=g("radius_max"),
                   visible=cfg.getboolean(section, "visible", fallback=True),
                   policy=cfg.get(section, "policy", fallback="orca"))


@dataclass
class SceneConfig:
    """Keys SceneGenerator.__init__ reads (scene_generator.py:20-72)."""
    adult_num: int = 5
    bicycle_num: int = 0
    children_num: int = 0
    train_val_sim_adult: str = "circle_crossing"
    test_sim_adult: str = "circle_crossing"
    train_val_sim_bicycle: Optional[str] = None
    test_sim_bicycle: Optional[str] = None
    train_val_sim_children: Optional[str] = None
    test_sim_children: Optional[str] = None
    square_width: float = 9.0
    circle_radius: float = 3.0
    randomize_attributes: bool = False
    map_resolution: float = 0.1
    map_size_m: float = 9.0
    min_wall_length: int = 2
    max_wall_length: int = 4
    num_circles: int = 0
    num_walls: int = 0
    discomfort_dist: float = 0.1
    val_size: int = 100
    test_size: int = 500
    adults: AgentSpec = field(default_factory=AgentSpec)
    bicycles: AgentSpec = field(default_factory=AgentSpec)
    children: AgentSpec = field(default_factory=AgentSpec)
    robot: AgentSpec = field(default_factory=AgentSpec)

    @classmethod
    def from_config(cls, cfg):
        s = cls()
        s.adult_num = cfg.getint("sim", "adult_num")
        s.bicycle_num = cfg.getint("sim", "bicycle_num", fallback=0)
        s.children_num = cfg.getint("sim", "children_num", fallback=0)
        if cfg.get("sim", "bicycle_type", fallback=None) == "rectangle":
            raise NotImplementedError("bicycle_type = rectangle is set by no shipped config")
        s.train_val_sim_adult = cfg.get("sim", "train_val_sim_adult")
        s.test_sim_adult = cfg.get("sim", "test_sim_adult")
        s.train_val_sim_bicycle = cfg.get("sim", "train_val_sim_bicycle", fallback=None)
        s.test_sim_bicycle = cfg.get("sim", "test_sim_bicycle", fallback=None)
        s.train_val_sim_children = cfg.get("sim", "train_val_sim_children", fallback=None)
        s.test_sim_children = cfg.get("sim", "test_sim_children", fallback=None)
        s.square_width = cfg.getfloat("sim", "square_width")
        s.circle_radius = cfg.getfloat("sim", "circle_radius")
        s.randomize_attributes = cfg.getboolean("env", "randomize_attributes")
        s.map_resolution = cfg.getfloat("map", "map_resolution")
        s.map_size_m = cfg.getfloat("map", "map_size_m")
        s.min_wall_length = cfg.getint("map", "min_wall_length", fallback=2)
        s.max_wall_length = cfg.getint("map", "max_wall_length", fallback=4)
        s.num_circles = cfg.getint("map", "num_circles")
        s.num_walls = cfg.getint("map", "num_walls")
        s.discomfort_dist = cfg.getfloat("reward", "discomfort_dist")
        s.val_size = cfg.getint("env", "val_size")
        s.test_size = cfg.getint("env", "test_size")
        for name in ("adults", "bicycles", "children", "robot"):
            if cfg.has_section(name):
                setattr(s, name, AgentSpec.from_config(cfg, name))
        return s


def max_static_rows(cfg: "SceneConfig") -> int:
    """Upper bound on the observation rows of a generated map (static_rows): one per circle; a wall of length L
    (width 1) is covered by circles of radius sqrt(2)/2 every sqrt(2) along it, as many as `static_rows` steps
    over its length (2 for L = 3)."""
    def wall_rows(length):
        rad = (0.5 - 0.0) * np.sqrt(2)
        x, n = -length / 2.0 + rad, 0
        while x < length / 2.0:
            n += 1
            x = x + 2 * rad
        return max(n, 1)
    longest = max((wall_rows(float(L)) for L in range(int(cfg.min_wall_length), int(cfg.max_wall_length) + 1)), default=1)
    return (cfg.num_circles or 0) + (cfg.num_walls or 0) * longest


_RULES = {"circle_crossing": _abi.RULE_CIRCLE_CROSSING, "square_crossing": _abi.RULE_SQUARE_CROSSING,
          "square_crossing_old": _abi.RULE_SQUARE_CROSSING_OLD}


def gen_struct(cfg: SceneConfig, phase: str = "test", multiagent_training: bool = True) -> "_abi.EbcSceneGen":
    """The EbcSceneGen of include/ebcsim.h for one phase: what `generate_scene(cfg, seed, phase)` would place,
    for the device generator (ebc_generate_*).  Rules the reference cannot run raise here, as in `_Gen.group`."""
    import ctypes as C
    g = _abi.EbcSceneGen()
    g.struct_size = C.sizeof(g)
    test = phase == "test"
    many = test or multiagent_training
    rules = ((cfg.test_sim_adult, cfg.test_sim_bicycle, cfg.test_sim_children) if test else
             (cfg.train_val_sim_adult, cfg.train_val_sim_bicycle, cfg.train_val_sim_children))
    counts = (cfg.adult_num, cfg.bicycle_num, cfg.children_num)
    g.randomize_attributes = int(bool(cfg.randomize_attributes))
    for t, (spec, rule, count) in enumerate(zip((cfg.adults, cfg.bicycles, cfg.children), rules, counts)):
        n = count if many else 1
        g.count[t] = n
        if n:
            if rule not in _RULES or (rule == "circle_crossing" and t == _abi.CHILD) or (
                    rule == "square_crossing_old" and t != _abi.BICYCLE):
                raise ValueError("unsupported crossing rule %r for type %d" % (rule, t))
            g.rule[t] = _RULES[rule]
        for key in ("radius", "v_pref", "radius_min", "radius_max", "v_pref_min", "v_pref_max"):
            v = getattr(spec, key)
            if v is None and n and (cfg.randomize_attributes) == key.endswith(("_min", "_max")):
                raise ValueError("[%s] %s missing" % (("adults", "bicycles", "children")[t], key))
            getattr(g, key)[t] = 0.0 if v is None else float(v)
    g.square_width, g.circle_radius = cfg.square_width, cfg.circle_radius
    g.discomfort_dist = cfg.discomfort_dist
    g.robot_radius, g.robot_v_pref = cfg.robot.radius, cfg.robot.v_pref
    g.map_resolution, g.map_size_m = cfg.map_resolution, cfg.map_size_m
    g.min_wall_length, g.max_wall_length = cfg.min_wall_length, cfg.max_wall_length
    g.num_circles, g.num_walls = cfg.num_circles or 0, cfg.num_walls or 0
    return g


@dataclass
class Human:
    px: float
    py: float
    gx: float
    gy: float
    radius: float
    v_pref: float
    type: int
    vx: float = 0.0
    vy: float = 0.0
    theta: float = 0.0


@dataclass
class Scene:
    """One env's reset state."""
    humans: List[Human]                  # adults, then bicycles, then children
    robot: np.ndarray                    # [9] FullState order
    grid: np.ndarray                     # [G, G] float64 ones / zeros (scene.map)
    static_rows: np.ndarray              # [S, 3] px, py, radius
    obstacle_vertices: list
    obstacles: list                      # [(loc_x, loc_y, (dx, dy))]
    num_circles: int = 0
    num_walls: int = 0


def _hyp(x, y):
    return np.linalg.norm((x, y))  # same BLAS path as the reference's rejection tests


class _Gen:
    """One generate_random_scene() (scene_generator.py:330-378) on a private RandomState."""

    def __init__(self, cfg: SceneConfig, seed: int):
        self.c = cfg
        self.rs = np.random.RandomState(seed)
        R = cfg.circle_radius
        # env.reset: robot.set(0, -R, 0, R, 0, 0, pi/2)  (simulator/env.py:159-161)
        self.robot = np.array([0.0, -R, 0.0, 0.0, cfg.robot.radius, 0.0, R, cfg.robot.v_pref,
                               np.pi / 2])

    # robot seen as an "agent" by the rejection tests
    def _robot_h(self):
        r = self.robot
        return Human(r[0], r[1], r[5], r[6], r[4], r[7], _abi.ROBOT)

    def _new(self, spec: AgentSpec, kind: int, randomize: bool):
        h = Human(None, None, None, None, spec.radius, spec.v_pref, kind)
        if randomize:  # Agent.sample_random_attributes, agent.py:48-56
            h.v_pref = self.rs.uniform(spec.v_pref_min, spec.v_pref_max)
            h.radius = self.rs.uniform(spec.radius_min, spec.radius_max)
        return h

    def circle_crossing(self, spec, kind, peers):  # scene_generator.py:593-648
        c = self.c
        h = self._new(spec, kind, c.randomize_attributes)
        px = py = None
        for _ in range(MAX_TRIES):
            angle = self.rs.random_sample() * np.pi * 2
            px = c.circle_radius * np.cos(angle) + 0
            py = c.circle_radius * np.sin(angle) + 0
            clash = False
            for o in [self._robot_h()] + peers:
                lim = h.radius + o.radius + c.discomfort_dist
                if _hyp(px - o.px, py - o.py) < lim or _hyp(px - o.gx, py - o.gy) < lim:
                    clash = True
                    break
            if not clash:
                break
        h.px, h.py, h.gx, h.gy = px, py, -px, -py
        return h

    def _start(self):  # scene_generator.py:650-670
        hw = self.c.square_width / 2
        side = self.rs.choice(["top", "bottom", "left", "right"])
        if side == "top":
            return (self.rs.uniform(-hw, hw), hw), "bottom"
        if side == "bottom":
            return (self.rs.uniform(-hw, hw), -hw), "top"
        if side == "left":
            return (-hw, self.rs.uniform(-hw, hw)), "right"
        return (hw, self.rs.uniform(-hw, hw)), "left"

    def square_crossing(self, spec, kind, peers):  # scene_generator.py:672-712
        c = self.c
        h = self._new(spec, kind, c.randomize_attributes)
        hw = c.square_width / 2
        for index in range(MAX_TRIES):
            (px, py), goal_side = self._start()
            clash = False
            for o in [self._robot_h()] + peers:
                if _hyp(px - o.px, py - o.py) < h.radius + o.radius + c.discomfort_dist:
                    clash = True
                    break
            if clash and index != MAX_TRIES - 1:
                continue
            if goal_side == "top":
                gx, gy = self.rs.uniform(-hw, hw), hw
            elif goal_side == "bottom":
                gx, gy = self.rs.uniform(-hw, hw), -hw
            elif goal_side == "left":
                gx, gy = -hw, self.rs.uniform(-hw, hw)
            else:
                gx, gy = hw, self.rs.uniform(-hw, hw)
            break
        h.px, h.py, h.gx, h.gy = px, py, gx, gy
        return h

    def square_crossing_old(self, spec, kind, peers):  # scene_generator.py:714-761
        c = self.c
        h = self._new(spec, kind, c.randomize_attributes)
        sign = self.rs.choice([1, -1], p=[0.5, 0.5])
        robot = self._robot_h()
        for index in range(MAX_TRIES):
            px = self.rs.random_sample() * c.square_width * 0.5 * sign
            py = c.square_width * 0.5
            if self.rs.random_sample() > 0.5:
                px, py = py, px
            clash = False
            for o in [robot] + peers:
                if _hyp(px - o.px, py - o.py) < h.radius + o.radius + c.discomfort_dist:
                    clash = True
                    break
            if clash and index != MAX_TRIES - 1:
                continue
            variant = [(-1, 1), (1, -1), (-1, -1)][self.rs.randint(3)]
            gx, gy = px * variant[0], py * variant[1]
            clash = False
            if index != MAX_TRIES - 1:
                if _hyp(gx - robot.gx, gy - robot.gy) < h.radius + robot.radius + c.discomfort_dist:
                    clash = True
            if not clash:
                break
        h.px, h.py, h.gx, h.gy = px, py, gx, gy
        return h

    def group(self, count, rule, spec, kind):
        out = []
        for _ in range(count):
            if rule == "circle_crossing":
                if kind == _abi.CHILD:
                    # the reference raises NameError here (scene_generator.py:449-457)
                    raise ValueError("circle_crossing is not defined for children")
                out.append(self.circle_crossing(spec, kind, out))
            elif rule == "square_crossing":
                out.append(self.square_crossing(spec, kind, out))
            elif rule == "square_crossing_old" and kind == _abi.BICYCLE:
                out.append(self.square_crossing_old(spec, kind, out))
            else:
                raise ValueError("unsupported crossing rule %r for type %d" % (rule, kind))
        return out

    # ---- static map: scene_generator.py:109-328 ----
    def static_map(self):
        c = self.c
        G = int(round(c.map_size_m / c.map_resolution))
        max_loc = int(round(G))
        obstacles, vertices = [], []
        reach = c.robot.radius + c.discomfort_dist
        r = self.robot
        for _ in range(c.num_circles or 0):
            for _ in range(MAX_TRIES):
                lx = self.rs.randint(-max_loc / 2.0, max_loc / 2.0)
                ly = self.rs.randint(-max_loc / 2.0, max_loc / 2.0)
                rad = (self.rs.random_sample() + 0.5) * 0.7
                xm, ym = lx * c.map_resolution, ly * c.map_resolution
                if not (_hyp(xm - r[0], ym - r[1]) < rad + reach
                        or _hyp(xm - r[5], ym - r[6]) < rad + reach):
                    break
            d = int(round(2 * rad / c.map_resolution))
            obstacles.append((int(round(lx + G / 2.0)), int(round(ly + G / 2.0)), (d, d)))
            vertices.append([(xm + rad, ym + rad), (xm - rad, ym + rad),
                             (xm - rad, ym - rad), (xm + rad, ym - rad)])
        for _ in range(c.num_walls or 0):
            for _ in range(MAX_TRIES):
                lx = self.rs.randint(-max_loc / 2.0, max_loc / 2.0)
                ly = self.rs.randint(-max_loc / 2.0, max_loc / 2.0)
                if self.rs.random_sample() > 0.5:
                    xd, yd = self.rs.randint(c.min_wall_length, c.max_wall_length + 1), 1
                else:
                    yd, xd = self.rs.randint(c.min_wall_length, c.max_wall_length + 1), 1
                xm, ym = lx * c.map_resolution, ly * c.map_resolution
                near_start = abs(xm - r[0]) < xd / 2.0 + reach and abs(ym - r[1]) < yd / 2.0 + reach
                near_goal = abs(xm - r[5]) < xd / 2.0 + reach and abs(ym - r[6]) < yd / 2.0 + reach
                if not (near_start or near_goal):
                    break
            dim = (int(round(xd / c.map_resolution)), int(round(yd / c.map_resolution)))
            obstacles.append((int(round(lx + G / 2.0)), int(round(ly + G / 2.0)), dim))
            vertices.append([(xm + xd / 2.0, ym + yd / 2.0), (xm - xd / 2.0, ym + yd / 2.0),
                             (xm - xd / 2.0, ym - yd / 2.0), (xm + xd / 2.0, ym - yd / 2.0)])
        return G, obstacles, vertices


def rasterize(obstacles, G):
    """place_obstacles_on_map, scene_generator.py:888-922 (edge striping included)."""
    grid = np.ones((G, G))
    for lx, ly, (dx, dy) in obstacles:
        if dx / 2.0 < lx < G - dx / 2.0 and dy / 2.0 < ly < G - dy / 2.0:
            sx = int(round(lx - dx / 2.0))
            sy = int(round(ly - dy / 2.0))
            grid[sx:sx + dx, sy:sy + dy] = 0
        else:
            for ix in range(dx):
                for iy in range(dy):
                    x = int(round(lx + (ix - dx / 2.0)))
                    y = int(round(ly + (iy - dy / 2.0)))
                    if 0 < x < G and 0 < y < G:
                        grid[x, y] = 0
    return grid


def static_rows(obstacles, vertices):
    """create_observation_from_static_obstacles, scene_generator.py:380-422."""
    rows = []
    for (_, _, dim), v in zip(obstacles, vertices):
        if dim[0] == dim[1]:
            px = (v[0][0] + v[2][0]) / 2.0
            py = (v[0][1] + v[2][1]) / 2.0
            rows.append((px, py, (v[0][0] - px) * np.sqrt(2)))
        elif dim[0] > dim[1]:
            py = (v[0][1] + v[2][1]) / 2.0
            rad = (v[0][1] - py) * np.sqrt(2)
            px = v[1][0] + rad
            while px < v[0][0]:
                rows.append((px, py, rad))
                px = px + 2 * rad
        else:
            px = (v[0][0] + v[2][0]) / 2.0
            rad = (v[0][0] - px) * np.sqrt(2)
            py = v[2][1] + rad
            while py < v[0][1]:
                rows.append((px, py, rad))
                py = py + 2 * rad
    return np.array(rows, dtype=np.float64).reshape(-1, 3)


COUNTER_OFFSET = {"train": 2000, "val": 0, "test": 1000}  # simulator/env.py:153-158


def generate_scene(cfg: SceneConfig, seed: int, phase: str = "test",
                   multiagent_training: bool = True) -> Scene:
    """generate_random_scene for one seed (scene_generator.py:330-378)."""
    g = _Gen(cfg, seed)
    test = phase == "test"
    many = test or multiagent_training
    rule_a = cfg.test_sim_adult if test else cfg.train_val_sim_adult
    rule_b = cfg.test_sim_bicycle if test else cfg.train_val_sim_bicycle
    rule_c = cfg.test_sim_children if test else cfg.train_val_sim_children
    humans = g.group(cfg.adult_num if many else 1, rule_a, cfg.adults, _abi.ADULT)
    humans += g.group(cfg.bicycle_num if many else 1, rule_b, cfg.bicycles, _abi.BICYCLE)
    humans += g.group(cfg.children_num if many else 1, rule_c, cfg.children, _abi.CHILD)
    G, obstacles, vertices = g.static_map()
    return Scene(humans, g.robot, rasterize(obstacles, G), static_rows(obstacles, vertices),
                 vertices, obstacles, cfg.num_circles or 0, cfg.num_walls or 0)


def load_scene(cfg: SceneConfig, path: str) -> Scene:
    """SceneGenerator.load_scene, scene_generator.py:807-863."""
    with open(path) as f:
        js = json.load(f)
    humans = []
    for key, spec, kind in (("adults", cfg.adults, _abi.ADULT),
                            ("bicycles", cfg.bicycles, _abi.BICYCLE),
                            ("children", cfg.children, _abi.CHILD)):
        for st in js.get(key, []):
            t = st.get("agent_type")
            humans.append(Human(st["pos"][0], st["pos"][1], st["goal"][0], st["goal"][1],
                                st["radius"], st["v_pref"], kind if t is None else int(t),
                                st["vel"][0], st["vel"][1], st["theta"]))
    m = js["map"]
    vertices = m["obstacle_vertices"]
    if len(vertices) != m["num_circles"] + m["num_walls"]:
        raise AssertionError("Error: length of obstacle_vertices != num_circles + num_walls")
    obstacles = [(o["location"][0], o["location"][1], tuple(o["dim"])) for o in m["obstacles"]]
    G = int(round(cfg.map_size_m / cfg.map_resolution))
    R = cfg.circle_radius
    robot = np.array([0.0, -R, 0.0, 0.0, cfg.robot.radius, 0.0, R, cfg.robot.v_pref, np.pi / 2])
    return Scene(humans, robot, rasterize(obstacles, G), static_rows(obstacles, vertices),
                 vertices, obstacles, m["num_circles"], m["num_walls"])


def save_scene(scene: Scene, path: str):
    """SceneGenerator.save_scene, scene_generator.py:865-886."""
    out = {"adults": [], "bicycles": [], "children": []}
    key = {_abi.ADULT: "adults", _abi.BICYCLE: "bicycles", _abi.CHILD: "children"}
    for h in scene.humans:
        out[key[h.type]].append({"pos": (h.px, h.py), "vel": (h.vx, h.vy), "radius": h.radius,
                                 "goal": (h.gx, h.gy), "v_pref": h.v_pref, "theta": h.theta,
                                 "agent_type": int(h.type)})
    out["map"] = {"num_circles": scene.num_circles, "num_walls": scene.num_walls,
                  "obstacle_vertices": scene.obstacle_vertices,
                  "obstacles": [{"location": (o[0], o[1]), "dim": o[2]} for o in scene.obstacles]}
    with open(path, "w") as f:
        json.dump(out, f, indent=4, sort_keys=True)


def pack_grid(grid: np.ndarray) -> np.ndarray:
    """[G, G] ones/zeros -> [G, 2] uint64, bit y of row x set <=> map[x, y] == 0."""
    G = grid.shape[0]
    if G > 128:
        raise ValueError("occupancy grids wider than 128 cells are not supported")
    occ = np.zeros((G, 128), dtype=np.uint8)
    occ[:, :G] = (grid == 0)
    bits = np.packbits(occ, axis=1, bitorder="little")  # [G, 16] bytes
    return bits.view(np.uint64).reshape(G, 2)


@dataclass
class SceneBatch:
    """Padded struct-of-arrays form of n scenes (the EbcScene of include/ebcsim.h)."""
    n: int
    N: int
    S: int
    n_humans: np.ndarray
    px: np.ndarray
    py: np.ndarray
    vx: np.ndarray
    vy: np.ndarray
    gx: np.ndarray
    gy: np.ndarray
    radius: np.ndarray
    v_pref: np.ndarray
    type: np.ndarray
    n_static: np.ndarray
    spx: np.ndarray
    spy: np.ndarray
    sradius: np.ndarray
    grid: Optional[np.ndarray]
    robot: np.ndarray

    @classmethod
    def from_scenes(cls, scenes: List[Scene], max_humans=None, max_static=None):
        n = len(scenes)
        N = max_humans if max_humans is not None else max(len(s.humans) for s in scenes)
        S = max_static if max_static is not None else max(len(s.static_rows) for s in scenes)
        f = lambda *sh: np.zeros(sh, dtype=np.float64)  # noqa: E731
        b = cls(n, N, S, np.zeros(n, np.int32), f(n, N), f(n, N), f(n, N), f(n, N), f(n, N),
                f(n, N), f(n, N), f(n, N), np.zeros((n, N), np.uint8), np.zeros(n, np.int32),
                f(n, max(S, 1)), f(n, max(S, 1)), f(n, max(S, 1)), None, f(n, 9))
        any_obstacle = any((s.grid == 0).any() for s in scenes)
        if any_obstacle:
            G = scenes[0].grid.shape[0]
            b.grid = np.zeros((n, G, 2), dtype=np.uint64)
        for e, s in enumerate(scenes):
            k = len(s.humans)
            if k > N or len(s.static_rows) > S:
                raise ValueError("scene %d exceeds max_humans/max_static" % e)
            b.n_humans[e] = k
            for i, h in enumerate(s.humans):
                b.px[e, i], b.py[e, i], b.vx[e, i], b.vy[e, i] = h.px, h.py, h.vx, h.vy
                b.gx[e, i], b.gy[e, i], b.radius[e, i], b.v_pref[e, i] = h.gx, h.gy, h.radius, h.v_pref
                b.type[e, i] = h.type
            m = len(s.static_rows)
            b.n_static[e] = m
            if m:
                b.spx[e, :m], b.spy[e, :m], b.sradius[e, :m] = s.static_rows.T
            if b.grid is not None:
                b.grid[e] = pack_grid(s.grid)
            b.robot[e] = s.robot
        return b
