"""SceneGenerator.generate_random_scene on the device (scene_generator.py:330-378, 593-761, 109-328,
380-422, 888-922).

CPU: the generator source the kernel compiles (eb-cadrl_amd/csrc/ebc_scene_gen.h) built for the host with g++
(tests/native/scene_gen_host.cc) against (a) the reference's own SceneGenerator outputs (tests/golden/scenes.npz:
31 (config, seed) pairs) and (b) ebcsim/scene.py — numpy's RandomState, the reference's draw order — over
every crossing rule with and without random attributes.  Everything is bit for bit except the positions
circle_crossing derives from cos / sin: numpy evaluates those with a SIMD routine whose last bit depends on the
CPU, the generator with a correctly rounded one; they may differ by 1 ulp (about 1 position in 1000 here) and the
tests say so.
GPU (-m gpu): the kernel against the same host build (identical source: every bit, cos / sin included), against
the goldens, and its two consumers (generate_reset, generate_pool + auto-reset against the oracle)."""
import configparser
import copy
import ctypes as C
import json
import os
import subprocess

import numpy as np
import pytest

from ebcsim import _abi, config as ebc_config, scene as ebc_scene
from helpers import load, params_of

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ARRAYS = ("n_humans", "px", "py", "vx", "vy", "gx", "gy", "radius", "v_pref", "type", "n_static", "spx", "spy", "sradius",
          "grid", "robot")


@pytest.fixture(scope="module")
def host_gen(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("scene_gen") / "libscene_gen_host.so")
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-Wall", "-Werror",
                    os.path.join(ROOT, "tests", "native", "scene_gen_host.cc"), "-o", so], check=True, timeout=300)
    lib = C.CDLL(so)

    def run(gen, seeds, N, S, G):
        n = len(seeds)
        f = lambda *sh: np.zeros(sh)  # noqa: E731
        S1 = max(S, 1)
        b = ebc_scene.SceneBatch(n, N, S, np.zeros(n, np.int32), f(n, N), f(n, N), f(n, N), f(n, N), f(n, N), f(n, N),
                                 f(n, N), f(n, N), np.zeros((n, N), np.uint8), np.zeros(n, np.int32), f(n, S1), f(n, S1),
                                 f(n, S1), np.zeros((n, G, 2), np.uint64), f(n, 9))
        sd = np.asarray(seeds, np.uint32)
        rc = lib.scene_gen_host(C.byref(gen), C.c_void_p(sd.ctypes.data), n, N, S, G,
                                *[C.c_void_p(getattr(b, k).ctypes.data) for k in ARRAYS])
        return rc, b

    run.lib = lib
    return run


def _golden_cases():
    z = load("scenes")
    for k in range(int(z["n"])):
        meta = json.loads(str(z["meta_%d" % k]))
        cfg = configparser.RawConfigParser()
        cfg.read_string(meta["config_text"])
        yield k, z, meta, ebc_scene.SceneConfig.from_config(cfg)


def _assert_golden(k, z, b, N_ref):
    n = int(b.n_humans[0])
    assert n == len(z["px_%d" % k])
    for key in ("px", "py", "gx", "gy", "radius", "v_pref", "type"):
        np.testing.assert_array_equal(getattr(b, key)[0, :n], z["%s_%d" % (key, k)], err_msg="%s of case %d" % (key, k))
        assert not getattr(b, key)[0, n:].any()
    st = z["static_%d" % k].reshape(-1, 3)
    m = int(b.n_static[0])
    assert m == len(st)
    np.testing.assert_array_equal(np.stack([b.spx[0, :m], b.spy[0, :m], b.sradius[0, :m]], 1).reshape(-1, 3), st)
    np.testing.assert_array_equal(b.robot[0], z["robot_%d" % k])
    grid = np.ones(z["grid_%d" % k].shape) - z["grid_%d" % k]  # fixture: 1 = occupied
    want = ebc_scene.pack_grid(grid)
    np.testing.assert_array_equal(b.grid[0] if b.grid is not None else np.zeros_like(want), want)


def test_host_build_reproduces_reference_scenes(host_gen):
    """The reference's SceneGenerator outputs for 31 (config, seed) pairs: humans, static rows, map, robot."""
    for k, z, meta, sc in _golden_cases():
        gen = ebc_scene.gen_struct(sc, "test")
        N, S = sum(gen.count), max(len(z["static_%d" % k].reshape(-1, 3)), 1)
        rc, b = host_gen(gen, [meta["seed"]], N + 1, S + 2, z["grid_%d" % k].shape[0])  # padded: the tail stays zero
        assert rc == 0
        _assert_golden(k, z, b, N)


def _variants():
    base = next(sc for k, z, meta, sc in _golden_cases() if sc.num_circles and sc.num_walls)
    for ra in ("circle_crossing", "square_crossing"):
        for rb in ("circle_crossing", "square_crossing", "square_crossing_old"):
            for rnd in (False, True):
                c = copy.deepcopy(base)
                c.test_sim_adult, c.test_sim_bicycle, c.test_sim_children = ra, rb, "square_crossing"
                c.adult_num, c.bicycle_num, c.children_num = 4, 3, 2
                c.randomize_attributes = rnd
                for sp in (c.adults, c.bicycles, c.children):
                    sp.radius_min, sp.radius_max, sp.v_pref_min, sp.v_pref_max = 0.2, 0.5, 0.5, 1.5
                    sp.radius = 0.3 if sp.radius is None else sp.radius
                    sp.v_pref = 1.0 if sp.v_pref is None else sp.v_pref
                c.num_circles, c.num_walls = 3, 4
                yield c


def _compare_batches(got, want, circle, tag):
    """Bit for bit; where circle_crossing placed the human, positions and goals within 1 ulp.  Returns the number
    of values that differ."""
    differ = 0
    for key in ARRAYS:
        a, g = getattr(want, key), getattr(got, key)
        if a is None:
            assert not g.any(), (tag, key)
            continue
        if key in ("px", "py", "gx", "gy") and circle is not None:
            exact = ~circle
            np.testing.assert_array_equal(g[exact], a[exact], err_msg="%s %s" % (tag, key))
            assert (np.abs(g[circle] - a[circle]) <= np.spacing(np.abs(a[circle]))).all(), (tag, key)
            differ += int((g[circle] != a[circle]).sum())
        else:
            np.testing.assert_array_equal(g, a, err_msg="%s %s" % (tag, key))
    return differ


def test_host_build_matches_numpy_generator_every_rule(host_gen):
    """12 rule / attribute variants x 60 seeds against ebcsim/scene.py (numpy RandomState in the reference's order):
    the draws, the rejection loops, the map and its rows."""
    total = differ = 0
    for c in _variants():
        gen = ebc_scene.gen_struct(c, "test")
        seeds = list(range(5000, 5060))
        want = ebc_scene.SceneBatch.from_scenes([ebc_scene.generate_scene(c, s, "test") for s in seeds])
        rc, got = host_gen(gen, seeds, want.N, want.S, want.grid.shape[1])
        assert rc == 0
        circle = np.zeros(want.px.shape, bool)
        if c.test_sim_adult == "circle_crossing":
            circle[:, :4] = True
        if c.test_sim_bicycle == "circle_crossing":
            circle[:, 4:7] = True
        differ += _compare_batches(got, want, circle, "%s/%s/%s" % (c.test_sim_adult, c.test_sim_bicycle, c.randomize_attributes))
        total += 4 * int(circle.sum())
    assert differ <= total // 100, (differ, total)  # 1-ulp disagreements with numpy's SIMD cos / sin are rare


def test_train_phase_and_single_agent_counts(host_gen):
    """Phase resolution of gen_struct: train rules, and one human per type without multiagent_training
    (scene_generator.py:343-355)."""
    k, z, meta, sc = next(x for x in _golden_cases() if x[3].bicycle_num)
    sc = copy.deepcopy(sc)
    sc.children_num, sc.train_val_sim_children = 0, None
    for phase, multi in (("train", True), ("val", True)):
        gen = ebc_scene.gen_struct(sc, phase, multi)
        want = ebc_scene.SceneBatch.from_scenes([ebc_scene.generate_scene(sc, s, phase, multi) for s in (2000, 2001, 7)])
        rc, got = host_gen(gen, [2000, 2001, 7], want.N, want.S, sc_grid(sc))
        assert rc == 0
        circle = np.ones(want.px.shape, bool)  # tolerant everywhere a circle rule may have acted
        _compare_batches(got, want, circle, phase)
    with pytest.raises(ValueError):  # one child on a rule the reference cannot run either
        ebc_scene.gen_struct(sc, "train", False)


def sc_grid(sc):
    return int(round(sc.map_size_m / sc.map_resolution))


def test_sincos_is_within_one_ulp_of_numpy(host_gen):
    rs = np.random.RandomState(3)
    ang = np.concatenate([rs.random_sample(20000) * np.pi * 2, [0.0, np.pi / 2, np.pi, 3 * np.pi / 2, 2 * np.pi,
                                                                 np.pi / 4, 7 * np.pi / 4, 1e-300, 5e-324]])
    c, s = np.zeros_like(ang), np.zeros_like(ang)
    host_gen.lib.sincos_host(C.c_void_p(ang.ctypes.data), len(ang), C.c_void_p(c.ctypes.data), C.c_void_p(s.ctypes.data))
    for got, want in ((c, np.cos(ang)), (s, np.sin(ang))):
        assert (np.abs(got - want) <= np.spacing(np.abs(want))).all()
        assert (got != want).mean() < 0.01


def test_max_static_rows_bounds_generated_maps():
    """scene.max_static_rows: what a handle needs as max_static to take any map of a config (walls of every length)."""
    for c in list(_variants())[:2] + [sc for _, _, _, sc in _golden_cases() if sc.num_walls][::5]:
        bound = ebc_scene.max_static_rows(c)
        most = max(len(ebc_scene.generate_scene(c, s, "test").static_rows) for s in range(7000, 7060))
        assert most <= bound <= most + (c.num_walls or 0), (bound, most)


def test_static_row_overflow_is_reported(host_gen):
    c = next(_variants())
    gen = ebc_scene.gen_struct(c, "test")
    rc, _ = host_gen(gen, [5000], 9, 2, sc_grid(c))  # 3 circles + 4 walls need more than 2 rows
    assert rc == 1  # EBC_GEN_STATIC_OVERFLOW


# ---------------------------------------------------------------------------------------------- GPU
def _params(sc_cfg_text):
    cfg = configparser.RawConfigParser()
    cfg.read_string(sc_cfg_text)
    return ebc_config.params_from_config(cfg)


@pytest.mark.gpu
def test_device_scenes_equal_host_build_and_goldens(host_gen):
    from ebcsim.batched import BatchedEnv
    for k, z, meta, sc in _golden_cases():
        gen = ebc_scene.gen_struct(sc, "test")
        N, S = sum(gen.count), max(len(z["static_%d" % k].reshape(-1, 3)), 1)
        env = BatchedEnv(_params(meta["config_text"]), 4, N + 1, S + 2)
        b = env.generate_scenes(gen, np.array([meta["seed"]], np.uint32), 1)
        _assert_golden(k, z, b, N)
    for c in _variants():  # the same source on both sides: every bit, cos / sin included
        gen = ebc_scene.gen_struct(c, "test")
        params = params_of(load("traj_n10_walls_t17_orcasub"))
        env = BatchedEnv(params, 8, 10, 16)
        seeds = np.arange(9000, 9300, dtype=np.uint32)
        got = env.generate_scenes(gen, seeds, len(seeds))
        rc, want = host_gen(gen, seeds, 10, 16, env.G)
        assert rc == 0
        _compare_batches(got, want, None, "device vs host build")
        by_offset = env.generate_scenes(gen, 9000, len(seeds))  # seed0 + i
        _compare_batches(by_offset, want, None, "seed0 form")


@pytest.mark.gpu
def test_generate_reset_and_pool_against_oracle(host_gen):
    """env.reset from device scenes = env.reset from the same scenes uploaded; the generated pool walks like an
    uploaded one (auto-reset, 200 steps, every output against the oracle)."""
    from ebcsim.batched import BatchedEnv
    from oracle import oracle
    from test_gpu_parity import _compare_step
    c = [v for v in _variants() if v.randomize_attributes][1]
    c.adult_num, c.bicycle_num, c.children_num = 3, 2, 1  # fewer humans than slots: ragged rows
    gen = ebc_scene.gen_struct(c, "test")
    params = params_of(load("traj_n10_walls_t17_orcasub"))
    params.time_limit = 5
    E, N, S = 48, 8, 16
    g = BatchedEnv(params, E, N, S)
    o = oracle.OracleEnv(params, E, N, S)
    first = g.generate_scenes(gen, 1000, E)
    pool = g.generate_scenes(gen, np.arange(50000, 50000 + 2 * E, dtype=np.uint32), 2 * E)
    o.reset(first)
    o.set_scene_pool(pool)
    g.generate_reset(gen, 1000)
    g.generate_pool(gen, np.arange(50000, 50000 + 2 * E, dtype=np.uint32), 2 * E)
    assert g.ragged
    sg, so = g.get_state(), o.get_state()
    for k in sg:
        np.testing.assert_array_equal(sg[k], so[k], err_msg=k)
    restarts = 0
    for t in range(200):
        og = g.step(human_policy=_abi.HUMAN_ORCA, robot_policy=_abi.ROBOT_LINEAR, flags=_abi.FLAG_AUTO_RESET)
        oo = o.step(human_policy=_abi.HUMAN_ORCA, robot_policy=_abi.ROBOT_LINEAR, flags=_abi.FLAG_AUTO_RESET)
        _compare_step(og, oo, "generated pool step %d" % t)
        restarts += int(og["done"].sum())
    assert restarts > 2 * E
    np.testing.assert_array_equal(g.row_counts(), o.row_counts())
    # a partial reset: envs 5..14 from other seeds, the rest untouched
    before = g.get_state()
    g.generate_reset(gen, 777, first=5, n=10)
    after = g.get_state()
    part = g.generate_scenes(gen, 777, 10)
    np.testing.assert_array_equal(after["px"][5:15], part.px)
    np.testing.assert_array_equal(after["global_time"][5:15], 0)
    np.testing.assert_array_equal(after["px"][:5], before["px"][:5])
    np.testing.assert_array_equal(after["px"][15:], before["px"][15:])


@pytest.mark.gpu
def test_generate_refuses_what_does_not_fit():
    from ebcsim import _capi
    from ebcsim.batched import BatchedEnv
    c = next(_variants())
    gen = ebc_scene.gen_struct(c, "test")
    params = params_of(load("traj_n10_walls_t17_orcasub"))
    with pytest.raises(_capi.EbcError, match="max_humans"):
        BatchedEnv(params, 4, 5, 16).generate_reset(gen, 1)
    with pytest.raises(_capi.EbcError, match="max_static"):
        BatchedEnv(params, 4, 9, 2).generate_reset(gen, 1)
    bad = copy.deepcopy(gen)
    bad.rule[2] = _abi.RULE_CIRCLE_CROSSING
    with pytest.raises(_capi.EbcError, match="children"):
        BatchedEnv(params, 4, 9, 16).generate_reset(bad, 1)
