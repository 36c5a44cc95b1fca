"""Pins the CPU oracle (oracle/) against golden vectors produced by importing the
reference (tests/golden/make_golden.py).  Masks and codes bit-exact; float64 values
to 1e-12 (they are expected to be identical); float32 rotated rows to 1e-5
(north_star's tolerance).  No GPU needed."""
import json
import os

import numpy as np
import pytest

from ebcsim import _abi, actions as ebc_actions, config as ebc_config, scene as ebc_scene
from oracle import oracle
from helpers import (GOLDEN, TRAJ_IL, TRAJ_ORCASUB, TRAJ_PINNED, batch_from_init, check_trajectory, load,
                     params_of)


def test_reference_unit_collisions():
    """tests/test_collisions.py:12-143: the reference's own six known answers."""
    z = load("collisions")
    n = int(z["n_unit"])
    for k in range(n):
        _, c = oracle.collision_agent_robot(z["h"][k], z["r"][k], int(z["kin"][k]), z["act"][k],
                                            float(z["dt"][k]), float(z["dmin_in"][k]))
        assert c == bool(z["unit_expected"][k])
        assert c == bool(z["coll"][k])


def test_collisions_random():
    z = load("collisions")
    bad = 0
    for k in range(len(z["h"])):
        d, c = oracle.collision_agent_robot(z["h"][k], z["r"][k], int(z["kin"][k]), z["act"][k],
                                            float(z["dt"][k]), float(z["dmin_in"][k]))
        assert c == bool(z["coll"][k]), k
        ref = float(z["dmin_out"][k])
        if np.isinf(ref):
            assert np.isinf(d)
        else:
            assert abs(d - ref) <= 1e-12, (k, d, ref)
            bad += d != ref
    # holonomic rows are expected to be bit-identical; unicycle rows go through libm cos/sin
    assert bad <= 0.02 * len(z["h"])


def test_point_to_segment():
    z = load("collisions")
    got = np.array([oracle.point_to_segment_dist(*row) for row in z["seg"]])
    np.testing.assert_array_equal(got, z["seg_dist"])


def test_reward_branches():
    z = load("reward")
    seen = set()
    for ci in range(int(z["n_configs"])):
        params = params_of(z, "params_%d" % ci)
        rin, rout = z["in_%d" % ci], z["out_%d" % ci]
        for row, exp in zip(rin, rout):
            robot, a, t = row[:9], row[9:11], row[11]
            dmin, coll = row[12:15], row[15:19].astype(np.int32)
            r, done, info, dg = oracle.reward(params, robot, a, t, dmin, coll)
            assert info == int(exp[2]) and done == bool(exp[1])
            if np.isnan(exp[0]):
                assert np.isnan(r)
            else:
                assert abs(r - exp[0]) <= 1e-12
            if not np.isnan(exp[3]):
                assert abs(dg - exp[3]) <= 1e-12
            seen.add(info)
    assert seen == set(range(8)), "every Info class must be exercised"


def test_grid_window():
    z = load("grid")
    border = z["border"]
    total = hits = 0
    for k in range(int(z["n"])):
        packed = ebc_scene.pack_grid(1.0 - z["grid_%d" % k])
        for (px, py, rad, use_border), exp in zip(z["pts_%d" % k], z["coll_%d" % k]):
            got = oracle.grid_collision(packed, packed.shape[0], float(z["map_size_m"]),
                                        float(z["map_resolution"]), px, py, rad,
                                        border if use_border else None)
            assert got == bool(exp), (k, px, py, rad)
            total += 1
            hits += got
    assert 0.05 < hits / total < 0.95


def test_rotate_rows():
    z = load("rotate")
    for vi in range(int(z["n"])):
        got = oracle.rotate_rows(z["in_%d" % vi], int(z["with_agent_type_%d" % vi]),
                                 int(z["unicycle_%d" % vi]))
        np.testing.assert_allclose(got, z["out_%d" % vi], atol=1e-5, rtol=1e-5)


def test_action_space():
    z = load("action_space")
    for k in range(int(z["n"])):
        meta = json.loads(str(z["meta_%d" % k]))
        got = ebc_actions.build_action_space(meta["v_pref"], meta["kinematics"],
                                             meta["speed_samples"], meta["rotation_samples"])
        np.testing.assert_array_equal(got, z["space_%d" % k])


def test_scene_generator_matches_reference():
    """SceneGenerator outputs for fixed seeds: humans, grid, static rows (bit-exact)."""
    import configparser
    z = load("scenes")
    for k in range(int(z["n"])):
        meta = json.loads(str(z["meta_%d" % k]))
        cfg = configparser.RawConfigParser()
        cfg.read_string(meta["config_text"])
        sc = ebc_scene.generate_scene(ebc_scene.SceneConfig.from_config(cfg), meta["seed"])
        b = ebc_scene.SceneBatch.from_scenes([sc])
        for key in ("px", "py", "gx", "gy", "radius", "v_pref", "type"):
            np.testing.assert_array_equal(getattr(b, key)[0], z["%s_%d" % (key, k)], err_msg=key)
        np.testing.assert_array_equal((sc.grid == 0).astype(np.uint8), z["grid_%d" % k])
        np.testing.assert_array_equal(sc.static_rows, z["static_%d" % k])
        np.testing.assert_array_equal(sc.robot, z["robot_%d" % k])
        assert [[list(map(float, p)) for p in poly] for poly in sc.obstacle_vertices] == meta["vertices"]


@pytest.mark.parametrize("name", TRAJ_PINNED)
def test_trajectory_pinned(name):
    """Reference orchestration end to end with the reference's own `linear` humans."""
    z = load(name)
    params = params_of(z)
    b = batch_from_init(z)
    env = oracle.OracleEnv(params, 1, b.N, b.S)
    env.reset(b)
    check_trajectory(env, z, atol=1e-12)


@pytest.mark.parametrize("name", TRAJ_ORCASUB)
def test_trajectory_orca_substituted(name):
    """Reference orchestration around ORCA (rvo2 replaced by the oracle's restatement):
    pins marshalling/order/update, NOT ORCA arithmetic (unpinned)."""
    z = load(name)
    params = params_of(z)
    b = batch_from_init(z)
    env = oracle.OracleEnv(params, 1, b.N, b.S)
    env.reset(b)
    check_trajectory(env, z, atol=1e-12)


@pytest.mark.parametrize("name", TRAJ_IL)
def test_trajectory_robot_on_orca(name):
    """The imitation-learning demonstrator: Robot.act -> ORCA.predict over the returned observation
    (humans + static obstacles as pedestrians), rvo2 replaced by the oracle's restatement, and the
    states the reference's explorer keeps for the IL memory."""
    z = load(name)
    params = params_of(z)
    b = batch_from_init(z)
    env = oracle.OracleEnv(params, 1, b.N, b.S)
    env.reset(b)
    check_trajectory(env, z, atol=1e-12)


def test_known_answer_scenes():
    """tests/test_collisions_simulation.py:12-32: eight frozen scenes, linear robot, ORCA humans,
    expected terminal Info class.  Runs scene loader -> oracle env with its own ORCA."""
    import configparser
    with open(os.path.join(GOLDEN, "known_answers.json")) as f:
        table = json.load(f)
    assert len(table) == 8
    for row in table:
        cfg = configparser.RawConfigParser()
        cfg.read_string(row["config_text"])
        sc = ebc_scene.load_scene(ebc_scene.SceneConfig.from_config(cfg),
                                  os.path.join(GOLDEN, "scenes", row["scene"]))
        b = ebc_scene.SceneBatch.from_scenes([sc])
        params = ebc_config.params_from_dict(row["params"])
        env = oracle.OracleEnv(params, 1, b.N, b.S)
        env.reset(b)
        info = None
        for _ in range(400):
            out = env.step(human_policy=_abi.HUMAN_ORCA, robot_policy=_abi.ROBOT_LINEAR)
            if out["done"][0]:
                info = int(out["info"][0])
                break
        assert info == row["expected_code"], (row["scene"], info, row["expected"])
        # and the whole trajectory equals what the reference's env produced with the same ORCA
        z = load("known_" + row["scene"].replace(".json", "") + "_orcasub")
        env.reset(b)
        check_trajectory(env, z, atol=1e-12)


def test_observe_matches_step_outputs():
    z = load("traj_n10_walls_t17")
    b = batch_from_init(z)
    env = oracle.OracleEnv(params_of(z), 1, b.N, b.S)
    env.reset(b)
    ob, obs = env.observe()
    n = len(z["init_px"])
    np.testing.assert_array_equal(ob[0, :n, 0], z["init_px"])
    out = env.step(robot_action=z["action"][0][None], human_policy=_abi.HUMAN_LINEAR)
    ob, obs = env.observe()
    np.testing.assert_array_equal(ob, out["ob"])
    np.testing.assert_array_equal(obs, out["obs_rotated"])
    np.testing.assert_allclose(obs[0, :len(z["rot"][0])], z["rot"][0], atol=1e-5, rtol=1e-5)


def test_angular_local_map():
    """simulator/env.py:468-628 restated on the host (ebcsim/local_map.py), bit-exact."""
    from ebcsim.local_map import angular_map
    z = load("local_map")
    for k in range(int(z["n"])):
        m = json.loads(str(z["meta_%d" % k]))
        for pose, ref in zip(z["pose_%d" % k], z["map_%d" % k]):
            got = angular_map(m["vertices"], pose[0], pose[1], pose[2], pose[3], m["max_range"],
                              m["dim"], m["angle_min"], m["angle_max"])
            np.testing.assert_array_equal(got, ref)
    assert (z["map_0"] < 1).any()


def test_scene_pool_semantics_cpu():
    """Auto-reset walks the pool: env e restarts from scenes e, e + E, e + 2E, ... (mod P)."""
    z = load("traj_a5_linear_orcasub")
    params = params_of(z)
    params.time_limit = 1  # every env times out at step 5 (t = 1.0 >= 1)
    b = batch_from_init(z, copies=2)
    pool = batch_from_init(z, copies=6)
    for c in range(6):
        pool.px[c] += 0.001 * c  # make pool scenes distinguishable
        pool.n_humans[c] = 5 - (c % 2)
    env = oracle.OracleEnv(params, 2, b.N, b.S)
    env.reset(b)
    env.set_scene_pool(pool, stride=2)
    visited = {0: [], 1: []}
    for t in range(30):
        out = env.step(human_policy=_abi.HUMAN_ORCA, robot_policy=_abi.ROBOT_LINEAR,
                       flags=_abi.FLAG_AUTO_RESET)
        if out["done"].all():
            st = env.get_state()
            for e in (0, 1):
                c = int(round((st["px"][e, 0] - z["init_px"][0]) / 0.001))
                visited[e].append(c)
                assert st["n_humans"][e] == pool.n_humans[c] and st["global_time"][e] == 0
    assert visited[0][:4] == [0, 2, 4, 0] and visited[1][:4] == [1, 3, 5, 1]
    # without a pool an env keeps restarting from its own reset() scene
    env2 = oracle.OracleEnv(params, 2, b.N, b.S)
    env2.reset(b)
    for t in range(12):
        out = env2.step(human_policy=_abi.HUMAN_ORCA, robot_policy=_abi.ROBOT_LINEAR,
                        flags=_abi.FLAG_AUTO_RESET)
        if out["done"].all():
            np.testing.assert_array_equal(env2.get_state()["px"], b.px)


@pytest.mark.parametrize("name", ["il_persistent_const_rows", "il_persistent_wall_rows"])
def test_demonstrator_keeps_its_simulator_across_episodes(name):
    """simulator/policy/orca.py:96-133 + rl/train.py:130-133: the reference's own IL loop on one il_policy."""
    from helpers import check_il_persistent
    check_il_persistent(lambda p, E, N, S: oracle.OracleEnv(p, E, N, S), name)


def test_scene_pool_reinstalled_on_running_episodes_cpu():
    """A running env keeps the map of the scene it is on when the pool is replaced (smaller, then larger)."""
    from helpers import pool_reinstall_run
    kinds = set()
    for phase, t, outs in pool_reinstall_run([lambda p, E, N, S: oracle.OracleEnv(p, E, N, S)]):
        if phase >= 0:
            kinds.update(outs[0]["info"].tolist())
    assert _abi.INFO_COLLISION_OBSTACLE in kinds  # the maps matter in this workload


def test_threaded_oracle_equals_scalar():
    """bench.py's all-cores CPU baseline splits orc_step's env loop over OpenMP threads: same
    results as the one-thread port, every output and the state, through restarts."""
    import configparser
    pkg = os.path.join(os.path.dirname(GOLDEN), "..", "eb-cadrl_amd", "configs")
    cfg, pol = configparser.RawConfigParser(), configparser.RawConfigParser()
    cfg.read(os.path.join(pkg, "bench_metric.config"))
    pol.read(os.path.join(pkg, "policy_agent_type.config"))
    params = ebc_config.params_from_config(cfg, pol)
    sc = ebc_scene.SceneConfig.from_config(cfg)
    batch = ebc_scene.SceneBatch.from_scenes([ebc_scene.generate_scene(sc, 2000 + e) for e in range(24)])
    kw = dict(human_policy=_abi.HUMAN_ORCA, robot_policy=_abi.ROBOT_LINEAR, flags=_abi.FLAG_AUTO_RESET)
    runs = []
    for threads in (1, 3):
        env = oracle.OracleEnv(params, batch.n, batch.N, batch.S)
        env.reset(batch)
        assert oracle.set_threads(threads) == threads
        try:
            outs = [env.step(**kw) for _ in range(80)]
        finally:
            oracle.set_threads(1)
        runs.append((outs, {k: v.copy() for k, v in env.a.items()}))
    assert any(o["done"].any() for o in runs[0][0])  # restarts happened
    for a, b in zip(runs[0][0], runs[1][0]):
        for k in a:
            assert np.array_equal(a[k], b[k], equal_nan=True), k
    for k in runs[0][1]:
        assert np.array_equal(runs[0][1][k], runs[1][1][k], equal_nan=True), k
