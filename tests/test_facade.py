"""The single-env object facade (ebcsim.env) against the reference's own scenario tests and the
golden trajectories.  On CPU the facade is driven with the oracle injected as backend (the
object plumbing is what is under test); with -m gpu the same scenarios run on the HIP library."""
import configparser
import json
import os

import numpy as np
import pytest

from ebcsim import _abi
from ebcsim import env as ebc_env
from ebcsim import info as ebc_info
from ebcsim.action import ActionRot, ActionXY
from ebcsim.agents import Robot
from ebcsim.policy import policy_factory
from ebcsim.state import FullState, JointState, ObservableState
from helpers import GOLDEN, load


def _oracle_backend(params, E, N, S):
    from oracle import oracle
    return oracle.OracleEnv(params, E, N, S)


def _make(cfg_text, backend, kinematics=None):
    cfg = configparser.RawConfigParser()
    cfg.read_string(cfg_text)
    env = ebc_env.make(backend_factory=backend)
    env.configure(cfg)
    robot = Robot(cfg, "robot")
    env.set_robot(robot)
    pol = policy_factory["linear"]()
    robot.set_policy(pol)
    if kinematics:
        pol.kinematics = robot.kinematics = kinematics
    pol.set_phase("test")
    return env, robot


def _known_answers(backend):
    """tests/test_collisions_simulation.py:35-69 of the reference, call for call."""
    with open(os.path.join(GOLDEN, "known_answers.json")) as f:
        table = json.load(f)
    for row in table:
        env, robot = _make(row["config_text"], backend)
        ob, local_map = env.reset("test", load_scene_path=os.path.join(GOLDEN, "scenes", row["scene"]))
        done = False
        steps = 0
        while not done:
            action = robot.act(ob, local_map=local_map, env=env)
            ob, _, reward, done, info = env.step(action)
            steps += 1
            assert steps < 500
        assert isinstance(info, getattr(ebc_info, row["expected"])), (row["scene"], str(info))
        assert len(env.states) == steps and isinstance(env.states[0][0], FullState)


def _golden_trajectory(name, backend):
    z = load(name)
    meta = json.loads(str(z["meta"]))
    zs = load("scenes")
    text = None
    for k in range(int(zs["n"])):
        m = json.loads(str(zs["meta_%d" % k]))
        if m["config"] == meta["config"]:
            cfg = configparser.RawConfigParser()
            cfg.read_string(m["config_text"])
            for key, val in meta["overrides"].items():
                sec, opt = key.split(".")
                if not cfg.has_section(sec):
                    cfg.add_section(sec)
                cfg.set(sec, opt, str(val))
            if meta["human_policy"] != "linear":
                for sec in ("adults", "bicycles", "children"):
                    if cfg.has_section(sec):
                        cfg.set(sec, "policy", "orca")
            import io
            buf = io.StringIO()
            cfg.write(buf)
            text = buf.getvalue()
            break
    kin = None if meta["kinematics"] == "holonomic" else meta["kinematics"]
    env, robot = _make(text, backend, kin)
    ob, _ = env.reset("test", test_case=meta["seed_case"])
    n = len(z["init_px"])
    np.testing.assert_array_equal([o.px for o in ob[:n]], z["init_px"])
    la_steps = list(z["la_step"]) if "la_step" in z.files else []
    for t in range(len(z["action"])):
        a = z["action"][t]
        action = ActionXY(*a) if kin is None else ActionRot(*a)
        if t in la_steps:
            k = la_steps.index(t)
            for ai in (0, 7, 40, 80):
                la = z["la_actions"][ai]
                cand = ActionXY(*la) if kin is None else ActionRot(*la)
                nob, r, d, inf = env.onestep_lookahead(cand)
                assert d == bool(z["la_done"][k][ai])
                np.testing.assert_allclose(r, z["la_reward"][k][ai], atol=1e-9)
                np.testing.assert_allclose([[o.px, o.py, o.vx, o.vy, o.radius] for o in nob],
                                           z["la_next_ob"][k][:, :5], atol=1e-9)
        ob, local_map, reward, done, info = env.step(action)
        assert local_map.shape == (48,) and (local_map <= 1).all()
        assert done == bool(z["done"][t]), t
        assert type(info).__name__ == ["Nothing", "Danger", "ReachGoal", "CollisionObstacle",
                                       "CollisionAdult", "CollisionBicycle", "CollisionChild",
                                       "Timeout"][int(z["info"][t])]
        np.testing.assert_allclose(reward, z["reward"][t], atol=1e-9)
        rows = np.array([[o.px, o.py, o.vx, o.vy, o.radius, int(o.obj_type)] for o in ob])
        np.testing.assert_allclose(rows, z["ob"][t], atol=1e-9)
        assert all(isinstance(o, ObservableState) for o in ob)
        if not np.isnan(z["min_dist"][t]):
            np.testing.assert_allclose(info.min_dist, z["min_dist"][t], atol=1e-9)
        np.testing.assert_allclose(env.global_time, z["time"][t], atol=1e-12)
        np.testing.assert_allclose([robot.px, robot.py, robot.vx, robot.vy, robot.theta],
                                   z["robot"][t][[0, 1, 2, 3, 8]], atol=1e-9)
        times = list(env.adult_times) + list(env.bicycle_times) + list(env.children_times)
        np.testing.assert_allclose(times, z["arrival"][t], atol=1e-12)
    # JointState of the robot and the returned ob is what rotate() consumes
    js = JointState(robot.get_full_state(), ob)
    assert len(js.self_state + js.agent_states[0]) == 15


def test_value_objects_concatenate_like_the_reference():
    f = FullState(1, 2, 3, 4, 5, 6, 7, 8, 9, 4)
    o = ObservableState(10, 11, 12, 13, 14, 2)
    assert f + o == (1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 2)
    assert f.position == (1, 2) and f.goal_position == (6, 7) and o.velocity == (12, 13)
    assert str(ebc_info.ReachGoal()) == "Reaching goal" and str(ebc_info.Nothing()) == ""
    assert str(ebc_info.Danger(0.1)) == "Too close" and str(ebc_info.Timeout()) == "Timeout"


def test_simulator_module_paths_resolve():
    from simulator.utils.info import ReachGoal, CollisionAdult  # noqa: F401
    from simulator.utils.action import ActionXY as A  # noqa: F401
    from simulator.utils.state import JointState as J  # noqa: F401
    from simulator.agents.robot import Robot as R  # noqa: F401
    from simulator.policy.policy_factory import policy_factory as pf
    from simulator.env import EntityBasedCollisionAvoidance as E  # noqa: F401
    assert "linear" in pf and ReachGoal is ebc_info.ReachGoal


def test_reset_errors_like_the_reference():
    z = json.load(open(os.path.join(GOLDEN, "known_answers.json")))[0]
    cfg = configparser.RawConfigParser()
    cfg.read_string(z["config_text"])
    env = ebc_env.make(backend_factory=_oracle_backend)
    env.configure(cfg)
    with pytest.raises(AttributeError):
        env.reset("test")
    env.set_robot(Robot(cfg, "robot"))
    with pytest.raises(AssertionError):
        env.reset("bogus")


def test_known_answer_scenes_cpu_backend():
    _known_answers(_oracle_backend)


@pytest.mark.parametrize("name", ["traj_a5_scripted", "traj_a3b3s2_scripted_orcasub",
                                  "traj_unicycle_rotpen", "traj_n10_walls_t17_orcasub"])
def test_golden_trajectory_cpu_backend(name):
    _golden_trajectory(name, _oracle_backend)


@pytest.mark.gpu
def test_known_answer_scenes_gpu():
    _known_answers(None)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["traj_a5_scripted", "traj_a3b3s2_scripted_orcasub",
                                  "traj_unicycle_rotpen", "traj_n10_walls_t17_orcasub",
                                  "traj_a5_linear_orcasub"])
def test_golden_trajectory_gpu(name):
    _golden_trajectory(name, None)


def test_save_load_map_roundtrip(tmp_path):
    """tests/test_save_load_map.py:15-30 of the reference: reset() saving its generated scene,
    a second env loading it -> equal occupancy maps; here also equal agents, static rows and the
    same first observation."""
    zs = load("scenes")
    text = None
    for k in range(int(zs["n"])):
        m = json.loads(str(zs["meta_%d" % k]))
        if m["config"].endswith("env_adults_5_bikes_5_static_5.config"):
            text = m["config_text"]
            break
    assert text is not None
    path = str(tmp_path / "scene.json")
    env1, _ = _make(text, _oracle_backend)
    ob1, _ = env1.reset("test", test_case=3, save_scene_path=path)
    assert os.path.exists(path)
    env2, _ = _make(text, _oracle_backend)
    ob2, _ = env2.reset("test", load_scene_path=path)
    assert env1.scene.map is not None and (env1.scene.map == 0).any()  # obstacles were placed
    np.testing.assert_array_equal(env1.scene.map, env2.scene.map)
    assert len(ob1) == len(ob2)
    for a, b in zip(ob1, ob2):
        assert (a.px, a.py, a.vx, a.vy, a.radius, a.obj_type) == (b.px, b.py, b.vx, b.vy, b.radius, b.obj_type)
    for h1, h2 in zip(env1.scene.current.humans, env2.scene.current.humans):
        assert (h1.gx, h1.gy, h1.v_pref, h1.type) == (h2.gx, h2.gy, h2.v_pref, h2.type)
    np.testing.assert_array_equal(np.array(env1.scene.obstacle_vertices), np.array(env2.scene.obstacle_vertices))


def _il_episode(name, backend):
    """rl/utils/explorer.py:33-45 with the robot on ORCA (rl/train.py:124-132): reset() returns
    (ob, obstacle_vertices, local_map); robot.act -> ORCA.predict -> env.step, call for call."""
    import io
    z = load(name)
    meta = json.loads(str(z["meta"]))
    zs = load("scenes")
    text = None
    for k in range(int(zs["n"])):
        m = json.loads(str(zs["meta_%d" % k]))
        if m["config"] == meta["config"]:
            cfg = configparser.RawConfigParser()
            cfg.read_string(m["config_text"])
            for key, val in meta["overrides"].items():
                sec, opt = key.split(".")
                cfg.set(sec, opt, str(val))
            buf = io.StringIO()
            cfg.write(buf)
            text = buf.getvalue()
            break
    env, robot = _make(text, backend)
    pol = policy_factory["orca"]()
    pol.multiagent_training = True
    pol.safety_space = meta["il_safety_space"]
    robot.set_policy(pol)
    ret = env.reset("test", test_case=meta["seed_case"])
    assert len(ret) == 3  # env.py:201-204
    ob, vertices, local_map = ret
    assert vertices is env.scene.obstacle_vertices
    done, t = False, 0
    while not done:
        action = robot.act(ob, local_map=local_map, env=env)
        assert isinstance(action, ActionXY)
        np.testing.assert_allclose([action.vx, action.vy], z["action"][t], atol=1e-12)
        assert pol.last_state.self_state.px == robot.px
        ob, local_map, reward, done, info = env.step(action)
        np.testing.assert_allclose(reward, z["reward"][t], atol=1e-9)
        t += 1
    assert t == len(z["action"])
    assert int(z["info"][-1]) == {"ReachGoal": 2, "Timeout": 7}[type(info).__name__]


@pytest.mark.parametrize("name", ["traj_a5_il_orcasub", "traj_n10_walls_il_orcasub"])
def test_imitation_learning_episode_cpu_backend(name):
    _il_episode(name, _oracle_backend)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["traj_a5_il_orcasub", "traj_n10_walls_il_orcasub"])
def test_imitation_learning_episode_gpu(name):
    _il_episode(name, None)


def test_policy_transform_without_env_is_the_reference_rotate():
    """`policy.transform(state)` with ONE argument (rl/utils/explorer.py:162 calls the target policy that way): the
    host rotate equals the reference's rotate() goldens (cadrl.py:236-337; T = 13, 17 and the unicycle frame) and the
    rows the env keeps for the same state."""
    import torch
    from ebcsim.rl_policy import SARL
    from ebcsim.state import FullState, JointState, ObservableState
    z = load("rotate")
    for v in range(int(z["n"])):
        pol = SARL()
        pol.with_agent_type = bool(z["with_agent_type_%d" % v])
        pol.agent_type_state_dim = 4 if pol.with_agent_type else 0
        pol.kinematics = "unicycle" if int(z["unicycle_%d" % v]) else "holonomic"
        pol.device = torch.device("cpu")
        rows = z["in_%d" % v]
        got = pol.rotate(torch.Tensor(rows.tolist())).numpy()
        np.testing.assert_allclose(got, z["out_%d" % v], atol=2e-6, rtol=2e-6)
        # through transform(): a JointState of one robot row and a few "others"
        r = rows[0]
        js = JointState(FullState(*[float(x) for x in r[:9]]),
                        [ObservableState(*[float(x) for x in q[9:14]], int(q[14])) for q in rows[:5]])
        want = pol.rotate(torch.Tensor([list(r[:9]) + list(q[9:15]) for q in rows[:5]]))
        np.testing.assert_array_equal(pol.transform(js).numpy(), want.numpy())


def _facade_il_episodes(name, backend):
    """rl/train.py:120-133 + explorer.py:33-45 through the facade: ONE ORCA policy object given to the robot plays the
    fixture's consecutive `train` episodes (env.reset("train") draws the reference's seeds); its actions are the
    reference's in every episode — the object keeps its simulator (radii of the episode that built it)."""
    import configparser
    from ebcsim import env as ebc_env
    from ebcsim.agents import Robot
    from ebcsim.policy import policy_factory
    from helpers import config_text_of
    z = load(name)
    meta = json.loads(str(z["meta"]))
    cfg = configparser.RawConfigParser()
    cfg.read_string(config_text_of(meta))
    env = ebc_env.make(backend_factory=backend)
    env.configure(cfg)
    robot = Robot(cfg, "robot")
    il_policy = policy_factory["orca"]()
    il_policy.multiagent_training = True
    il_policy.safety_space = float(z["safety_space"])
    robot.set_policy(il_policy)
    env.set_robot(robot)
    il_policy.set_phase("train")
    for k in range(int(z["n_episodes"])):
        ob, _, local_map = env.reset("train")
        assert len(ob) == int(z["rows"][k])
        done, t = False, 0
        while not done:
            action = robot.act(ob, local_map=local_map, env=env)
            np.testing.assert_allclose([action.vx, action.vy], z["action%d" % k][t], atol=1e-9, rtol=0,
                                       err_msg="episode %d step %d" % (k, t))
            ob, local_map, reward, done, info = env.step(action)
            assert abs(reward - z["reward%d" % k][t]) <= 1e-9
            t += 1
        assert t == len(z["action%d" % k])


@pytest.mark.parametrize("name", ["il_persistent_const_rows", "il_persistent_wall_rows"])
def test_orca_policy_object_keeps_its_simulator_cpu_backend(name):
    from oracle import oracle
    _facade_il_episodes(name, lambda p, E, N, S: oracle.OracleEnv(p, E, N, S))


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["il_persistent_const_rows", "il_persistent_wall_rows"])
def test_orca_policy_object_keeps_its_simulator_gpu(name):
    _facade_il_episodes(name, None)
