"""The drop-in boundary: every module path the reference's callers import from `simulator`
resolves here; host functions behind those paths equal the reference's goldens; the reference's own `rl`
package (read in place when the reference tree is present, i.e. in the build container) imports and
drives this repo's env unchanged."""
import configparser
import importlib
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from ebcsim import _abi
from ebcsim import env as ebc_env
from ebcsim import info as ebc_info
from ebcsim.action import ActionRot, ActionXY
from helpers import GOLDEN, load

REFERENCE = os.environ.get("EBC_REFERENCE_ROOT", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
needs_reference = pytest.mark.skipif(not os.path.isdir(os.path.join(REFERENCE, "rl")),
                                     reason="the reference tree only exists in the build container")

# `grep -rhn "from simulator" rl tests` over the reference, de-duplicated (module, names):
SIMULATOR_IMPORTS = [
    ("simulator.agents.robot", ["Robot"]),
    ("simulator.agents.agents", ["Adult"]),
    ("simulator.policy.orca", ["ORCA"]),
    ("simulator.policy.policy", ["Policy"]),
    ("simulator.policy.policy_factory", ["policy_factory"]),
    ("simulator.utils.action", ["ActionRot", "ActionXY"]),
    ("simulator.utils.collisions", ["compute_collision_agent_with_robot"]),
    ("simulator.utils.info", ["ReachGoal", "Collision", "CollisionChild", "CollisionAdult", "CollisionBicycle",
                              "CollisionObstacle", "Timeout", "Danger", "Nothing"]),
    ("simulator.utils.state", ["ObservableState", "FullState", "JointState"]),
    ("simulator.utils.test_utils", ["configure_env_policy_robot"]),
    ("simulator.utils.utils", ["AgentType"]),
]


def test_every_simulator_import_of_the_reference_callers_resolves():
    for module, names in SIMULATOR_IMPORTS:
        m = importlib.import_module(module)
        for n in names:
            assert hasattr(m, n), (module, n)
    from simulator.utils.info import __all__ as star  # `from simulator.utils.info import *` (tests/*.py)
    assert {"ReachGoal", "CollisionAdult", "CollisionBicycle", "CollisionChild", "CollisionObstacle"} <= set(star)


@needs_reference
def test_import_list_covers_the_reference_tree():
    """Re-derive the list from the reference's sources: a new `from simulator...` line there fails here."""
    have = {m: set(n) for m, n in SIMULATOR_IMPORTS}
    pat = re.compile(r"^\s*from\s+(simulator[\w.]*)\s+import\s+(.*)$")
    for sub in ("rl", "tests"):
        for dirpath, _, files in os.walk(os.path.join(REFERENCE, sub)):
            for f in files:
                if not f.endswith(".py"):
                    continue
                text = open(os.path.join(dirpath, f)).read()
                text = re.sub(r"\(\s*([^)]*?)\s*\)", lambda mo: mo.group(1).replace("\n", " "), text)
                for line in text.splitlines():
                    mo = pat.match(line)
                    if not mo:
                        continue
                    module, names = mo.group(1), [x.strip() for x in mo.group(2).split(",") if x.strip()]
                    assert module in have, (f, module)
                    for n in names:
                        assert n == "*" or n in have[module] or hasattr(importlib.import_module(module), n), (f, module, n)


def test_host_collision_functions_equal_the_reference_goldens():
    """simulator/utils/collisions.py:4-57 behind its module path: the reference's six unit cases
    (tests/test_collisions.py:12-143) and 12 002 random pairs, bit for bit."""
    from simulator.utils.collisions import compute_collision_agent_with_robot, point_to_segment_dist
    from types import SimpleNamespace as NS
    z = dict(load("collisions"))  # an NpzFile decompresses on every access
    for s, d in zip(z["seg"], z["seg_dist"]):
        assert point_to_segment_dist(*s) == d
    n_unit = int(z["n_unit"])
    for i in range(len(z["h"])):
        h, r = z["h"][i], z["r"][i]
        agent = NS(px=h[0], py=h[1], vx=h[2], vy=h[3], radius=h[4])
        robot = NS(px=r[0], py=r[1], theta=r[2], radius=r[3],
                   kinematics="holonomic" if z["kin"][i] == _abi.HOLONOMIC else "unicycle")
        a = z["act"][i]
        action = ActionXY(a[0], a[1]) if z["kin"][i] == _abi.HOLONOMIC else ActionRot(a[0], a[1])
        dmin, hit = compute_collision_agent_with_robot(agent, robot, action, z["dmin_in"][i], z["dt"][i])
        assert hit == bool(z["coll"][i]), i
        assert dmin == z["dmin_out"][i] or (np.isinf(dmin) and np.isinf(z["dmin_out"][i])), i
        if i < n_unit:
            assert hit == bool(z["unit_expected"][i])


def test_reference_unit_test_objects():
    """tests/test_collisions.py:12-35 with this package's classes: Robot / Adult built from a config,
    `set`, attribute overrides, one pair."""
    from simulator.agents.agents import Adult, Bicycle, Child
    from simulator.agents.robot import Robot
    from simulator.utils.collisions import compute_collision_agent_with_robot
    from simulator.utils.utils import AgentType
    zs = load("scenes")
    text = None
    for k in range(int(zs["n"])):
        m = json.loads(str(zs["meta_%d" % k]))
        if m["config"].endswith("env_adults_5_bikes_5_static_5.config"):
            text = m["config_text"]
    cfg = configparser.RawConfigParser()
    cfg.read_string(text)
    robot = Robot(cfg, "robot")
    robot.kinematics = "holonomic"
    robot.set(0, 0, 0, 0, 0, 0, np.pi / 2)
    robot.radius = 1
    adult = Adult(cfg, "adults")
    adult.set(0, -2, 0, -2, 0, 0, 0)
    adult.radius = 0.9
    assert adult.agent_type == AgentType.ADULT and Bicycle(cfg, "bicycles").agent_type == AgentType.BICYCLE
    assert Child(cfg, "adults").agent_type == AgentType.CHILD
    for dt, expect in ((0.07, False), (0.12, True)):
        assert compute_collision_agent_with_robot(adult, robot, ActionXY(-1, -1), float("inf"), dt)[1] is expect
    robot.time_step = 0.25
    nxt = robot.get_next_observable_state(ActionXY(1, 0))
    assert (nxt.px, nxt.py, nxt.vx) == (0.25, 0, 1)
    d = robot.get_state_dict()
    robot.set_from_state_dict(d)
    robot.step(ActionXY(1, 0))
    assert robot.get_position() == (0.25, 0) and robot.get_velocity() == (1, 0)
    robot.print_info()


def _rl_test_script_sequence(backend):
    """rl/test.py:95-135 call for call (the --visualize branch; policy `orca`, as its default argument
    says): isinstance(robot.policy, ORCA) with the class rl/test.py:11 imports, print_info, the 3-tuple
    reset, act / step until done, isinstance dispatch on the info object."""
    from simulator.agents.robot import Robot
    from simulator.policy.orca import ORCA
    from simulator.policy.policy_factory import policy_factory
    with open(os.path.join(GOLDEN, "known_answers.json")) as f:
        table = json.load(f)
    seen = set()
    for row in table[:4]:
        cfg = configparser.RawConfigParser()
        cfg.read_string(row["config_text"])
        env = ebc_env.make(backend_factory=backend)
        env.configure(cfg)
        robot = Robot(cfg, "robot")
        policy = policy_factory["orca"]()
        robot.set_policy(policy)
        env.set_robot(robot)
        policy.set_phase("test")
        policy.set_device("cpu")
        assert isinstance(robot.policy, ORCA)
        robot.policy.safety_space = 0
        robot.print_info()
        ob, global_map, local_map = env.reset("test", load_scene_path=os.path.join(GOLDEN, "scenes", row["scene"]))
        done, steps = False, 0
        last_pos = np.array(robot.get_position())
        while not done:
            action = robot.act(ob, local_map=local_map, env=env)
            ob, local_map, reward, done, info = env.step(action)
            cur = np.array(robot.get_position())
            assert np.linalg.norm(cur - last_pos) / robot.time_step <= robot.v_pref * (1 + 1e-5)  # ORCA works in float32
            last_pos = cur
            steps += 1
            assert steps < 500
        assert isinstance(info, (ebc_info.ReachGoal, ebc_info.Timeout, ebc_info.CollisionAdult, ebc_info.CollisionBicycle,
                                 ebc_info.CollisionChild, ebc_info.CollisionObstacle))
        assert abs(env.global_time - steps * env.time_step) < 1e-9 or isinstance(info, ebc_info.Timeout)
        assert str(info)  # rl/test.py:127 formats it into the file name
        seen.add(type(info).__name__)
        # one ORCA evaluation for the humans per step; the robot's own ORCA is a call of its own
        assert env.orca_evaluations == steps
    return seen


def test_rl_test_script_sequence_cpu_backend():
    from oracle import oracle
    _rl_test_script_sequence(lambda p, E, N, S: oracle.OracleEnv(p, E, N, S))


@pytest.mark.gpu
def test_rl_test_script_sequence_gpu():
    _rl_test_script_sequence(None)


def _native_sarl_episode(name, backend, device):
    """tests/test_basic_simulation.py:10-24 of the reference through this package's own `sarl` policy (one
    sweep + one batched forward per decision) on the reference's shipped weights: its actions and its 81
    action values per decision are the reference's (golden from the reference's own policy + simulator)."""
    import torch
    from ebcsim.rl_policy import SARL
    from ebcsim.agents import Robot
    z = load(name)
    meta = json.loads(str(z["meta"]))
    with open(os.path.join(GOLDEN, "sarl_configs.json")) as f:
        texts = json.load(f)[name]
    cfg = configparser.RawConfigParser()
    cfg.read_string(texts["config_text"])
    pcfg = configparser.RawConfigParser()
    pcfg.read_string(texts["policy_config_text"])
    env = ebc_env.make(backend_factory=backend)
    env.configure(cfg)
    robot = Robot(cfg, "robot")
    env.set_robot(robot)
    policy = SARL()
    policy.configure(pcfg)
    policy.get_model().load_state_dict(torch.load(os.path.join(GOLDEN, "weights", meta["weights"]), map_location="cpu"))
    robot.set_policy(policy)
    policy.set_phase("test")
    policy.set_device(device)
    if meta.get("scene_json"):  # tests/test_scene_simulation.py:16
        ob, local_map = env.reset("test", load_scene_path=os.path.join(GOLDEN, "scenes", os.path.basename(meta["scene_json"])))
    else:
        ob, local_map = env.reset("test", test_case=meta["seed_case"])
    done, t, worst, agree = False, 0, 0.0, 0
    while not done:
        action = robot.act(ob, local_map=local_map, env=env)
        if not np.isnan(z["values"][t]).any():
            worst = max(worst, float(np.abs(np.array(policy.action_values) - z["values"][t]).max()))
            agree += tuple(action) == tuple(z["action"][t])
            w = policy.get_attention_weights()
            assert w.shape == (len(ob),) and abs(float(w.sum()) - 1.0) < 1e-5
        # keep the episode on the golden's path: a value tie may be broken differently in float32
        a = z["action"][t]
        ob, local_map, reward, done, info = env.step(ActionXY(a[0], a[1]))
        assert abs(reward - z["reward"][t]) <= 1e-9
        t += 1
    assert isinstance(info, ebc_info.ReachGoal) and t == len(z["action"])
    assert worst <= 5e-5, worst
    assert agree == int((~np.isnan(z["values"]).any(1)).sum()), (agree, t)  # every decision picks the reference's action
    assert env.orca_evaluations == t and env.backend_calls == 2 * t  # one sweep + one step per decision


KNOWN_SARL = ["sarl_a3b3s2_baseline", "sarl_a3b3_baseline", "sarl_scene_a3b3s10_baseline"]  # tests/run_tests.py:23-41


@pytest.mark.parametrize("name", ["sarl_a5_baseline"] + KNOWN_SARL)
def test_native_sarl_policy_cpu_backend(name):
    from oracle import oracle
    _native_sarl_episode(name, lambda p, E, N, S: oracle.OracleEnv(p, E, N, S), "cpu")


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["sarl_a5_baseline", "sarl_n10_ebcadrl"] + KNOWN_SARL)
def test_native_sarl_policy_gpu(name):
    _native_sarl_episode(name, None, "cuda:0")


def _driver(*args):
    r = subprocess.run([sys.executable, os.path.join(HERE, "dropin_driver.py"), REFERENCE] + list(args),
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


@needs_reference
def test_reference_rl_modules_import_against_this_simulator():
    """rl/test.py, rl/train.py and every rl module that imports `simulator` load unchanged."""
    out = _driver("imports")
    assert "rl.test" in out["imported"] and "rl.train" in out["imported"]


@needs_reference
def test_reference_rl_policy_and_explorer_drive_this_env():
    """The reference's SARL policy object and Explorer, unchanged, on this repo's env (oracle backend):
    same actions as with the reference's own simulator, values within float32 noise, and ORCA
    evaluated once per real step although the policy asks 81 times."""
    out = _driver("episode", "sarl_a5_baseline")
    assert out["max_value_err"] <= 5e-5 and out["orca_evaluations"] == out["steps"]


@needs_reference
@pytest.mark.parametrize("name", KNOWN_SARL)
def test_reference_known_answer_runs_on_this_env(name):
    """tests/run_tests.py:23-41 of the reference: run_basic_simulation on its other two configs and
    run_scene_simulation on the frozen scene, with the reference's own SARL policy object on this env."""
    out = _driver("episode", name)
    assert out["max_value_err"] <= 5e-5 and out["orca_evaluations"] == out["steps"]


@needs_reference
def test_reference_run_train_runs_unchanged_on_this_env(tmp_path):
    """rl/train.py:152-276 (`run_train`), called with the parameters of the reference's own smoke test
    (tests/test_basic_train.py:46-92; env_adults_3_bikes_3_child_3_static_3_fast_train.config): 3 imitation-learning
    episodes on the ORCA demonstrator (this package's `simulator.policy.orca.ORCA` through Explorer), a validation sweep
    and a round of 8 train episodes in the reference's Pool(8) workers — each of which rebuilds THIS env from the
    config files and drives it with the pickled SARL policy (81 onestep_lookahead calls per decision) —, optimize_batch,
    checkpoints.  Pass = the reference's criterion (no exception) + its weight files written and loadable here."""
    from ebcsim.sarl import SarlValueNet
    out = _driver("train", str(tmp_path))
    assert out["episode"] == 8 and out["runtime_errors_logged"] == 0 and out["il_memory_logged"]
    assert {"il_model.pth", "rl_model_8.pth"} <= set(out["files"])
    assert out["train_episode_lines"] >= 8 and out["val_episode_lines"] >= 2
    for f in ("il_model.pth", "rl_model_8.pth"):
        net = SarlValueNet.load(os.path.join(str(tmp_path), f))
        assert net is not None
