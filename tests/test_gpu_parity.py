"""GPU parity proper (-m gpu): the HIP path through the C ABI against (a) the golden vectors
of the reference, (b) the CPU oracle on seeded random scene batches, (c) size-independent
properties at BASELINE sizes.  Masks/codes bit-exact; float64 state 1e-9 absolute (ORCA-only
dynamics are expected identical: no transcendental on that path); float32 rows 1e-5."""
import configparser
import json
import os

import numpy as np
import pytest

from ebcsim import _abi, actions as ebc_actions, config as ebc_config, scene as ebc_scene
from helpers import (GOLDEN, TRAJ_IL, TRAJ_ORCASUB, TRAJ_PINNED, batch_from_init, check_trajectory, load,
                     params_of)

pytestmark = pytest.mark.gpu


def _env(params, E, N, S):
    from ebcsim.batched import BatchedEnv
    return BatchedEnv(params, E, N, S)


@pytest.mark.parametrize("name", TRAJ_PINNED + TRAJ_ORCASUB + TRAJ_IL)
def test_golden_trajectories(name):
    z = load(name)
    b = batch_from_init(z, copies=3)
    env = _env(params_of(z), 3, b.N, b.S)
    env.reset(b)
    check_trajectory(env, z, atol=1e-9)


def test_known_answer_scenes():
    with open(os.path.join(GOLDEN, "known_answers.json")) as f:
        table = json.load(f)
    for row in table:
        cfg = configparser.RawConfigParser()
        cfg.read_string(row["config_text"])
        sc = ebc_scene.load_scene(ebc_scene.SceneConfig.from_config(cfg),
                                  os.path.join(GOLDEN, "scenes", row["scene"]))
        b = ebc_scene.SceneBatch.from_scenes([sc])
        env = _env(ebc_config.params_from_dict(row["params"]), 1, b.N, b.S)
        env.reset(b)
        info = None
        for _ in range(400):
            out = env.step(human_policy=_abi.HUMAN_ORCA, robot_policy=_abi.ROBOT_LINEAR)
            if out["done"][0]:
                info = int(out["info"][0])
                break
        assert info == row["expected_code"], (row["scene"], info)
        z = load("known_" + row["scene"].replace(".json", "") + "_orcasub")
        env.reset(b)
        check_trajectory(env, z, atol=1e-9)


def _random_batch(cfg_text, seeds, N=None, S=None):
    cfg = configparser.RawConfigParser()
    cfg.read_string(cfg_text)
    sc = ebc_scene.SceneConfig.from_config(cfg)
    scenes = [ebc_scene.generate_scene(sc, s) for s in seeds]
    return ebc_scene.SceneBatch.from_scenes(scenes, N, S), cfg


def _compare_step(out_g, out_o, tag, atol=1e-9):
    np.testing.assert_array_equal(out_g["done"], out_o["done"], err_msg=tag)
    np.testing.assert_array_equal(out_g["info"], out_o["info"], err_msg=tag)
    for k in ("reward", "dmin", "dist_to_goal", "human_action", "ob", "robot_action_out"):
        np.testing.assert_allclose(out_g[k], out_o[k], atol=atol, rtol=0, err_msg=tag + " " + k)
    np.testing.assert_allclose(out_g["obs_rotated"], out_o["obs_rotated"], atol=1e-5, rtol=1e-5,
                               err_msg=tag)


CASES = [
    # (fixture holding the config text, envs, steps, human policy, flags)
    ("traj_a5_linear_orcasub", 257, 60, _abi.HUMAN_ORCA, 0),
    ("traj_n10_walls_t17_orcasub", 130, 50, _abi.HUMAN_ORCA, 0),
    ("traj_a3b3s2_scripted", 64, 40, _abi.HUMAN_LINEAR, 0),
    ("traj_n10_walls_t17_orcasub", 96, 120, _abi.HUMAN_ORCA, _abi.FLAG_AUTO_RESET),
]


@pytest.mark.parametrize("fixture,E,steps,policy,flags", CASES)
def test_random_batches_vs_oracle(fixture, E, steps, policy, flags):
    """Seeded random scenes (ragged static rows), scripted robot, every output every step."""
    from oracle import oracle
    z = load(fixture)
    meta = json.loads(str(z["meta"]))
    params = params_of(z)
    text = _config_text(meta)
    b, cfg = _random_batch(text, [3000 + e for e in range(E)])
    g = _env(params, E, b.N, b.S)
    o = oracle.OracleEnv(params, E, b.N, b.S)
    g.reset(b)
    o.reset(b)
    space = ebc_actions.build_action_space(float(b.robot[0, 7]))
    rs = np.random.RandomState(7)
    kinds = set()
    for t in range(steps):
        act = space[rs.randint(len(space), size=E)]
        og = g.step(robot_action=act, human_policy=policy, flags=flags)
        oo = o.step(robot_action=act, human_policy=policy, flags=flags)
        _compare_step(og, oo, "%s step %d" % (fixture, t))
        kinds.update(og["info"].tolist())
    sg, so = g.get_state(), o.get_state()
    for k in sg:
        np.testing.assert_allclose(sg[k], so[k], atol=1e-9, rtol=0, err_msg=k)
    assert len(kinds) >= 3


def _config_text(meta):
    """Rebuild the INI text a trajectory fixture ran with from the scenes fixture (same config
    file) plus the overrides recorded in its meta."""
    zs = load("scenes")
    for k in range(int(zs["n"])):
        m = json.loads(str(zs["meta_%d" % k]))
        if m["config"] == meta["config"]:
            cfg = configparser.RawConfigParser()
            cfg.read_string(m["config_text"])
            for key, val in meta["overrides"].items():
                sec, opt = key.split(".")
                if sec in ("adults", "bicycles", "children") and opt == "policy":
                    continue
                cfg.set(sec, opt, str(val))
            import io
            buf = io.StringIO()
            cfg.write(buf)
            return buf.getvalue()
    raise KeyError(meta["config"])


def test_lookahead_vs_oracle_and_cached_step():
    from oracle import oracle
    z = load("traj_n10_walls_t17_orcasub")
    meta = json.loads(str(z["meta"]))
    params = params_of(z)
    E = 48
    b, _ = _random_batch(_config_text(meta), [5000 + e for e in range(E)])
    g = _env(params, E, b.N, b.S)
    o = oracle.OracleEnv(params, E, b.N, b.S)
    g.reset(b)
    o.reset(b)
    space = ebc_actions.build_action_space(float(b.robot[0, 7]))
    rs = np.random.RandomState(9)
    for t in range(12):
        lg = g.lookahead(space, human_policy=_abi.HUMAN_ORCA)
        lo = o.lookahead(space, human_policy=_abi.HUMAN_ORCA)
        np.testing.assert_array_equal(lg["done"], lo["done"])
        np.testing.assert_array_equal(lg["info"], lo["info"])
        for k in ("reward", "dmin", "next_ob"):
            np.testing.assert_allclose(lg[k], lo[k], atol=1e-9, rtol=0, err_msg=k)
        np.testing.assert_allclose(lg["rows_rotated"], lo["rows_rotated"], atol=1e-5, rtol=1e-5)
        act = space[rs.randint(len(space), size=E)]
        # the real step re-uses the human velocities the look-ahead computed (ORCA once per step)
        og = g.step(robot_action=act, human_policy=_abi.HUMAN_CACHED)
        oo = o.step(robot_action=act, human_policy=_abi.HUMAN_ORCA)
        _compare_step(og, oo, "cached step %d" % t)
        # look-ahead reward of the chosen action == reward of the step (same state, same action)
        idx = [int(np.where((space == a).all(1))[0][0]) for a in act]
        np.testing.assert_allclose(lg["reward"][np.arange(E), idx], og["reward"], atol=1e-12)


def test_external_human_actions_config2():
    """BASELINE config 2: humans moved by the host (ORCA on host = the oracle here)."""
    from oracle import oracle
    z = load("traj_a5_linear_orcasub")
    meta = json.loads(str(z["meta"]))
    params = params_of(z)
    E = 200
    b, _ = _random_batch(_config_text(meta), [7000 + e for e in range(E)])
    g = _env(params, E, b.N, b.S)
    o = oracle.OracleEnv(params, E, b.N, b.S)
    g.reset(b)
    o.reset(b)
    for t in range(30):
        oo = o.step(human_policy=_abi.HUMAN_ORCA, robot_policy=_abi.ROBOT_LINEAR)
        g.set_human_actions(oo["human_action"])
        og = g.step(human_policy=_abi.HUMAN_EXTERNAL, robot_policy=_abi.ROBOT_LINEAR)
        _compare_step(og, oo, "external step %d" % t)


def test_partial_reset_and_ragged_humans():
    from oracle import oracle
    z = load("traj_a5_linear_orcasub")
    meta = json.loads(str(z["meta"]))
    params = params_of(z)
    E = 40
    cfg = configparser.RawConfigParser()
    cfg.read_string(_config_text(meta))
    sc = ebc_scene.SceneConfig.from_config(cfg)
    scenes = []
    for e in range(E):
        sc.adult_num = 1 + e % 7      # ragged: 1..7 humans, padded to 8
        scenes.append(ebc_scene.generate_scene(sc, 9000 + e))
    b = ebc_scene.SceneBatch.from_scenes(scenes, 8, 0)
    g = _env(params, E, 8, 0)
    o = oracle.OracleEnv(params, E, 8, 0)
    g.reset(b)
    o.reset(b)
    for t in range(25):
        _compare_step(g.step(human_policy=_abi.HUMAN_ORCA, robot_policy=_abi.ROBOT_LINEAR),
                      o.step(human_policy=_abi.HUMAN_ORCA, robot_policy=_abi.ROBOT_LINEAR),
                      "ragged %d" % t)
    ids = np.array([3, 17, 4, 39], np.int32)
    sub = ebc_scene.SceneBatch.from_scenes([scenes[i] for i in ids], 8, 0)
    g.reset(sub, ids)
    o.reset(sub, ids)
    for t in range(10):
        _compare_step(g.step(human_policy=_abi.HUMAN_ORCA, robot_policy=_abi.ROBOT_LINEAR),
                      o.step(human_policy=_abi.HUMAN_ORCA, robot_policy=_abi.ROBOT_LINEAR),
                      "after partial reset %d" % t)
    sg, so = g.get_state(), o.get_state()
    np.testing.assert_array_equal(sg["global_time"], so["global_time"])


def test_full_size_properties():
    """BASELINE sizes (4096 x 10, ORCA, walls): properties that do not need the oracle.
    - replicas: the same 64 scenes tiled 64x give identical results in every replica;
    - speed limit: |v_human| <= v_pref (LP2 disc), positions finite;
    - done/info consistency; time advances by dt."""
    z = load("traj_n10_walls_t17_orcasub")
    meta = json.loads(str(z["meta"]))
    params = params_of(z)
    base, _ = _random_batch(_config_text(meta), [11000 + e for e in range(64)])
    E = 4096
    tile = lambda a: None if a is None else np.concatenate([a] * (E // 64), axis=0)  # noqa: E731
    b = ebc_scene.SceneBatch(E, base.N, base.S, *[tile(getattr(base, k)) for k in (
        "n_humans", "px", "py", "vx", "vy", "gx", "gy", "radius", "v_pref", "type", "n_static",
        "spx", "spy", "sradius", "grid", "robot")])
    g = _env(params, E, b.N, b.S)
    g.reset(b)
    for t in range(40):
        out = g.step(human_policy=_abi.HUMAN_ORCA, robot_policy=_abi.ROBOT_LINEAR)
        for k in ("reward", "info", "obs_rotated", "human_action"):
            v = out[k].reshape(E // 64, 64, -1)
            assert (v == v[0:1]).all() or np.allclose(v, v[0:1], equal_nan=True), (t, k)
        speed = np.hypot(out["human_action"][..., 0], out["human_action"][..., 1])
        # LP1 intersects float lines with the speed disc: |v| can exceed v_pref by float error
        # (3.5e-4 seen in the oracle too), never by more
        assert (speed <= b.v_pref * (1 + 2e-3) + 1e-3).all()
        terminal = np.isin(out["info"], [_abi.INFO_REACH_GOAL, _abi.INFO_COLLISION_OBSTACLE,
                                         _abi.INFO_COLLISION_ADULT, _abi.INFO_COLLISION_BICYCLE,
                                         _abi.INFO_COLLISION_CHILD, _abi.INFO_TIMEOUT])
        np.testing.assert_array_equal(terminal, out["done"].astype(bool))
    st = g.get_state()
    assert np.isfinite(st["px"]).all() and np.isfinite(st["robot"]).all()
    np.testing.assert_allclose(st["global_time"], 40 * params.time_step, atol=1e-12)


def test_full_size_parity_vs_oracle():
    """The bench workload at its full size (4096 x 10, ORCA, walls, auto-reset, device-resident
    outputs) against the oracle, every env every step: the one-launch step at full occupancy —
    late-starting ORCA waves, consumer roles waiting for slots, restarts — must be what the scalar
    CPU code computes (codes bit-exact, float64 state 1e-9, float32 rows 1e-5)."""
    import torch
    from oracle import oracle
    z = load("traj_n10_walls_t17_orcasub")
    meta = json.loads(str(z["meta"]))
    params = params_of(z)
    params.time_limit = 6  # restarts from step 24 on, for every env that is still running
    E = 4096
    b, _ = _random_batch(_config_text(meta), [21000 + e for e in range(E)])
    g = _env(params, E, b.N, b.S)
    o = oracle.OracleEnv(params, E, b.N, b.S)
    g.reset(b)
    o.reset(b)
    g.use_torch_stream()
    outs = g.alloc_step_outputs(("reward", "done", "info", "obs_rotated", "human_action"))
    fl = _abi.FLAG_AUTO_RESET
    restarts = 0
    for t in range(32):
        g.step_device(outs, human_policy=_abi.HUMAN_ORCA, robot_policy=_abi.ROBOT_LINEAR, flags=fl)
        ref = o.step(human_policy=_abi.HUMAN_ORCA, robot_policy=_abi.ROBOT_LINEAR, flags=fl)
        g.synchronize()
        np.testing.assert_array_equal(outs["done"].cpu().numpy(), ref["done"], err_msg="step %d" % t)
        np.testing.assert_array_equal(outs["info"].cpu().numpy(), ref["info"], err_msg="step %d" % t)
        np.testing.assert_allclose(outs["reward"].cpu().numpy(), ref["reward"], atol=1e-9, rtol=0)
        np.testing.assert_allclose(outs["human_action"].cpu().numpy(), ref["human_action"], atol=1e-9, rtol=0)
        np.testing.assert_allclose(outs["obs_rotated"].cpu().numpy(), ref["obs_rotated"], atol=1e-5, rtol=1e-5)
        restarts += int(ref["done"].sum())
    assert restarts > E  # every env restarted at least once (time limit), many twice
    sg, so = g.get_state(), o.get_state()
    for k in sg:
        np.testing.assert_allclose(sg[k], so[k], atol=1e-9, rtol=0, err_msg=k)


BASELINE_SIZES = [
    # (bench workload, envs on one GPU, humans by, steps): BASELINE.json configs 2, 3 and the per-GPU slice of 4
    ("cfg2", 4096, "external", 12),   # 4096 x 5, human velocities supplied by the host (ORCA on host = the oracle)
    ("cfg3", 16384, "orca", 12),      # 16384 x 10 + 4 walls, ORCA as a HIP kernel
    ("cfg4", 16384, "orca", 12),      # 131072 x 5 over 8 GPUs: one rank's 16384-env slice
]


@pytest.mark.parametrize("workload,E,humans,steps", BASELINE_SIZES)
def test_baseline_sizes_vs_oracle(workload, E, humans, steps):
    """The other BASELINE configurations at their full one-GPU size, every env every step against the
    oracle (its env loop on all host threads), scenes from bench.py's own builder (rank 3 of 8 for the
    config-4 slice: the seeds that rank would own)."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from oracle import oracle
    params, b = bench.build_batch(workload, E, 3 if workload == "cfg4" else 0)
    assert b.n == E and (b.N, b.S) == {"cfg2": (5, 0), "cfg3": (10, 8), "cfg4": (5, 0)}[workload]
    params.time_limit = 2  # restarts inside the window (step 8)
    g = _env(params, E, b.N, b.S)
    o = oracle.OracleEnv(params, E, b.N, b.S)
    g.reset(b)
    o.reset(b)
    g.use_torch_stream()
    outs = g.alloc_step_outputs(("reward", "done", "info", "obs_rotated", "human_action"))
    fl = _abi.FLAG_AUTO_RESET
    oracle.set_threads(bench.host_cores())
    try:
        restarts = 0
        for t in range(steps):
            ref = o.step(human_policy=_abi.HUMAN_ORCA, robot_policy=_abi.ROBOT_LINEAR, flags=fl)
            if humans == "external":
                g.set_human_actions(ref["human_action"])
                g.step_device(outs, human_policy=_abi.HUMAN_EXTERNAL, robot_policy=_abi.ROBOT_LINEAR, flags=fl)
            else:
                g.step_device(outs, human_policy=_abi.HUMAN_ORCA, robot_policy=_abi.ROBOT_LINEAR, flags=fl)
            g.synchronize()
            np.testing.assert_array_equal(outs["done"].cpu().numpy(), ref["done"], err_msg="step %d" % t)
            np.testing.assert_array_equal(outs["info"].cpu().numpy(), ref["info"], err_msg="step %d" % t)
            np.testing.assert_allclose(outs["reward"].cpu().numpy(), ref["reward"], atol=1e-9, rtol=0)
            np.testing.assert_allclose(outs["human_action"].cpu().numpy(), ref["human_action"], atol=1e-9, rtol=0)
            np.testing.assert_allclose(outs["obs_rotated"].cpu().numpy(), ref["obs_rotated"], atol=1e-5, rtol=1e-5)
            restarts += int(ref["done"].sum())
        assert restarts >= E
    finally:
        oracle.set_threads(1)
    sg, so = g.get_state(), o.get_state()
    for k in sg:
        np.testing.assert_allclose(sg[k], so[k], atol=1e-9, rtol=0, err_msg=k)


@pytest.mark.parametrize("robot", ["orca", "linear", "external"])
def test_step_k_equals_k_oracle_steps(robot):
    """ebc_step_k: K steps in one call (rl/utils/explorer.py:33-45 without a host round trip per step) against
    K oracle steps — the state the policy sees, the robot's action, reward / done / info and the returned
    observation of every step, through restarts (auto-reset, short time limit)."""
    from oracle import oracle
    z = load("traj_n10_walls_t17_orcasub")
    meta = json.loads(str(z["meta"]))
    params = params_of(z)
    params.time_limit = 4
    E, K = 70, 45
    b, _ = _random_batch(_config_text(meta), [31000 + e for e in range(E)])
    g = _env(params, E, b.N, b.S)
    o = oracle.OracleEnv(params, E, b.N, b.S)
    g.reset(b)
    o.reset(b)
    keys = ("state_rotated", "n_rows", "robot_action_out", "reward", "done", "info", "dmin", "dist_to_goal", "obs_rotated")
    kw = dict(flags=_abi.FLAG_AUTO_RESET, human_policy=_abi.HUMAN_ORCA)
    if robot == "orca":
        kw.update(robot_policy=_abi.ROBOT_ORCA, robot_safety_space=0.15)
    elif robot == "linear":
        kw.update(robot_policy=_abi.ROBOT_LINEAR)
    else:
        rs = np.random.RandomState(5)
        space = ebc_actions.build_action_space(float(b.robot[0, 7]))
        kw.update(robot_policy=_abi.ROBOT_EXTERNAL, robot_action=space[rs.randint(len(space), size=(K, E))])
        keys = tuple(k for k in keys if k != "robot_action_out")
    og = g.step_k(K, keys, **kw)
    oo = o.step_k(K, keys, **kw)
    for k in ("done", "info", "n_rows"):
        np.testing.assert_array_equal(og[k], oo[k], err_msg=k)
    for k in ("reward", "dmin", "dist_to_goal", "robot_action_out"):
        if k in og:
            both_inf = np.isinf(og[k]) & np.isinf(oo[k])
            np.testing.assert_allclose(np.where(both_inf, 0, og[k]), np.where(both_inf, 0, oo[k]), atol=1e-9, rtol=0, err_msg=k)
    for k in ("state_rotated", "obs_rotated"):
        np.testing.assert_allclose(og[k], oo[k], atol=1e-5, rtol=1e-5, err_msg=k)
    assert int(oo["done"].sum()) > E  # restarts happened inside the window
    sg, so = g.get_state(), o.get_state()
    for k in sg:
        np.testing.assert_allclose(sg[k], so[k], atol=1e-9, rtol=0, err_msg=k)
    # the device form writes the same into caller-owned tensors
    import torch
    g.reset(b)
    g.use_torch_stream()
    outs = g.alloc_step_k_outputs(K, ("reward", "done", "state_rotated"))
    kw2 = dict(kw)
    if "robot_action" in kw2:
        kw2["robot_action"] = torch.tensor(kw2["robot_action"], dtype=torch.float64, device="cuda:0")
    g.step_k_device(outs, K, **kw2)
    g.synchronize()
    np.testing.assert_array_equal(outs["done"].cpu().numpy(), oo["done"])
    np.testing.assert_allclose(outs["reward"].cpu().numpy(), oo["reward"], atol=1e-9, rtol=0)
    np.testing.assert_allclose(outs["state_rotated"].cpu().numpy(), oo["state_rotated"], atol=1e-5, rtol=1e-5)


def test_mailbox_fault_is_reported_once_and_reset_rearms():
    """The give-up path of the one-launch step (ebc_kernels.h mailbox_wait / EBC_SPIN_LIMIT), never taken in a
    healthy run, exercised ONCE with the test build: a withheld hand-off ends in EBC_ERR_DEVICE instead of a
    hang, is reported once, the handle refuses steps until ebc_reset, and then matches the oracle again."""
    import subprocess
    import sys
    lib = os.path.join(os.path.dirname(GOLDEN), "..", "eb-cadrl_amd", "lib", "libebcsim_fault.so")
    if not os.path.exists(lib):
        pytest.fail("libebcsim_fault.so is not built (make -C eb-cadrl_amd/csrc fault; __graft_entry__.build() does)")
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(GOLDEN), "fault_driver.py")],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["healthy_steps"] == 3
    assert out["first"] == _abi.ERR_DEVICE            # surfaced by the call that ran the broken step
    assert out["sync_after"] == _abi.OK               # reported once, flag cleared
    assert out["step_while_faulted"] == _abi.ERR_STATE
    assert out["after_reset_max_err"] <= 1e-5 and out["sync_end"] == _abi.OK


def test_device_resident_step_matches_host_step():
    import torch
    z = load("traj_a5_linear_orcasub")
    meta = json.loads(str(z["meta"]))
    params = params_of(z)
    E = 128
    b, _ = _random_batch(_config_text(meta), [13000 + e for e in range(E)])
    g1 = _env(params, E, b.N, b.S)
    g2 = _env(params, E, b.N, b.S)
    g1.reset(b)
    g2.reset(b)
    g2.use_torch_stream()
    outs = g2.alloc_step_outputs(("reward", "done", "info", "obs_rotated", "dmin"))
    for t in range(20):
        h = g1.step(human_policy=_abi.HUMAN_ORCA, robot_policy=_abi.ROBOT_LINEAR)
        g2.step_device(outs, human_policy=_abi.HUMAN_ORCA, robot_policy=_abi.ROBOT_LINEAR)
        torch.cuda.synchronize()
        for k in outs:
            np.testing.assert_array_equal(outs[k].cpu().numpy(), h[k], err_msg=k)


def _synthetic_batch(rs, E, N, S, n_lo=0, walls=True):
    """Random scenes built directly as SoA (no generator): ragged humans incl. empty envs,
    random static rows and a few occupied grid cells."""
    n = rs.randint(n_lo, N + 1, size=E).astype(np.int32)
    f = lambda *s: np.zeros(s)  # noqa: E731
    b = ebc_scene.SceneBatch(E, N, S, n, f(E, N), f(E, N), f(E, N), f(E, N), f(E, N), f(E, N),
                             f(E, N), f(E, N), np.zeros((E, N), np.uint8),
                             rs.randint(0, S + 1, size=E).astype(np.int32) if S else np.zeros(E, np.int32),
                             f(E, max(S, 1)), f(E, max(S, 1)), f(E, max(S, 1)), None, f(E, 9))
    for e in range(E):
        k = n[e]
        b.px[e, :k] = rs.uniform(-4, 4, k); b.py[e, :k] = rs.uniform(-4, 4, k)
        b.vx[e, :k] = rs.uniform(-0.5, 0.5, k); b.vy[e, :k] = rs.uniform(-0.5, 0.5, k)
        b.gx[e, :k] = rs.uniform(-4, 4, k); b.gy[e, :k] = rs.uniform(-4, 4, k)
        b.radius[e, :k] = rs.uniform(0.1, 0.5, k); b.v_pref[e, :k] = rs.uniform(0.3, 1.2, k)
        b.type[e, :k] = np.sort(rs.randint(0, 3, k))
        m = b.n_static[e]
        b.spx[e, :m] = rs.uniform(-4, 4, m); b.spy[e, :m] = rs.uniform(-4, 4, m)
        b.sradius[e, :m] = rs.uniform(0.3, 0.8, m)
        b.robot[e] = [rs.uniform(-1, 1), -3.0, 0, 0, 0.3, 0.0, 3.0, 0.7, np.pi / 2]
    if walls:
        grid = np.ones((E, 90, 90))
        for e in range(E):
            for _ in range(3):
                x, y = rs.randint(5, 80, 2)
                grid[e, x:x + rs.randint(1, 12), y:y + rs.randint(1, 12)] = 0
        b.grid = np.stack([ebc_scene.pack_grid(g) for g in grid])
    return b


EDGE = [
    # name, N, S, params overrides, robot kinematics, border
    ("empty-and-ragged-N7", 7, 3, {}, "holonomic", None),
    ("single-human", 1, 0, {}, "holonomic", None),
    ("visible-robot", 6, 2, {"robot_visible": 1}, "holonomic", None),
    ("unicycle-rotation-penalty", 5, 0, {"rotation_penalty_factor": -0.004}, "unicycle", None),
    ("border", 5, 2, {}, "holonomic", [-3.5, 3.5, -3.2, 3.4]),
    ("group-of-32-N24", 24, 6, {}, "holonomic", None),
    ("visible-robot-N33-others", 32, 0, {"robot_visible": 1}, "holonomic", None),
]


@pytest.mark.parametrize("name,N,S,over,kin,border", EDGE)
def test_edge_cases_vs_oracle(name, N, S, over, kin, border):
    from oracle import oracle
    params = params_of(load("traj_n10_walls_t17_orcasub"))
    for k, v in over.items():
        setattr(params, k, v)
    params.robot_kinematics = _abi.HOLONOMIC if kin == "holonomic" else _abi.UNICYCLE
    params.rotate_unicycle = int(kin == "unicycle")
    import zlib
    rs = np.random.RandomState(zlib.crc32(name.encode()) % 2 ** 31)
    E = 70
    b = _synthetic_batch(rs, E, N, S)
    g = _env(params, E, N, S)
    o = oracle.OracleEnv(params, E, N, S)
    g.reset(b)
    o.reset(b)
    if kin == "holonomic":
        space = ebc_actions.build_action_space(0.7)
    else:
        space = ebc_actions.build_action_space(0.7, "unicycle")
    for t in range(25):
        act = space[rs.randint(len(space), size=E)]
        if t % 6 == 0:
            lg = g.lookahead(space, human_policy=_abi.HUMAN_ORCA, border=border)
            lo = o.lookahead(space, human_policy=_abi.HUMAN_ORCA, border=border)
            np.testing.assert_array_equal(lg["info"], lo["info"])
            np.testing.assert_allclose(lg["reward"], lo["reward"], atol=1e-9)
            np.testing.assert_allclose(lg["rows_rotated"], lo["rows_rotated"], atol=1e-5, rtol=1e-5)
        _compare_step(g.step(robot_action=act, human_policy=_abi.HUMAN_ORCA, border=border,
                             flags=_abi.FLAG_AUTO_RESET),
                      o.step(robot_action=act, human_policy=_abi.HUMAN_ORCA, border=border,
                             flags=_abi.FLAG_AUTO_RESET), "%s step %d" % (name, t))
    sg, so = g.get_state(), o.get_state()
    for k in sg:
        np.testing.assert_allclose(sg[k], so[k], atol=1e-9, rtol=0, err_msg=k)


def test_maximum_actions_and_rows():
    """128 actions x (33 + 95 = 128) rows: the limits ebc_create / ebc_lookahead accept."""
    from oracle import oracle
    params = params_of(load("traj_a5_linear_orcasub"))
    rs = np.random.RandomState(5)
    E, N, S = 6, 33, 95
    b = _synthetic_batch(rs, E, N, S, n_lo=30, walls=False)
    g = _env(params, E, N, S)
    o = oracle.OracleEnv(params, E, N, S)
    g.reset(b)
    o.reset(b)
    space = rs.uniform(-0.7, 0.7, size=(128, 2))
    lg = g.lookahead(space, human_policy=_abi.HUMAN_ORCA)
    lo = o.lookahead(space, human_policy=_abi.HUMAN_ORCA)
    np.testing.assert_array_equal(lg["info"], lo["info"])
    np.testing.assert_allclose(lg["rows_rotated"], lo["rows_rotated"], atol=1e-5, rtol=1e-5)
    _compare_step(g.step(robot_action=space[:E], human_policy=_abi.HUMAN_CACHED),
                  o.step(robot_action=space[:E], human_policy=_abi.HUMAN_ORCA), "max sizes")


def test_orca_step_roles_without_rows_and_wide_envs():
    """The one-launch ORCA step (ENV / ORCA / ROWS / STATE roles meeting through mailbox words):
    without observation outputs the ROWS role is absent and its mailboxes must stay empty for a
    later step that has rows; an env with more rows than a wave has lanes takes several passes."""
    import torch
    from oracle import oracle
    params = params_of(load("traj_n10_walls_t17_orcasub"))
    rs = np.random.RandomState(77)
    E, N, S = 50, 7, 3
    b = _synthetic_batch(rs, E, N, S)
    g = _env(params, E, N, S)
    o = oracle.OracleEnv(params, E, N, S)
    g.reset(b)
    o.reset(b)
    g.use_torch_stream()
    slim = g.alloc_step_outputs(("reward", "done"))
    full = g.alloc_step_outputs(("reward", "done", "info", "ob", "obs_rotated", "human_action"))
    fl = _abi.FLAG_AUTO_RESET
    for t in range(24):
        outs = slim if (t // 4) % 2 == 0 else full  # 4 steps without rows, 4 with, ...
        g.step_device(outs, human_policy=_abi.HUMAN_ORCA, robot_policy=_abi.ROBOT_LINEAR, flags=fl)
        ref = o.step(human_policy=_abi.HUMAN_ORCA, robot_policy=_abi.ROBOT_LINEAR, flags=fl)
        g.synchronize()  # also reports a mailbox timeout
        np.testing.assert_array_equal(outs["done"].cpu().numpy(), ref["done"], err_msg="step %d" % t)
        np.testing.assert_allclose(outs["reward"].cpu().numpy(), ref["reward"], atol=1e-9, rtol=0)
        if outs is full:
            np.testing.assert_allclose(outs["ob"].cpu().numpy(), ref["ob"], atol=1e-9, rtol=0, err_msg="step %d" % t)
            np.testing.assert_allclose(outs["human_action"].cpu().numpy(), ref["human_action"], atol=1e-9, rtol=0)
            np.testing.assert_allclose(outs["obs_rotated"].cpu().numpy(), ref["obs_rotated"], atol=1e-5, rtol=1e-5)
    sg, so = g.get_state(), o.get_state()
    for k in sg:
        np.testing.assert_allclose(sg[k], so[k], atol=1e-9, rtol=0, err_msg=k)
    # 33 + 95 = 128 rows per env: two passes of the ROWS lanes, one env per wave
    E, N, S = 6, 33, 95
    b = _synthetic_batch(rs, E, N, S, n_lo=28, walls=False)
    g = _env(params, E, N, S)
    o = oracle.OracleEnv(params, E, N, S)
    g.reset(b)
    o.reset(b)
    act = rs.uniform(-0.7, 0.7, size=(E, 2))
    for t in range(4):
        _compare_step(g.step(robot_action=act, human_policy=_abi.HUMAN_ORCA, flags=fl),
                      o.step(robot_action=act, human_policy=_abi.HUMAN_ORCA, flags=fl), "wide step %d" % t)


def test_observe_is_the_returned_observation():
    """ebc_observe of the current state = the ob / rotated rows the last step returned, and =
    the oracle's; right after reset it is env.reset()'s observation."""
    from oracle import oracle
    z = load("traj_n10_walls_t17_orcasub")
    meta = json.loads(str(z["meta"]))
    params = params_of(z)
    E = 33
    b, _ = _random_batch(_config_text(meta), [15000 + e for e in range(E)])
    g = _env(params, E, b.N, b.S)
    o = oracle.OracleEnv(params, E, b.N, b.S)
    g.reset(b)
    o.reset(b)
    ob_g, obs_g = g.observe()
    ob_o, obs_o = o.observe()
    np.testing.assert_array_equal(ob_g, ob_o)
    np.testing.assert_allclose(obs_g, obs_o, atol=1e-5, rtol=1e-5)
    np.testing.assert_array_equal(ob_g[:, :b.N, 0], b.px)
    for t in range(5):
        out = g.step(human_policy=_abi.HUMAN_ORCA, robot_policy=_abi.ROBOT_LINEAR)
        o.step(human_policy=_abi.HUMAN_ORCA, robot_policy=_abi.ROBOT_LINEAR)
        ob_g, obs_g = g.observe()
        np.testing.assert_array_equal(ob_g, out["ob"])
        np.testing.assert_array_equal(obs_g, out["obs_rotated"])
        np.testing.assert_allclose(obs_g, o.observe()[1], atol=1e-5, rtol=1e-5)


def test_scene_pool_auto_reset():
    """Device-resident scene pool: 3 E host-generated scenes with ragged human counts, different
    goals / static rows / maps; every terminal env restarts from its next pool scene on the
    device.  300 steps so that envs go through several episodes; every output every step against
    the oracle, and the state after a restart equals the pool scene."""
    from oracle import oracle
    params = params_of(load("traj_n10_walls_t17_orcasub"))
    params.time_limit = 6  # short episodes: many restarts
    rs = np.random.RandomState(77)
    E, N, S = 40, 8, 4
    first = _synthetic_batch(rs, E, N, S, n_lo=2)
    pool = _synthetic_batch(rs, 3 * E, N, S, n_lo=1)
    g = _env(params, E, N, S)
    o = oracle.OracleEnv(params, E, N, S)
    for env in (g, o):
        env.reset(first)
        env.set_scene_pool(pool)
    restarts = 0
    seen_scene = np.zeros(E, int)
    for t in range(300):
        og = g.step(human_policy=_abi.HUMAN_ORCA, robot_policy=_abi.ROBOT_LINEAR, flags=_abi.FLAG_AUTO_RESET)
        oo = o.step(human_policy=_abi.HUMAN_ORCA, robot_policy=_abi.ROBOT_LINEAR, flags=_abi.FLAG_AUTO_RESET)
        _compare_step(og, oo, "pool step %d" % t)
        if og["done"].any() and t % 7 == 0:
            st = g.get_state()
            for e in np.where(og["done"])[0]:
                # after a restart the env holds a pool scene at time 0
                assert st["global_time"][e] == 0
                k = int(st["n_humans"][e])
                cands = [c for c in range(e, 3 * E, E) if pool.n_humans[c] == k and
                         np.array_equal(pool.px[c, :k], st["px"][e, :k])]
                assert cands, (t, e)
        restarts += int(og["done"].sum())
    assert restarts > 3 * E
    sg, so = g.get_state(), o.get_state()
    for k in sg:
        np.testing.assert_allclose(sg[k], so[k], atol=1e-9, rtol=0, err_msg=k)
    ob_g, obs_g = g.observe()
    ob_o, obs_o = o.observe()
    np.testing.assert_array_equal(ob_g, ob_o)


def test_scene_pool_reinstalled_on_running_episodes():
    """ebc_set_scene_pool while envs are in the middle of episodes on scenes of the old pool — what run_training does
    every few steps with device-generated scenes: a smaller pool, then a larger one; every output every step against
    the oracle (whose own map bookkeeping is checked in test_oracle_golden.py)."""
    from helpers import pool_reinstall_run
    from oracle import oracle
    for phase, t, (og, oo) in pool_reinstall_run([_env, lambda p, E, N, S: oracle.OracleEnv(p, E, N, S)]):
        if phase >= 0:
            _compare_step(og, oo, "pool %d step %d" % (phase, t))
        else:
            for k in og:
                np.testing.assert_allclose(og[k], oo[k], atol=1e-9, rtol=0, err_msg=k)


@pytest.mark.parametrize("fixture,E,steps,safety", [
    ("traj_a5_il_orcasub", 200, 60, 0.15),          # 5 rows: 5-lane groups
    ("traj_n10_walls_il_orcasub", 150, 90, 0.15),   # 18 rows (ragged static rows): 21-lane groups
    ("traj_a3b3s2_scripted_orcasub", 77, 50, 0.0),  # visible-robot style safety 0
])
def test_robot_orca_rollouts_vs_oracle(fixture, E, steps, safety):
    """The imitation-learning demonstrator on the device (ebc_robot_orca -> ebc_step) against the
    oracle on seeded random scenes, restarts included: the robot's ORCA action every step, bit for
    bit, and the episode it produces; host buffers and device-resident tensors."""
    import torch
    from oracle import oracle
    z = load(fixture)
    params = params_of(z)
    b, cfg = _random_batch(_config_text(json.loads(str(z["meta"]))), [5000 + e for e in range(E)])
    g = _env(params, E, b.N, b.S)
    o = oracle.OracleEnv(params, E, b.N, b.S)
    g.reset(b)
    o.reset(b)
    dev_act = torch.zeros((E, 2), dtype=torch.float64, device="cuda")
    kinds = set()
    for t in range(steps):
        ag, ao = g.robot_orca(safety), o.robot_orca(safety)
        np.testing.assert_array_equal(ag, ao, err_msg="robot ORCA step %d" % t)
        g.robot_orca_device(dev_act, safety)
        g.synchronize()
        np.testing.assert_array_equal(dev_act.cpu().numpy(), ao)
        og = g.step(robot_action=ag, human_policy=_abi.HUMAN_ORCA, flags=_abi.FLAG_AUTO_RESET)
        oo = o.step(robot_action=ao, human_policy=_abi.HUMAN_ORCA, flags=_abi.FLAG_AUTO_RESET)
        _compare_step(og, oo, "%s step %d" % (fixture, t))
        kinds.update(og["info"].tolist())
    assert len(kinds) >= 2


@pytest.mark.gpu
def test_step_on_a_capturing_stream_is_refused():
    """A step's launch carries per-call state (the double-buffered robot, the mailboxes' launch counter): replayed from
    a HIP graph it would run with the capture's arguments.  ebc_step on a stream under capture must return
    EBC_ERR_UNSUPPORTED, record nothing, and leave the handle usable afterwards."""
    import torch
    from ebcsim import _capi
    from ebcsim.batched import BatchedEnv
    import bench
    params, batch = bench.build_batch("metric", 64, 0)
    env = BatchedEnv(params, 64, batch.N, batch.S)
    env.reset(batch)
    outs = env.alloc_step_outputs(("reward", "done"))
    act = torch.zeros((64, 2), dtype=torch.float64, device="cuda:0")
    side = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        env.use_torch_stream()
        env.step_device(outs, robot_action=act, human_policy=_abi.HUMAN_ORCA)
        side.synchronize()
        graph.capture_begin()
        try:
            with pytest.raises(_capi.EbcError, match="captured"):
                env.step_device(outs, robot_action=act, human_policy=_abi.HUMAN_ORCA)
        finally:
            graph.capture_end()
        env.step_device(outs, robot_action=act, human_policy=_abi.HUMAN_ORCA)
        side.synchronize()
    assert bool(torch.isfinite(outs["reward"]).all())
