"""Shared helpers for the parity tests: fixture loading and scene/params rebuilds."""
import json
import os

import numpy as np

from ebcsim import _abi, config as ebc_config
from ebcsim.scene import SceneBatch, pack_grid

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def params_of(z, key="params"):
    return ebc_config.params_from_dict(json.loads(str(z[key])))


def batch_from_init(z, prefix="init_", copies=1, max_humans=None, max_static=None):
    """SceneBatch from the init_* arrays a trajectory fixture stores (one scene, tiled)."""
    n = len(z[prefix + "px"])
    st = z[prefix + "static"].reshape(-1, 3)
    N = max_humans or n
    S = max_static if max_static is not None else len(st)
    f = lambda *s: np.zeros(s)  # noqa: E731
    b = SceneBatch(copies, N, S, np.full(copies, n, np.int32), f(copies, N), f(copies, N),
                   f(copies, N), f(copies, N), f(copies, N), f(copies, N), f(copies, N),
                   f(copies, N), np.zeros((copies, N), np.uint8),
                   np.full(copies, len(st), np.int32), f(copies, max(S, 1)), f(copies, max(S, 1)),
                   f(copies, max(S, 1)), None, f(copies, 9))
    for k in ("px", "py", "vx", "vy", "gx", "gy", "radius", "v_pref", "type"):
        getattr(b, k)[:, :n] = z[prefix + k]
    if len(st):
        b.spx[:, :len(st)], b.spy[:, :len(st)], b.sradius[:, :len(st)] = st.T
    grid = z[prefix + "grid"]
    if grid.any():
        b.grid = np.repeat(pack_grid(1.0 - grid)[None], copies, axis=0)
    b.robot[:] = z[prefix + "robot"]
    return b


HUMAN_POLICY = {"linear": _abi.HUMAN_LINEAR}


def human_policy_of(z):
    meta = json.loads(str(z["meta"]))
    return _abi.HUMAN_LINEAR if meta["human_policy"] == "linear" else _abi.HUMAN_ORCA


TRAJ_PINNED = ["traj_a5_linear", "traj_a5_scripted", "traj_a3b3s2_scripted", "traj_n10_walls_t17",
               "traj_unicycle_rotpen"]
TRAJ_ORCASUB = ["traj_a5_linear_orcasub", "traj_a5_scripted_orcasub",
                "traj_a3b3s2_scripted_orcasub", "traj_n10_walls_t17_orcasub"]
# the imitation-learning demonstrator: the robot itself on ORCA (rl/train.py:99-143)
TRAJ_IL = ["traj_a5_il_orcasub", "traj_n10_walls_il_orcasub"]


def check_trajectory(env, z, atol=1e-9, rot_atol=1e-5, lookahead=True):
    """Replay fixture `z` on `env` (OracleEnv or BatchedEnv; env 0 is compared; every env of
    the batch gets the same scene and actions) and assert parity step by step."""
    pol = human_policy_of(z)
    E = env.E
    n = len(z["init_px"])
    ns = len(z["init_static"].reshape(-1, 3))
    la_steps = list(z["la_step"]) if ("la_step" in z.files and lookahead) else []
    meta = json.loads(str(z["meta"]))
    robot_orca = meta.get("robot_mode") == "orca"
    for t in range(len(z["action"])):
        if robot_orca:
            # the demonstrator's action (Robot.act -> ORCA.predict) and the state the explorer keeps
            # for the imitation-learning memory (policy.last_state, transformed)
            a = env.robot_orca(meta["il_safety_space"])
            np.testing.assert_allclose(a[0], z["action"][t], atol=atol, rtol=0, err_msg="robot ORCA, step %d" % t)
            assert (a == a[0:1]).all()
            np.testing.assert_allclose(env.observe()[1][0][:n + ns], z["il_state"][t], atol=rot_atol,
                                       rtol=rot_atol, err_msg="IL state, step %d" % t)
        if t in la_steps:
            k = la_steps.index(t)
            la = env.lookahead(z["la_actions"], human_policy=pol)
            np.testing.assert_array_equal(la["done"][0].astype(bool), z["la_done"][k].astype(bool))
            np.testing.assert_array_equal(la["info"][0], z["la_info"][k])
            np.testing.assert_allclose(la["reward"][0], z["la_reward"][k], atol=atol, rtol=0)
            np.testing.assert_allclose(la["next_ob"][0][:n + ns], z["la_next_ob"][k][:, :5],
                                       atol=atol, rtol=0)
            np.testing.assert_allclose(la["rows_rotated"][0][:, :n + ns], z["la_rows"][k],
                                       atol=rot_atol, rtol=rot_atol)
        act = np.tile(z["action"][t], (E, 1))
        out = env.step(robot_action=act, human_policy=pol)
        msg = "step %d" % t
        assert bool(out["done"][0]) == bool(z["done"][t]), msg
        assert int(out["info"][0]) == int(z["info"][t]), msg
        np.testing.assert_allclose(out["reward"][0], z["reward"][t], atol=atol, rtol=0, err_msg=msg)
        np.testing.assert_allclose(out["dmin"][0], z["dmin"][t], atol=atol, rtol=0, err_msg=msg)
        if not np.isnan(z["dist_to_goal"][t]):
            np.testing.assert_allclose(out["dist_to_goal"][0], z["dist_to_goal"][t], atol=atol,
                                       rtol=0, err_msg=msg)
        np.testing.assert_allclose(out["human_action"][0][:n], z["human_action"][t], atol=atol,
                                   rtol=0, err_msg=msg)
        np.testing.assert_allclose(out["ob"][0][:n + ns], z["ob"][t][:, :5], atol=atol, rtol=0,
                                   err_msg=msg)
        np.testing.assert_allclose(out["obs_rotated"][0][:n + ns], z["rot"][t], atol=rot_atol,
                                   rtol=rot_atol, err_msg=msg)
        st = env.get_state()
        np.testing.assert_allclose(st["robot"][0], z["robot"][t], atol=atol, rtol=0, err_msg=msg)
        np.testing.assert_allclose(st["global_time"][0], z["time"][t], atol=1e-12, rtol=0)
        np.testing.assert_allclose(st["arrival_time"][0][:n], z["arrival"][t], atol=1e-12, rtol=0,
                                   err_msg=msg)
        hum = np.stack([st["px"][0][:n], st["py"][0][:n], st["vx"][0][:n], st["vy"][0][:n]], 1)
        np.testing.assert_allclose(hum, z["humans"][t], atol=atol, rtol=0, err_msg=msg)
        if E > 1:  # every replica of the scene must agree with env 0
            for key in ("reward", "info", "obs_rotated"):
                assert (out[key] == out[key][0:1]).all(), msg


def config_text_of(meta):
    """The INI text a trajectory fixture ran with: the scenes fixture's text of the same config file plus the
    overrides recorded in the fixture's meta."""
    import configparser
    import io
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
            buf = io.StringIO()
            cfg.write(buf)
            return buf.getvalue()
    raise KeyError(meta["config"])


def pool_reinstall_run(envs, E=24, steps=(60, 40, 60)):
    """ebc_set_scene_pool on a RUNNING batch (walls: every scene has its own occupancy grid): install 3 E scenes,
    step until envs have restarted from them, replace the pool by a SMALLER one (E / 2 scenes), keep stepping, then
    by a larger one.  Every env of `envs` gets the same calls; yields (phase, step, [outputs per env]) and checks,
    on envs[-1] after each re-installation, that every env still collides against the map of the scene whose
    static rows it holds — the env must not follow the pool's slots."""
    import configparser
    from ebcsim import scene as ebc_scene
    z = load("traj_n10_walls_t17_orcasub")
    meta = json.loads(str(z["meta"]))
    params = params_of(z)
    params.time_limit = 4  # short episodes: many restarts
    cfg = configparser.RawConfigParser()
    cfg.read_string(config_text_of(meta))
    sc = ebc_scene.SceneConfig.from_config(cfg)
    gen = lambda seeds: [ebc_scene.generate_scene(sc, int(s)) for s in seeds]  # noqa: E731
    all_scenes = {}

    def batch(seeds, N=None, S=None):
        b = ebc_scene.SceneBatch.from_scenes(gen(seeds), N, S)
        for c in range(b.n):
            all_scenes[b.spx[c].tobytes() + b.spy[c].tobytes()] = b.grid[c].copy()
        return b
    first = batch(range(7000, 7000 + E), None, 12)
    N, S = first.N, first.S
    pools = [batch(range(7100, 7100 + 3 * E), N, S), batch(range(7300, 7300 + E // 2), N, S),
             batch(range(7400, 7400 + 5 * E), N, S)]
    made = [mk(params, E, N, S) for mk in envs]
    for env in made:
        env.reset(first)
    probe = made[-1]
    restarts = 0
    for phase, pool in enumerate(pools):
        for env in made:
            env.set_scene_pool(pool, stride=E)
        if hasattr(probe, "pool"):  # the oracle: white box
            gs = probe.a["grid_scene"]
            for e in range(E):
                key = probe.a["spx"][e].tobytes() + probe.a["spy"][e].tobytes()
                np.testing.assert_array_equal(probe.pool["grid"][gs[e]], all_scenes[key], err_msg="phase %d env %d" % (phase, e))
        for t in range(steps[phase]):
            outs = [env.step(human_policy=_abi.HUMAN_ORCA, robot_policy=_abi.ROBOT_LINEAR, flags=_abi.FLAG_AUTO_RESET)
                    for env in made]
            restarts += int(outs[-1]["done"].sum())
            yield phase, t, outs
    assert restarts > 4 * E
    yield -1, 0, [env.get_state() for env in made]


class CpuDeviceEnv(object):
    """The *_device call surface of ebcsim.batched.BatchedEnv on CPU torch tensors, computed by the
    oracle: lets the trainer's schedule (ebcsim/train.py: collect, collect_il, run_training — host
    logic over that surface) run under gloo without a GPU.  Test scaffolding only."""

    def __init__(self, params, E, N, S):
        from oracle import oracle
        self._o = oracle.OracleEnv(params, E, N, S)
        self.params, self.E, self.N, self.S, self.R, self.T = params, E, N, S, N + S, self._o.T
        self.device = 0
        self.ragged = False
        self.steps = 0

    def _note(self, scene):
        rows = np.asarray(scene.n_humans, np.int64) + (np.asarray(scene.n_static, np.int64) if self.S else 0)
        self.ragged = bool(self.ragged or (rows < self.R).any())

    def reset(self, scene, env_ids=None):
        self._o.reset(scene, env_ids)
        self._note(scene)

    def set_scene_pool(self, scene, stride=None):
        self._o.set_scene_pool(scene, stride)
        self._note(scene)

    def get_state(self):
        return self._o.get_state()

    def use_torch_stream(self):
        pass

    def synchronize(self):
        pass

    def alloc_step_outputs(self, keys=("reward", "done", "info", "obs_rotated")):
        import torch
        E, N, R, T = self.E, self.N, self.R, self.T
        shapes = dict(reward=((E,), torch.float64), done=((E,), torch.uint8), info=((E,), torch.uint8),
                      dmin=((E, 3), torch.float64), dist_to_goal=((E,), torch.float64),
                      obs_rotated=((E, R, T), torch.float32))
        return {k: torch.zeros(shapes[k][0], dtype=shapes[k][1]) for k in keys}

    def alloc_lookahead_outputs(self, n_actions, keys=("reward", "done", "info", "rows_rotated")):
        import torch
        E, R, T, A = self.E, self.R, self.T, int(n_actions)
        shapes = dict(reward=((E, A), torch.float64), done=((E, A), torch.uint8), info=((E, A), torch.uint8),
                      rows_rotated=((E, A, R, T), torch.float32))
        return {k: torch.zeros(shapes[k][0], dtype=shapes[k][1]) for k in keys}

    def step_device(self, outputs, robot_action=None, human_policy=_abi.HUMAN_ORCA,
                    robot_policy=_abi.ROBOT_EXTERNAL, flags=0):
        import torch
        out = self._o.step(robot_action=None if robot_action is None else robot_action.numpy(),
                           human_policy=human_policy, robot_policy=robot_policy, flags=flags)
        for k, t in outputs.items():
            t.copy_(torch.from_numpy(out[k]))
        self.steps += 1

    def lookahead_device(self, actions, outputs, human_policy=_abi.HUMAN_ORCA, flags=0):
        import torch
        out = self._o.lookahead(actions.numpy(), human_policy=human_policy, flags=flags)
        for k, t in outputs.items():
            t.copy_(torch.from_numpy(out[k]))

    def observe_device(self, obs_rotated):
        import torch
        obs_rotated.copy_(torch.from_numpy(self._o.observe()[1]))

    def row_counts_device(self, n_rows):
        import torch
        n_rows.copy_(torch.from_numpy(self._o.row_counts()))

    def alloc_step_k_outputs(self, K, keys=("reward", "done", "info", "state_rotated")):
        import torch
        E, R, T = self.E, self.R, self.T
        shapes = dict(state_rotated=((E, R, T), torch.float32), n_rows=((E,), torch.int64),
                      robot_action_out=((E, 2), torch.float64), reward=((E,), torch.float64), done=((E,), torch.uint8),
                      info=((E,), torch.uint8), dmin=((E, 3), torch.float64), dist_to_goal=((E,), torch.float64),
                      obs_rotated=((E, R, T), torch.float32))
        return {k: torch.zeros((K,) + shapes[k][0], dtype=shapes[k][1]) for k in keys}

    def step_k_device(self, outputs, K, robot_action=None, human_policy=_abi.HUMAN_ORCA,
                      robot_policy=_abi.ROBOT_LINEAR, flags=0, robot_safety_space=0.0):
        import torch
        out = self._o.step_k(K, tuple(outputs), None if robot_action is None else robot_action.numpy(), human_policy,
                             robot_policy, flags, robot_safety_space)
        for k, t in outputs.items():
            t.copy_(torch.from_numpy(out[k]))
        self.steps += K

    def robot_orca_sim(self, enable=True):
        self._o.robot_orca_sim(enable)

    def robot_orca_device(self, actions, safety_space=0.0):
        import torch
        actions.copy_(torch.from_numpy(self._o.robot_orca(safety_space)))


def il_persistent_episodes(name, copies=1):
    """(z, params, N, S, [SceneBatch per episode]) of an il_persistent_* fixture."""
    z = load(name)
    K = int(z["n_episodes"])
    N = max(len(z["init%d_px" % k]) for k in range(K))
    S = max(len(z["init%d_static" % k].reshape(-1, 3)) for k in range(K))
    return z, params_of(z), N, S, [batch_from_init(z, prefix="init%d_" % k, copies=copies, max_humans=N, max_static=S)
                                    for k in range(K)]


def check_il_persistent(make_env, name, copies=1):
    """Consecutive imitation-learning episodes on ONE ORCA policy object (the reference's Explorer.run_k_episodes on
    its il_policy, rl/train.py:130-133): with the persistent simulator enabled once and env.reset() per episode the
    robot's actions, rewards and terminal classes are the reference's in EVERY episode; without it they are not."""
    z, params, N, S, batches = il_persistent_episodes(name, copies)
    safety = float(z["safety_space"])
    tile = lambda b: b  # noqa: E731
    env = make_env(params, copies, N, S)
    env.robot_orca_sim(True)
    for k, b in enumerate(batches):
        env.reset(tile(b))
        for t in range(len(z["action%d" % k])):
            a = env.robot_orca(safety)
            np.testing.assert_allclose(a[0], z["action%d" % k][t], atol=1e-9, rtol=0, err_msg="episode %d step %d" % (k, t))
            assert (a == a[0:1]).all()
            out = env.step(robot_action=a, human_policy=_abi.HUMAN_ORCA)
            np.testing.assert_allclose(out["reward"][0], z["reward%d" % k][t], atol=1e-9, rtol=0)
            assert int(out["info"][0]) == int(z["info%d" % k][t]), (k, t)
        assert bool(out["done"][0])
    # the fixture has teeth: a fresh simulator per call departs from the reference once the radii have changed
    if not bool(z["sim_rebuilt"][1:].all()):
        env.robot_orca_sim(False)
        differs = False
        for k, b in enumerate(batches[:2]):
            env.reset(tile(b))
            for t in range(len(z["action%d" % k])):
                a = env.robot_orca(safety)
                differs = differs or not np.allclose(a[0], z["action%d" % k][t], atol=1e-9, rtol=0)
                env.step(robot_action=np.tile(z["action%d" % k][t], (copies, 1)), human_policy=_abi.HUMAN_ORCA)
        assert differs
