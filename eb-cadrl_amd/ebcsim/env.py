"""Single-env object facade with the reference's gym-style surface.

`EntityBasedCollisionAvoidance` here answers the calls rl/train.py, rl/test.py and the
reference's tests make on simulator/env.py:19 — configure / set_robot / reset / step /
onestep_lookahead and the attributes they read — but every step is one call through the C ABI
(BatchedEnv with E = 1).  It returns the reference's value objects (ObservableState, Info
subclasses) rebuilt from the kernel outputs.  Batched callers should use BatchedEnv directly;
this class exists so legacy single-env drivers keep working unchanged.
"""
import numpy as np

from . import _abi, config as ebc_config, info as ebc_info, scene as ebc_scene
from .action import ActionRot, ActionXY
from .agents import AgentType, HumanView, KIND
from .policy import DeviceHumanPolicy
from .state import ObservableState

_POLICY_CODE = {"orca": _abi.HUMAN_ORCA, "linear": _abi.HUMAN_LINEAR}


class _SceneFacade(object):
    """What callers read from env.scene (simulator/scene/scene_generator.py:19-72)."""

    def __init__(self, cfg):
        self.cfg = cfg
        self.case_counter = {"train": 0, "test": 0, "val": 0}
        self.case_size = {"train": np.iinfo(np.uint32).max - 2000, "val": cfg.val_size,
                          "test": cfg.test_size}
        for k in ("train_val_sim_adult", "test_sim_adult", "train_val_sim_bicycle",
                  "test_sim_bicycle", "train_val_sim_children", "test_sim_children", "square_width",
                  "circle_radius", "randomize_attributes", "map_resolution", "map_size_m",
                  "adult_num", "bicycle_num", "children_num", "discomfort_dist"):
            setattr(self, k, getattr(cfg, k))
        self.robot = None
        self.adults, self.bicycles, self.children = [], [], []
        self.map = None
        self.obstacle_vertices = []
        self.static_obstacles_as_pedestrians = []
        self.current = None  # ebcsim.scene.Scene of the running episode

    def set_robot(self, robot):
        self.robot = robot


class EntityBasedCollisionAvoidance(object):
    metadata = {"render.modes": ["adult"]}
    PHASES = ["train", "val", "test"]

    def __init__(self, backend_factory=None, device=0):
        """backend_factory(params, n_envs, max_humans, max_static) -> object with the BatchedEnv
        call surface; default = the HIP library (tests may inject a checker)."""
        self.name = "EntityBasedCollisionAvoidance"
        self._factory = backend_factory
        self._device = device
        self._backend = None
        self._backend_key = None
        self.time_step = None
        self.time_limit = None
        self.robot = None
        self.global_time = None
        self.states = None
        self.action_values = None
        self.attention_weights = None
        self.phase = None
        self._humans_cached = False
        self.backend_calls = 0      # launches-level calls through the C ABI (tests count them)
        self.orca_evaluations = 0   # of which evaluated ORCA for the humans

    # ---------------------------------------------------------------- configuration
    def configure(self, config):
        """simulator/env.py:58-87"""
        self.config = config
        self._scene_cfg = ebc_scene.SceneConfig.from_config(config)
        self.scene = _SceneFacade(self._scene_cfg)
        self.time_step = config.getfloat("env", "time_step")
        self.time_limit = config.getint("env", "time_limit")
        self.case_capacity = {"train": np.iinfo(np.uint32).max - 2000, "val": 1000, "test": 1000}
        self.use_grid_map = config.getboolean("map", "use_grid_map")  # raises without [map], like env.py:79
        if self.use_grid_map:
            raise NotImplementedError("use_grid_map = true (cv2 local maps) is set by no shipped config")
        self.angular_map_max_range = config.getfloat("map", "angular_map_max_range")
        self.angular_map_dim = config.getint("map", "angular_map_dim")
        self.angular_map_min_angle = config.getfloat("map", "angle_min") * np.pi
        self.angular_map_max_angle = config.getfloat("map", "angle_max") * np.pi
        self.local_maps_angular = None
        kinds = set()
        for section, count in (("adults", self._scene_cfg.adult_num),
                               ("bicycles", self._scene_cfg.bicycle_num),
                               ("children", self._scene_cfg.children_num)):
            if count and config.has_section(section):
                kinds.add(config.get(section, "policy"))
        if len(kinds) > 1 or (kinds and not kinds <= set(_POLICY_CODE)):
            raise NotImplementedError("human policies %s: one of orca / linear for all humans" % sorted(kinds))
        self._human_policy_name = kinds.pop() if kinds else "orca"
        self._human_policy = _POLICY_CODE[self._human_policy_name]

    def set_robot(self, robot):
        self.robot = robot
        self.scene.set_robot(robot)

    def _ensure_backend(self, n_humans, n_static):
        kin = self.robot.kinematics or "holonomic"
        # the width of a rotated row follows the robot's policy (rl/policy/sarl.py:103-110)
        typed = bool(getattr(self.robot.policy, "with_agent_type", False))
        key = (n_humans, n_static, kin, typed)
        if key != self._backend_key:
            params = ebc_config.params_from_config(self.config, robot_kinematics=kin)
            params.with_agent_type = int(typed)
            if self._backend is not None and hasattr(self._backend, "close"):
                self._backend.close()
            if self._factory is not None:
                self._backend = self._factory(params, 1, n_humans, n_static)
            else:
                from .batched import BatchedEnv
                self._backend = BatchedEnv(params, 1, n_humans, n_static, device=self._device)
            self._params = params
            self._backend_key = key
        return self._backend

    def get_local_map_angular(self, ob, normalize=True, append=True):
        """simulator/env.py:570-628, on the host (ebcsim/local_map.py): no policy on the path reads it."""
        from .local_map import angular_map
        m = angular_map(self.scene.obstacle_vertices, ob.px, ob.py, ob.radius, ob.theta,
                        self.angular_map_max_range, self.angular_map_dim, self.angular_map_min_angle,
                        self.angular_map_max_angle, normalize)
        if append:
            self.local_maps_angular.append(m)
        return m

    # ---------------------------------------------------------------- reset
    def reset_times(self, phase):
        """simulator/env.py:106-126"""
        many = phase == "test" or self.robot.policy.multiagent_training
        sc = self.scene
        self.adult_times = [0] * (sc.adult_num if many else 1)
        self.bicycle_times = [0] * (sc.bicycle_num if many else 1)
        self.children_times = [0] * (sc.children_num if many else 1)

    def reset(self, phase="test", test_case=None, imitation_learning=False, compute_local_map=True,
              save_scene_path=None, load_scene_path=None, scene_number=None):
        """simulator/env.py:128-205"""
        if self.robot is None:
            raise AttributeError("robot has to be set!")
        self.phase = phase
        assert phase in self.PHASES, "phase must be one of {}".format(self.PHASES)
        sc = self.scene
        if test_case is not None:
            sc.case_counter[phase] = test_case
        self.global_time = 0
        R = sc.circle_radius
        self.robot.set(0, -R, 0, R, 0, 0, np.pi / 2)
        cfg = self._scene_cfg
        cfg.robot.radius, cfg.robot.v_pref = self.robot.radius, self.robot.v_pref
        if load_scene_path is not None:
            scene = ebc_scene.load_scene(cfg, load_scene_path)
        else:
            seed = scene_number if scene_number is not None else (
                ebc_scene.COUNTER_OFFSET[phase] + sc.case_counter[phase])
            multi = bool(getattr(self.robot.policy, "multiagent_training", True))
            scene = ebc_scene.generate_scene(cfg, seed, phase, multi)
        sc.case_counter[phase] = (sc.case_counter[phase] + 1) % sc.case_size[phase]
        if save_scene_path is not None and load_scene_path is None:
            ebc_scene.save_scene(scene, save_scene_path)
        sc.current = scene
        sc.map = scene.grid
        sc.obstacle_vertices = scene.obstacle_vertices
        sc.adult_num = sum(h.type == _abi.ADULT for h in scene.humans)
        sc.bicycle_num = sum(h.type == _abi.BICYCLE for h in scene.humans)
        sc.children_num = sum(h.type == _abi.CHILD for h in scene.humans)
        self.reset_times(phase)

        marker = DeviceHumanPolicy(self._human_policy_name.upper(), self._human_policy)
        self._humans = []
        sc.adults, sc.bicycles, sc.children = [], [], []
        for h in scene.humans:
            v = HumanView(AgentType(int(h.type)), marker)
            v.set(h.px, h.py, h.gx, h.gy, h.vx, h.vy, h.theta, h.radius, h.v_pref)
            v.time_step = self.time_step
            self._humans.append(v)
            getattr(sc, KIND[int(h.type)]).append(v)
        self.robot.time_step = self.time_step
        if self.robot.policy is not None:
            self.robot.policy.time_step = self.time_step
        sc.static_obstacles_as_pedestrians = [
            ObservableState(px, py, 0, 0, r, AgentType.ADULT_STATIC) for px, py, r in scene.static_rows]

        batch = ebc_scene.SceneBatch.from_scenes([scene])
        backend = self._ensure_backend(batch.N, batch.S)
        backend.reset(batch)
        self._humans_cached = False

        self.states = list()
        self.local_maps_angular = list()
        if hasattr(self.robot.policy, "action_values"):
            self.action_values = list()
        if hasattr(self.robot.policy, "get_attention_weights"):
            self.attention_weights = list()
        ob = [h.get_observable_state() for h in self._humans] + sc.static_obstacles_as_pedestrians
        local_map = self.get_local_map_angular(self.robot.get_full_state()) if compute_local_map else None
        if self.robot.policy is not None and self.robot.policy.name == "ORCA":
            return ob, sc.obstacle_vertices, local_map
        return ob, local_map

    # ---------------------------------------------------------------- step / look-ahead
    def _action_row(self, action):
        if self.robot.kinematics == "holonomic":
            assert isinstance(action, ActionXY)
            return np.array([[action.vx, action.vy]], dtype=np.float64)
        assert isinstance(action, ActionRot)
        return np.array([[action.v, action.r]], dtype=np.float64)

    def _border(self, border):
        return None if border is None else [border[0][0], border[0][1], border[1][0], border[1][1]]

    def _rows_to_ob(self, rows):
        n = len(self._humans)
        ob = [ObservableState(r[0], r[1], r[2], r[3], r[4], h.agent_type)
              for r, h in zip(rows[:n].tolist(), self._humans)]
        return ob + self.scene.static_obstacles_as_pedestrians

    def _info(self, code, dist_to_goal, dmin):
        return ebc_info.from_code(code, dist_to_goal, dmin, list(self._params.discomfort_dist))

    def onestep_lookahead(self, action):
        """simulator/env.py:207-209 (the reference also recomputes the local map here and drops it)"""
        ob, _, reward, done, info = self.step(action, update=False, compute_local_map=False)
        return ob, reward, done, info

    def _humans_for_lookahead(self):
        """ORCA's answer depends on the state only, not on the robot's candidate action
        (simulator/env.py:395-419 runs the humans before it looks at the action): the first query of a
        state computes the humans' velocities and leaves them cached in the backend, every later query
        of the same state — the other 80 of MultiHumanRL.predict's loop, and the real step that follows
        — reads them (EBC_HUMAN_CACHED)."""
        if self._human_policy != _abi.HUMAN_ORCA:
            return self._human_policy
        if self._humans_cached:
            return _abi.HUMAN_CACHED
        self._humans_cached = True
        self.orca_evaluations += 1
        return _abi.HUMAN_ORCA

    def lookahead_all(self, actions):
        """All candidate actions in one launch: what MultiHumanRL.predict's loop asks for
        (rl/policy/multi_human_rl.py:38-61).  Returns the backend's arrays for env 0, plus `n_rows`
        (humans + static rows of the scene: the rows the reference would have built)."""
        acts = np.array([[a[0], a[1]] for a in actions], dtype=np.float64)
        out = self._backend.lookahead(acts, human_policy=self._humans_for_lookahead())
        self.backend_calls += 1
        out = {k: v[0] for k, v in out.items()}
        out["n_rows"] = len(self._humans) + len(self.scene.static_obstacles_as_pedestrians)
        return out

    def observe_rotated(self):
        """The rotated joint state of the current state, [rows, T] float32: MultiHumanRL.transform
        (rl/policy/multi_human_rl.py:128-149) of JointState(robot, last observation)."""
        n = len(self._humans) + len(self.scene.static_obstacles_as_pedestrians)
        return self._backend.observe()[1][0, :n].copy()

    def step(self, action, update=True, compute_local_map=True, border=None):
        """simulator/env.py:388-466"""
        b = self._backend
        act = self._action_row(action)
        if not update:
            out = b.lookahead(act, human_policy=self._humans_for_lookahead(), border=self._border(border),
                              want_rows=False)
            self.backend_calls += 1
            ob = self._rows_to_ob(out["next_ob"][0])
            nx, ny = self.robot.compute_position(action, self.time_step)
            dg = float(np.linalg.norm(np.array((nx, ny)) - np.array(self.robot.get_goal_position())))
            info = self._info(out["info"][0, 0], dg, out["dmin"][0, 0])
            # env.py:460-465: the map of the CURRENT robot state, also in look-ahead
            local_map = self.get_local_map_angular(self.robot.get_full_state()) if compute_local_map else None
            return ob, local_map, float(out["reward"][0, 0]), bool(out["done"][0, 0]), info

        # render history: full states before the update (env.py:344-351)
        sc = self.scene
        self.states.append([self.robot.get_full_state(), [h.get_full_state() for h in sc.adults],
                            [h.get_full_state() for h in sc.bicycles],
                            [h.get_full_state() for h in sc.children]])
        if hasattr(self.robot.policy, "action_values"):
            self.action_values.append(self.robot.policy.action_values)
        if hasattr(self.robot.policy, "get_attention_weights"):
            self.attention_weights.append(self.robot.policy.get_attention_weights())

        out = b.step(robot_action=act, human_policy=self._humans_for_lookahead(), border=self._border(border))
        self.backend_calls += 1
        self._humans_cached = False  # the state has moved on
        st = b.get_state()
        r = st["robot"][0]
        self.robot.px, self.robot.py, self.robot.vx, self.robot.vy = r[0], r[1], r[2], r[3]
        self.robot.theta = r[8]
        for i, h in enumerate(self._humans):
            h.px, h.py, h.vx, h.vy = st["px"][0, i], st["py"][0, i], st["vx"][0, i], st["vy"][0, i]
        self.global_time = float(st["global_time"][0])
        arr = st["arrival_time"][0]
        k = 0
        for times, group in ((self.adult_times, sc.adults), (self.bicycle_times, sc.bicycles),
                             (self.children_times, sc.children)):
            for j in range(min(len(times), len(group))):
                times[j] = float(arr[k + j]) if arr[k + j] != 0 else 0
            k += len(group)
        ob = self._rows_to_ob(out["ob"][0])
        info = self._info(out["info"][0], out["dist_to_goal"][0], out["dmin"][0])
        local_map = self.get_local_map_angular(self.robot.get_full_state()) if compute_local_map else None
        return ob, local_map, float(out["reward"][0]), bool(out["done"][0]), info

    def render(self, mode="adult", output_file=None):
        raise NotImplementedError("rendering (simulator/utils/render.py) is outside the accelerated "
                                  "path; env.states keeps the history it would consume")


ENV_ID = "EntityBasedCollisionAvoidance-v0"


def make(env_id=ENV_ID, **kwargs):
    """gym.make stand-in (simulator/__init__.py:4-7) that needs no gym."""
    if env_id != ENV_ID:
        raise KeyError(env_id)
    return EntityBasedCollisionAvoidance(**kwargs)


def _policy_factory():
    """rl/policy/policy_factory.py:8-12 extends the simulator's table with the learnt policies.  When the
    caller's `rl` package is importable its table is used as it is (its SARL then drives this env through
    `onestep_lookahead`); otherwise `sarl` is this package's own policy (one sweep + one batched forward)."""
    try:
        from rl.policy.policy_factory import policy_factory as table
        return table
    except ImportError:
        from .policy import policy_factory
        from .rl_policy import SARL
        table = dict(policy_factory)
        table["sarl"] = SARL
        return table


def configure_env_policy_robot(env_config_path, policy_config_path=None, model_path=None, phase="test",
                               device="cpu", policy="sarl", env_name=ENV_ID, backend_factory=None):
    """simulator/utils/test_utils.py:8-36, same positional order (the reference's tests pass the weight
    file third: tests/test_basic_simulation.py:11-13).  `backend_factory` is this package's test hook."""
    import torch
    from .agents import Robot
    env_config = ebc_config.read_config(env_config_path)
    env = make(env_name, backend_factory=backend_factory)
    env.configure(env_config)
    robot = Robot(env_config, "robot")
    env.set_robot(robot)
    pol = _policy_factory()[policy]()
    if policy_config_path is not None:
        pol.configure(ebc_config.read_config(policy_config_path))
    if model_path is not None:
        pol.get_model().load_state_dict(torch.load(model_path, map_location="cpu"))
    robot.set_policy(pol)
    pol.set_phase(phase)
    pol.set_device(device)
    return env, pol, robot
