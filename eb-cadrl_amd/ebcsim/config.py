"""INI files -> EbcParams: the keys env.configure(), Reward.__init__ and the policy
config contribute to the hot path (simulator/env.py:58-87, simulator/utils/reward.py:18-75,
rl/policy/cadrl.py:72-81, rl/policy/sarl.py:103-110)."""
import configparser
import ctypes as C

from . import _abi

_NAN = float("nan")


def read_config(path):
    cfg = configparser.RawConfigParser()
    if not cfg.read(path):
        raise FileNotFoundError(path)
    return cfg


def _opt(cfg, section, key, default=_NAN):
    v = cfg.getfloat(section, key, fallback=None)
    return default if v is None else v


def params_from_config(env_cfg, policy_cfg=None, robot_kinematics=None):
    """env_cfg / policy_cfg: RawConfigParser objects in the reference's schema."""
    p = _abi.default_params()
    p.time_step = env_cfg.getfloat("env", "time_step")
    p.time_limit = env_cfg.getint("env", "time_limit")
    # env.configure itself raises without a [map] section (env.py:79); Reward alone does not
    p.map_size_m = _opt(env_cfg, "map", "map_size_m", 9.0) if env_cfg.has_section("map") else 9.0
    p.map_resolution = _opt(env_cfg, "map", "map_resolution", 0.1) if env_cfg.has_section("map") else 0.1
    p.robot_visible = int(env_cfg.getboolean("robot", "visible"))
    r = "reward"
    p.new_reward = int(env_cfg.getboolean(r, "new_reward", fallback=False))
    p.time_max = _opt(env_cfg, r, "time_max")
    p.max_goal_distance = _opt(env_cfg, r, "max_goal_distance")
    p.time_good = _opt(env_cfg, r, "time_good", 10.0)
    p.success_reward = env_cfg.getfloat(r, "success_reward")
    for i, k in enumerate(("adult", "bicycle", "child", "obstacle")):
        p.collision_penalty[i] = _opt(env_cfg, r, "collision_penalty_" + k)
    dd = env_cfg.getfloat(r, "discomfort_dist")
    df = env_cfg.getfloat(r, "discomfort_penalty_factor")
    for i, k in enumerate(("adult", "bicycle", "child")):
        p.discomfort_dist[i] = _opt(env_cfg, r, "discomfort_dist_" + k, dd)
        p.discomfort_factor[i] = _opt(env_cfg, r, "discomfort_penalty_factor_" + k, df)
    p.rotation_penalty_factor = env_cfg.getfloat(r, "rotation_penalty_factor")
    kin = robot_kinematics
    if policy_cfg is not None:
        if kin is None and policy_cfg.has_option("action_space", "kinematics"):
            kin = policy_cfg.get("action_space", "kinematics")
        if policy_cfg.has_option("sarl", "with_agent_type"):
            p.with_agent_type = int(policy_cfg.getboolean("sarl", "with_agent_type"))
    kin = kin or "holonomic"
    # agents treat every non-"holonomic" string as rotational (agent.py:166-169);
    # rotate() keeps theta only for the literal "unicycle" (cadrl.py:261)
    p.robot_kinematics = _abi.HOLONOMIC if kin == "holonomic" else _abi.UNICYCLE
    p.rotate_unicycle = int(kin == "unicycle")
    return p


_SCALARS = [n for n, t in _abi.EbcParams._fields_ if not hasattr(t, "_length_")]
_ARRAYS = [n for n, t in _abi.EbcParams._fields_ if hasattr(t, "_length_")]


def params_to_dict(p):
    d = {k: getattr(p, k) for k in _SCALARS}
    d.update({k: list(getattr(p, k)) for k in _ARRAYS})
    return d


def params_from_dict(d):
    p = _abi.default_params()
    for k in _SCALARS:
        if k in d and k != "struct_size":
            setattr(p, k, d[k])
    for k in _ARRAYS:
        if k in d:
            arr = getattr(p, k)
            for i, v in enumerate(d[k]):
                arr[i] = v
    p.struct_size = C.sizeof(_abi.EbcParams)
    return p
