"""ctypes mirror of include/ebcsim.h (struct layouts and enums only; loads nothing)."""
import ctypes as C

ABI_VERSION = 1

OK, ERR_INVALID, ERR_UNSUPPORTED, ERR_DEVICE, ERR_STATE = 0, -1, -2, -3, -4

# simulator/utils/utils.py:9-14
ADULT, BICYCLE, CHILD, ADULT_STATIC, ROBOT = 0, 1, 2, 3, 4

(INFO_NOTHING, INFO_DANGER, INFO_REACH_GOAL, INFO_COLLISION_OBSTACLE, INFO_COLLISION_ADULT,
 INFO_COLLISION_BICYCLE, INFO_COLLISION_CHILD, INFO_TIMEOUT) = range(8)

HOLONOMIC, UNICYCLE = 0, 1
HOST, DEVICE = 0, 1
HUMAN_EXTERNAL, HUMAN_LINEAR, HUMAN_ORCA, HUMAN_CACHED = 0, 1, 2, 3
ROBOT_EXTERNAL, ROBOT_LINEAR, ROBOT_ORCA = 0, 1, 2
FLAG_AUTO_RESET, FLAG_BORDER = 1, 2

_pd = C.c_void_p  # every buffer pointer travels as an address


class EbcParams(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("robot_kinematics", C.c_int32),
        ("robot_visible", C.c_int32),
        ("new_reward", C.c_int32),
        ("time_step", C.c_double),
        ("time_limit", C.c_double),
        ("map_size_m", C.c_double),
        ("map_resolution", C.c_double),
        ("time_max", C.c_double),
        ("time_good", C.c_double),
        ("max_goal_distance", C.c_double),
        ("success_reward", C.c_double),
        ("collision_penalty", C.c_double * 4),
        ("discomfort_dist", C.c_double * 3),
        ("discomfort_factor", C.c_double * 3),
        ("rotation_penalty_factor", C.c_double),
        ("orca_safety_space", C.c_double),
        ("orca_neighbor_dist", C.c_float),
        ("orca_time_horizon", C.c_float),
        ("orca_max_neighbors", C.c_int32),
        ("with_agent_type", C.c_int32),
        ("rotate_unicycle", C.c_int32),
        ("reserved", C.c_int32),
    ]


class EbcScene(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("n", C.c_int32), ("n_humans", _pd)] + [
        (k, _pd) for k in ("px", "py", "vx", "vy", "gx", "gy", "radius", "v_pref", "type",
                           "n_static", "spx", "spy", "sradius", "grid", "robot")
    ]


class EbcSceneGen(C.Structure):
    """include/ebcsim.h: the keys SceneGenerator.__init__ reads, phase resolved (scene.SceneConfig.gen_struct)."""
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("randomize_attributes", C.c_int32),
        ("count", C.c_int32 * 3),
        ("rule", C.c_int32 * 3),
    ] + [(k, C.c_double * 3) for k in ("radius", "v_pref", "radius_min", "radius_max", "v_pref_min", "v_pref_max")] + [
        (k, C.c_double) for k in ("square_width", "circle_radius", "discomfort_dist", "robot_radius", "robot_v_pref",
                                  "map_resolution", "map_size_m")
    ] + [(k, C.c_int32) for k in ("min_wall_length", "max_wall_length", "num_circles", "num_walls")]


class EbcSceneOut(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("location", C.c_int32), ("n_humans", _pd)] + [
        (k, _pd) for k in ("px", "py", "vx", "vy", "gx", "gy", "radius", "v_pref", "type",
                           "n_static", "spx", "spy", "sradius", "grid", "robot")
    ]


RULE_CIRCLE_CROSSING, RULE_SQUARE_CROSSING, RULE_SQUARE_CROSSING_OLD = 0, 1, 2


class EbcStepArgs(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("location", C.c_int32),
        ("human_policy", C.c_int32),
        ("robot_policy", C.c_int32),
        ("flags", C.c_int32),
        ("reserved", C.c_int32),
    ] + [
        (k, _pd) for k in ("robot_action", "border", "reward", "done", "info", "dmin",
                           "dist_to_goal", "robot_action_out", "human_action", "ob", "obs_rotated")
    ]


class EbcLookaheadArgs(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("location", C.c_int32),
        ("human_policy", C.c_int32),
        ("n_actions", C.c_int32),
        ("flags", C.c_int32),
        ("reserved", C.c_int32),
    ] + [
        (k, _pd) for k in ("actions", "border", "reward", "done", "info", "dmin", "next_ob",
                           "rows_rotated")
    ]


class EbcStepKArgs(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("location", C.c_int32),
        ("K", C.c_int32),
        ("human_policy", C.c_int32),
        ("robot_policy", C.c_int32),
        ("flags", C.c_int32),
        ("robot_safety_space", C.c_double),
    ] + [
        (k, _pd) for k in ("robot_action", "state_rotated", "n_rows", "robot_action_out", "reward", "done", "info",
                           "dmin", "dist_to_goal", "obs_rotated")
    ]


class EbcStateView(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("location", C.c_int32)] + [
        (k, _pd) for k in ("px", "py", "vx", "vy", "gx", "gy", "radius", "v_pref", "type",
                           "n_humans", "robot", "global_time", "arrival_time", "done")
    ]


MLP_IN_FRAGMENTS = 1
MLP_GENERAL_KERNEL = 1


class EbcMlpArgs(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("M", C.c_int32), ("relu_out", C.c_int32), ("group_rows", C.c_int32),
                ("seg_rows", C.c_int32), ("flags", C.c_int32)] + [(k, C.c_void_p) for k in ("x", "frag_in", "row_bias", "row_weight", "y", "partial",
                                                                       "frag_out")]


def default_params():
    """Reference defaults: ORCA constants simulator/policy/orca.py:58-69, reward.py:25."""
    p = EbcParams()
    p.struct_size = C.sizeof(EbcParams)
    p.robot_kinematics = HOLONOMIC
    p.robot_visible = 0
    p.new_reward = 0
    p.time_step = 0.25
    p.time_limit = 25
    p.map_size_m = 9.0
    p.map_resolution = 0.1
    nan = float("nan")
    p.time_max = nan
    p.time_good = 10.0
    p.max_goal_distance = nan
    p.success_reward = 1.0
    for i in range(4):
        p.collision_penalty[i] = nan
    for i in range(3):
        p.discomfort_dist[i] = 0.1
        p.discomfort_factor[i] = 0.5
    p.rotation_penalty_factor = 0.0
    p.orca_safety_space = 0.0
    p.orca_neighbor_dist = 10.0
    p.orca_time_horizon = 5.0
    p.orca_max_neighbors = 10
    p.with_agent_type = 0
    p.rotate_unicycle = 0
    return p


def rot_width(params):
    return 17 if params.with_agent_type else 13
