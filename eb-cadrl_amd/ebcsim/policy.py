"""Host-side robot policies of the simulator package (simulator/policy/*).  Human policies run
inside the kernels; what stays on the host is what a *robot* can be given through
`robot.set_policy()` without a value network."""
import numpy as np

from .action import ActionXY


class Policy(object):
    """simulator/policy/policy.py:5-54"""

    def __init__(self):
        self.trainable = False
        self.phase = None
        self.model = None
        self.device = None
        self.last_state = None
        self.time_step = None
        self.kinematics = None
        self.name = None
        self.multiagent_training = None

    def configure(self, config):
        return

    def set_phase(self, phase):
        self.phase = phase

    def set_device(self, device):
        self.device = device

    def get_model(self):
        return self.model

    def predict(self, state, env=None):
        raise NotImplementedError

    @staticmethod
    def reach_destination(state):
        s = state.self_state
        return bool(np.linalg.norm((s.py - s.gy, s.px - s.gx)) < s.radius)


class Linear(Policy):
    """simulator/policy/linear.py:6-23"""

    def __init__(self):
        Policy.__init__(self)
        self.name = "Linear"
        self.kinematics = "holonomic"
        self.multiagent_training = True

    def predict(self, state, env=None):
        s = state.self_state
        theta = np.arctan2(s.gy - s.py, s.gx - s.px)
        return ActionXY(np.cos(theta) * s.v_pref, np.sin(theta) * s.v_pref)


class DeviceHumanPolicy(Policy):
    """Marker for `policy = orca` / `policy = linear` in a human section: the arithmetic lives in
    the HIP kernels (EBC_HUMAN_ORCA / EBC_HUMAN_LINEAR), not in a host object."""

    def __init__(self, name, code):
        Policy.__init__(self)
        self.name = name
        self.code = code
        self.kinematics = "holonomic"

    def predict(self, state, env=None):
        raise NotImplementedError(
            "%s runs on the device for the humans; as a host-side *robot* policy it is not part "
            "of the accelerated path" % self.name)


class ORCA(DeviceHumanPolicy):
    """simulator/policy/orca.py:8-157.  In a human section it is the marker above (EBC_HUMAN_ORCA in the
    kernels).  Given to the ROBOT (`robot.set_policy(policy_factory["orca"]())`, the imitation-learning
    demonstrator of rl/train.py:99-143) its predict() is ebc_robot_orca on the env's backend: the
    robot's ORCA velocity against the rows of the current observation (humans, then the static
    obstacles as pedestrians), radii + 0.01 + safety_space."""

    def __init__(self):
        from . import _abi
        DeviceHumanPolicy.__init__(self, "ORCA", _abi.HUMAN_ORCA)
        self.trainable = False
        self.multiagent_training = None
        self.safety_space = 0
        self.neighbor_dist, self.max_neighbors = 10, 10  # orca.py:64-69: what the kernels use (EbcParams)
        self.time_horizon = self.time_horizon_obst = 5
        self.radius, self.max_speed = 0.3, 200
        self.sim = None  # orca.py:70: this object's rvo2 simulator (rows, the radii and maxSpeed it was built with)

    def predict(self, state, env=None):
        """`state` must be the env's current joint state (Robot.act builds it from the observation the
        env has just returned): the arithmetic reads the device-resident copy of it.  Like the reference's
        object this one keeps its simulator from call to call (orca.py:96-133): it is rebuilt only when the
        number of agents changes, so between rebuilds the radii (+ 0.01 + safety_space) and the robot's maxSpeed
        are those of the call that built it — also across env.reset()."""
        if env is None or getattr(env, "_backend", None) is None:
            raise ValueError("ORCA as a robot policy needs the env it acts in (robot.act(ob, env=env)) after reset()")
        s = state.self_state
        rb = env._backend.get_state()["robot"][0]
        if (s.px, s.py, s.gx, s.gy) != (rb[0], rb[1], rb[5], rb[6]):
            raise ValueError("ORCA.predict: the state passed in is not the env's current state")
        be = env._backend
        if self.sim is not None and self.sim["radius"].shape[1] == be.R:
            be.robot_orca_sim_state(self.sim)   # this object's simulator onto the handle that serves the call
        else:
            be.robot_orca_sim(True)             # none yet, or another number of rows: built from this state
        vx, vy = be.robot_orca(self.safety_space)[0]
        self.sim = be.robot_orca_sim_state()
        self.last_state = state
        return ActionXY(float(vx), float(vy))


def none_policy():
    return None


def _orca():
    return ORCA()


policy_factory = {"linear": Linear, "orca": _orca, "none": none_policy}
