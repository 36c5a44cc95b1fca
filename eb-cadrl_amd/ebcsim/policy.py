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


def none_policy():
    return None


def _orca():
    from . import _abi
    return DeviceHumanPolicy("ORCA", _abi.HUMAN_ORCA)


policy_factory = {"linear": Linear, "orca": _orca, "none": none_policy}
