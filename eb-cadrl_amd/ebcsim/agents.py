"""Agent objects of the facade (simulator/agents/agent.py, robot.py).  The robot is a real host
object (policies are attached to it); humans are read-only views of the device state."""
import numpy as np

from . import _abi
from .action import ActionRot, ActionXY
from .policy import policy_factory
from .state import FullState, JointState, ObservableState

try:  # IntEnum like the reference (simulator/utils/utils.py:9-14)
    from enum import IntEnum

    class AgentType(IntEnum):
        ADULT = 0
        BICYCLE = 1
        CHILD = 2
        ADULT_STATIC = 3
        ROBOT = 4
except ImportError:  # pragma: no cover
    AgentType = None


class Agent(object):
    def __init__(self, config, section):
        self.visible = config.getboolean(section, "visible")
        self.v_pref = config.getfloat(section, "v_pref", fallback=None)
        self.radius = config.getfloat(section, "radius", fallback=None)
        self.policy = policy_factory[config.get(section, "policy")]()
        self.sensor = config.get(section, "sensor")
        self.kinematics = self.policy.kinematics if self.policy is not None else None
        self.px = self.py = self.gx = self.gy = self.vx = self.vy = self.theta = None
        self.time_step = None
        self.agent_type = None

    def set_policy(self, policy):
        self.policy = policy
        self.kinematics = policy.kinematics

    def set(self, px, py, gx, gy, vx, vy, theta, radius=None, v_pref=None, agent_type=None):
        self.px, self.py, self.gx, self.gy, self.vx, self.vy, self.theta = px, py, gx, gy, vx, vy, theta
        if radius is not None:
            self.radius = radius
        if v_pref is not None:
            self.v_pref = v_pref
        if agent_type is not None:
            self.agent_type = agent_type

    def get_observable_state(self):
        return ObservableState(self.px, self.py, self.vx, self.vy, self.radius, self.agent_type)

    def get_full_state(self):
        return FullState(self.px, self.py, self.vx, self.vy, self.radius, self.gx, self.gy,
                         self.v_pref, self.theta, self.agent_type)

    def get_position(self):
        return self.px, self.py

    def get_goal_position(self):
        return self.gx, self.gy

    def get_velocity(self):
        return self.vx, self.vy

    def check_validity(self, action):
        if self.kinematics == "holonomic":
            assert isinstance(action, ActionXY)
        else:
            assert isinstance(action, ActionRot)

    def compute_position(self, action, delta_t):
        """agent.py:164-188 (host mirror, used by callers that peek ahead)"""
        self.check_validity(action)
        if self.kinematics == "holonomic":
            return self.px + action.vx * delta_t, self.py + action.vy * delta_t
        theta = self.theta + action.r
        return (self.px + np.cos(theta) * action.v * delta_t,
                self.py + np.sin(theta) * action.v * delta_t)

    def reached_destination(self):
        return bool(np.linalg.norm(np.array(self.get_position()) - np.array(self.get_goal_position()))
                    < self.radius)


class Robot(Agent):
    """simulator/agents/robot.py"""

    def __init__(self, config, section):
        Agent.__init__(self, config, section)
        self.agent_type = AgentType.ROBOT
        self.action_index = None
        self.attention_weights = None
        self.last_state = None

    def act(self, ob, local_map=None, env=None):
        if self.policy is None:
            raise AttributeError("Policy attribute has to be set!")
        return self.policy.predict(JointState(self.get_full_state(), ob), env)


class HumanView(Agent):
    """One human of the scene, refreshed from the device state after every step."""

    def __init__(self, agent_type, policy):
        self.agent_type = agent_type
        self.policy = policy
        self.kinematics = "holonomic"
        self.visible = True
        self.sensor = "coordinates"
        self.time_step = None
        self.theta = 0


KIND = {_abi.ADULT: "adults", _abi.BICYCLE: "bicycles", _abi.CHILD: "children"}
