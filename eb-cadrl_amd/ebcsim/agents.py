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
        for k in ("v_pref_min", "v_pref_max", "radius_min", "radius_max"):  # agent.py:32-35
            setattr(self, k, config.getfloat(section, k, fallback=None))

    def print_info(self):
        import logging
        logging.info("Agent is {} and has {} kinematic constraint".format(
            "visible" if self.visible else "invisible", self.kinematics))

    def sample_random_attributes(self):
        """agent.py:48-56 (numpy's global stream, as there)"""
        self.v_pref = np.random.uniform(self.v_pref_min, self.v_pref_max)
        self.radius = np.random.uniform(self.radius_min, self.radius_max)
        assert 0 < self.v_pref < 20 and 0 < self.radius < 20

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

    def get_next_observable_state(self, action):
        """agent.py:80-93"""
        nx, ny = self.compute_position(action, self.time_step)
        if self.kinematics == "holonomic":
            vx, vy = action.vx, action.vy
        else:
            th = self.theta + action.r
            vx, vy = action.v * np.cos(th), action.v * np.sin(th)
        return ObservableState(nx, ny, vx, vy, self.radius, self.agent_type)

    def get_state_dict(self):
        return {"pos": (self.px, self.py), "vel": (self.vx, self.vy), "radius": self.radius,
                "goal": (self.gx, self.gy), "v_pref": self.v_pref, "theta": self.theta,
                "agent_type": self.agent_type}

    def set_from_state_dict(self, state):
        (self.px, self.py), (self.vx, self.vy) = state["pos"], state["vel"]
        self.gx, self.gy = state["goal"]
        self.radius, self.v_pref, self.theta = state["radius"], state["v_pref"], state["theta"]
        if state.get("agent_type") is not None:
            self.agent_type = AgentType(state["agent_type"])

    def get_position(self):
        return self.px, self.py

    def set_position(self, position):
        self.px, self.py = position[0], position[1]

    def get_goal_position(self):
        return self.gx, self.gy

    def get_velocity(self):
        return self.vx, self.vy

    def set_velocity(self, velocity):
        self.vx, self.vy = velocity[0], velocity[1]

    def act(self, ob):
        return None

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

    def compute_velocity(self, action):
        """agent.py:190-200 (rotation actions only: ActionXYRot is not on the accelerated path)"""
        self.check_validity(action)
        th = self.theta + action.r
        return action.v * np.cos(th), action.v * np.sin(th)

    def step(self, action):
        """agent.py:202-228: the host mirror of what the kernels do to the robot row.  The env never
        calls it (the device state is the truth); it serves host-only callers such as unit tests."""
        self.px, self.py = self.compute_position(action, self.time_step)
        if self.kinematics == "holonomic":
            self.vx, self.vy = action.vx, action.vy
        else:
            self.theta = (self.theta + action.r) % (2 * np.pi)
            self.vx, self.vy = action.v * np.cos(self.theta), action.v * np.sin(self.theta)

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
        self.adults_in_FOV = None

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


class _ConfigHuman(Agent):
    """simulator/agents/agents.py:8-105: a human built from a config section.  The env moves its humans
    on the device; an object of this kind is what host-only callers construct (the reference's
    tests/test_collisions.py:21-24) and what a scene JSON's agents are checked against."""
    TYPE = None

    def __init__(self, config, section):
        Agent.__init__(self, config, section)
        self.agent_type = self.TYPE

    def act(self, ob=None, global_map=None, local_map=None):
        if ob is None:
            return self.policy.predict(self)
        state = JointState(self.get_full_state(), ob)
        if global_map is not None:
            return self.policy.predict(state, global_map, self)
        return self.policy.predict(state)


class Adult(_ConfigHuman):
    TYPE = AgentType.ADULT


class Bicycle(_ConfigHuman):
    TYPE = AgentType.BICYCLE


class Child(_ConfigHuman):
    TYPE = AgentType.CHILD


KIND = {_abi.ADULT: "adults", _abi.BICYCLE: "bicycles", _abi.CHILD: "children"}
