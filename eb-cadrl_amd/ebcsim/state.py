"""Value objects crossing the env boundary (simulator/utils/state.py:1-92).

`FullState + ObservableState` concatenates to the 15-column tuple
(px, py, vx, vy, radius, gx, gy, v_pref, theta | px1, py1, vx1, vy1, radius1, type) that
rl/policy/multi_human_rl.py:52-60 feeds to rotate()."""

_OBS = ("px", "py", "vx", "vy", "radius")
_FULL = ("px", "py", "vx", "vy", "radius", "gx", "gy", "v_pref", "theta")


class _Row(object):
    __slots__ = ()
    _fields = ()

    def _values(self):
        return tuple(getattr(self, f) for f in self._fields)

    def __str__(self):
        return " ".join(str(x) for x in self._values() + (self.obj_type,))

    @property
    def position(self):
        return (self.px, self.py)

    @property
    def velocity(self):
        return (self.vx, self.vy)


class ObservableState(_Row):
    __slots__ = _OBS + ("obj_type",)
    _fields = _OBS

    def __init__(self, px, py, vx, vy, radius, obj_type=None):
        self.px, self.py, self.vx, self.vy, self.radius, self.obj_type = px, py, vx, vy, radius, obj_type

    def __add__(self, other):  # tuple + observable -> tuple, the type rides last
        return other + self._values() + (self.obj_type,)


class FullState(_Row):
    __slots__ = _FULL + ("obj_type",)
    _fields = _FULL

    def __init__(self, px, py, vx, vy, radius, gx, gy, v_pref, theta, obj_type=None):
        (self.px, self.py, self.vx, self.vy, self.radius, self.gx, self.gy, self.v_pref,
         self.theta, self.obj_type) = px, py, vx, vy, radius, gx, gy, v_pref, theta, obj_type

    @property
    def goal_position(self):
        return (self.gx, self.gy)

    def __add__(self, other):  # full + observable: observable.__add__(full tuple)
        return other + self._values()


class JointState(object):
    def __init__(self, self_state, agent_states):
        assert isinstance(self_state, FullState)
        for a in agent_states:
            assert isinstance(a, ObservableState)
        self.self_state = self_state
        self.agent_states = agent_states
