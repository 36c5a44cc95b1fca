"""Terminal / per-step info objects (simulator/utils/info.py).  Callers dispatch on the class
(rl/utils/explorer.py:48-80) and scrape str(info) from logs (rl/utils/plot.py:64-89), so both
are kept; the kernels return the code, `from_code` rebuilds the object."""
from . import _abi


class Info(object):
    text = None

    def __init__(self, dist_to_goal=None, dmin_adult=None, dmin_bicycle=None, dmin_child=None):
        self.dist_to_goal = dist_to_goal
        self.dmin_adult = dmin_adult
        self.dmin_bicycle = dmin_bicycle
        self.dmin_child = dmin_child

    def __str__(self):
        return self.text


def _kind(name, text):
    return type(name, (Info,), {"text": text})


Timeout = _kind("Timeout", "Timeout")
ReachGoal = _kind("ReachGoal", "Reaching goal")
CollisionAdult = _kind("CollisionAdult", "CollisionAdult")
CollisionBicycle = _kind("CollisionBicycle", "CollisionBicycle")
CollisionObstacle = _kind("CollisionObstacle", "CollisionObstacle")
CollisionChild = _kind("CollisionChild", "CollisionChild")
Collision = _kind("Collision", "Collision")
CollisionOtherAgent = _kind("CollisionOtherAgent", "Collision from other agent")


class Danger(Info):
    text = "Too close"

    def __init__(self, min_dist, dist_to_goal=None, dmin_adult=None, dmin_bicycle=None, dmin_child=None):
        Info.__init__(self, dist_to_goal, dmin_adult, dmin_bicycle, dmin_child)
        self.min_dist = min_dist


class Nothing(Info):
    text = ""

    def __init__(self, dmin_adult=None, dmin_bicycle=None, dmin_child=None):
        Info.__init__(self, None, dmin_adult, dmin_bicycle, dmin_child)


_BY_CODE = {_abi.INFO_TIMEOUT: Timeout, _abi.INFO_REACH_GOAL: ReachGoal,
            _abi.INFO_COLLISION_ADULT: CollisionAdult, _abi.INFO_COLLISION_BICYCLE: CollisionBicycle,
            _abi.INFO_COLLISION_OBSTACLE: CollisionObstacle, _abi.INFO_COLLISION_CHILD: CollisionChild}


def from_code(code, dist_to_goal, dmin, discomfort_dist):
    """code: EBC_INFO_*; dmin = (adult, bicycle, child).  Danger carries the dmin of the first
    type under its discomfort distance in the order child, bicycle, adult (reward.py:136-165)."""
    code = int(code)
    da, db, dc = (float(x) for x in dmin)
    if code == _abi.INFO_NOTHING:
        return Nothing(da, db, dc)
    if code == _abi.INFO_DANGER:
        if dc < discomfort_dist[2]:
            m = dc
        elif db < discomfort_dist[1]:
            m = db
        else:
            m = da
        return Danger(m, float(dist_to_goal), da, db, dc)
    return _BY_CODE[code](float(dist_to_goal), da, db, dc)


__all__ = ["Info", "Timeout", "ReachGoal", "Danger", "CollisionAdult", "CollisionBicycle",
           "CollisionObstacle", "CollisionChild", "Collision", "CollisionOtherAgent", "Nothing"]
