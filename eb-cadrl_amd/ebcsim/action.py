"""Robot / human actions (simulator/utils/action.py:3-5)."""
import collections

ActionXY = collections.namedtuple("ActionXY", "vx vy")
ActionRot = collections.namedtuple("ActionRot", "v r")
ActionXYRot = collections.namedtuple("ActionXYRot", "vx vy r")  # defined by the reference, unused by its step path
