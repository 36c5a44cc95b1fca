from ebcsim.action import ActionRot, ActionXY, ActionXYRot  # noqa: F401  (simulator/utils/action.py)
