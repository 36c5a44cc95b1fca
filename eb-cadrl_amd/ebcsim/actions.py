"""Robot action space of the value-based policies (rl/policy/cadrl.py:91-116)."""
import numpy as np


def build_action_space(v_pref, kinematics="holonomic", speed_samples=5, rotation_samples=16):
    """[1 + rotation_samples * speed_samples][2] array: the zero action first, then
    rotation-major (rotation, speed) pairs.  Holonomic rows are ActionXY(vx, vy),
    every other kinematics gives ActionRot(v, r)."""
    holonomic = kinematics == "holonomic"
    scale = np.e - 1
    speeds = [(np.exp((i + 1) / speed_samples) - 1) / scale * v_pref for i in range(speed_samples)]
    if holonomic:
        rotations = np.linspace(0, 2 * np.pi, rotation_samples, endpoint=False)
    else:
        rotations = np.linspace(-np.pi / 4, np.pi / 4, rotation_samples)
    rows = [(0.0, 0.0)]
    for rot in rotations:
        for speed in speeds:
            rows.append((speed * np.cos(rot), speed * np.sin(rot)) if holonomic else (speed, rot))
    return np.array(rows, dtype=np.float64)
