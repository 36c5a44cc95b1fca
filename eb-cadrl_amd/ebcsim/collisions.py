"""Host form of the swept circle-circle test (simulator/utils/collisions.py:4-57).

Inside `env.step` this arithmetic runs in the kernels (`ebc::closest_dist`, csrc/ebc_device.h); these
two functions keep the reference's module path alive for host-only callers — its unit tests build two
agent objects and ask for one pair (tests/test_collisions.py:12-143).  float64 throughout, the norm
through numpy like the reference, so results equal the `collisions.npz` goldens bit for bit."""
import numpy as np


def point_to_segment_dist(x1, y1, x2, y2, x3, y3):
    """Distance from (x3, y3) to the segment (x1, y1)-(x2, y2); collisions.py:4-26."""
    dx, dy = x2 - x1, y2 - y1
    if dx == 0 and dy == 0:
        return np.linalg.norm((x3 - x1, y3 - y1))
    u = ((x3 - x1) * dx + (y3 - y1) * dy) / (dx * dx + dy * dy)
    if u > 1:  # comparisons, not min/max: a NaN parameter stays NaN like in the reference
        u = 1
    elif u < 0:
        u = 0
    return np.linalg.norm((x1 + u * dx - x3, y1 + u * dy - y3))


def compute_collision_agent_with_robot(agent, robot, action, dmin, time_step):
    """(dmin', collided) for one agent against the robot taking `action` over `time_step`, in the
    agent's frame: the robot sweeps from the relative position along the relative velocity
    (collisions.py:29-57)."""
    px, py = agent.px - robot.px, agent.py - robot.py
    if robot.kinematics == "holonomic":
        vx, vy = agent.vx - action.vx, agent.vy - action.vy
    else:
        heading = action.r + robot.theta
        vx, vy = agent.vx - action.v * np.cos(heading), agent.vy - action.v * np.sin(heading)
    gap = (point_to_segment_dist(px, py, px + vx * time_step, py + vy * time_step, 0, 0)
           - agent.radius - robot.radius)
    if gap < 0:
        return dmin, True
    return (gap if gap < dmin else dmin), False
