"""Angular local map of the static obstacles around the robot (simulator/env.py:468-628).

Host-side (numpy scalars): SURVEY 8(f)(3) — the reference spends 95 % of `step` here when a scene
has obstacles, yet no policy on the path reads the result (agents/robot.py:20-24 ignores
`local_map`).  The facade therefore computes it only on request; the kernels do not."""
import math

import numpy as np


def _sweep(vertex, edge, theta, res, dim, angle_min, rdv, seen):
    """One calculate_angular_map_distances call (env.py:468-568).  `seen` = [(sector, point)] of the
    earlier calls of the same sweep; the sectors between this point's and each earlier one are
    filled by walking along the segment in steps of 1 / idx_diff."""
    c, s = np.cos(theta), np.sin(theta)

    def polar(x, y):
        rx = (x - edge[0]) * c + (y - edge[1]) * s
        ry = (y - edge[1]) * c - (x - edge[0]) * s
        return rx, ry

    rx, ry = polar(vertex[0], vertex[1])
    sector = int((math.atan2(ry, rx) - angle_min) / float(res))
    if 0 <= sector < dim:
        rdv[sector] = min(rdv[sector], np.linalg.norm([rx, ry]))
    for old, loc in seen:
        wrapped = abs(sector - old) > np.pi / res
        if wrapped:
            span = dim - sector + old if sector > old else dim - old + sector
        else:
            span = abs(sector - old)
        from_new = (sector < old and not wrapped) or (sector > old and wrapped)
        start, a, b = (sector, vertex, loc) if from_new else (old, loc, vertex)
        for i in range(span):
            if 0 <= start + i < dim:
                t = i / float(span)
                rx, ry = polar(a[0] + t * (b[0] - a[0]), a[1] + t * (b[1] - a[1]))
                k = (start + i) % dim
                rdv[k] = min(rdv[k], np.linalg.norm([rx, ry]))
    seen.append((sector, vertex))


def angular_map(obstacle_vertices, px, py, radius, theta, max_range, dim, angle_min, angle_max,
                normalize=True):
    """get_local_map_angular (env.py:570-628): `dim` sector minima of the distance from the four
    corners of the robot's bounding box to the obstacle polygons, in the robot's heading frame."""
    rdv = max_range * np.ones([dim])
    res = (angle_max - angle_min) / float(dim)
    corners = [(px + sx * radius, py + sy * radius) for sx, sy in ((-1, -1), (1, -1), (-1, 1), (1, 1))]
    for poly in obstacle_vertices:           # polygon outlines seen from each corner
        for corner in corners:
            seen = []
            for vertex in poly:
                _sweep(vertex, corner, theta, res, dim, angle_min, rdv, seen)
    for poly in obstacle_vertices:           # each vertex seen from the four corners
        for vertex in poly:
            seen = []
            for corner in corners:
                _sweep(vertex, corner, theta, res, dim, angle_min, rdv, seen)
    if normalize:
        rdv /= float(max_range)
    return rdv
