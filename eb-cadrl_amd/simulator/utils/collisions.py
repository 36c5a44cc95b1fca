from ebcsim.collisions import compute_collision_agent_with_robot, point_to_segment_dist  # noqa: F401  (simulator/utils/collisions.py:4,29)
