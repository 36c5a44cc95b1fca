from ebcsim.env import EntityBasedCollisionAvoidance  # noqa: F401  (simulator/env.py:19)
