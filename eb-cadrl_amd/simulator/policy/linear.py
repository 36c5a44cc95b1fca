from ebcsim.policy import Linear  # noqa: F401  (simulator/policy/linear.py)
