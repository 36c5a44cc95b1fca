from ebcsim.policy import Policy  # noqa: F401  (simulator/policy/policy.py)
