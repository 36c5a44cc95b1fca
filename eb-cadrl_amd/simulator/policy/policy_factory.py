from ebcsim.policy import policy_factory  # noqa: F401  (simulator/policy/policy_factory.py)
