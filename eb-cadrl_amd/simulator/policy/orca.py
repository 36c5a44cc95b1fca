from ebcsim.policy import ORCA  # noqa: F401  (simulator/policy/orca.py:8; rl/test.py:11 imports it)
