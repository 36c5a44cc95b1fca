from ebcsim.env import configure_env_policy_robot  # noqa: F401  (simulator/utils/test_utils.py:8-36)
