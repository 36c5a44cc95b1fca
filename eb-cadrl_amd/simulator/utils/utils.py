from ebcsim.agents import AgentType  # noqa: F401  (simulator/utils/utils.py:9-14)
