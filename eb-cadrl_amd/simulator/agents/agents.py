from ebcsim.agents import Adult, Bicycle, Child  # noqa: F401  (simulator/agents/agents.py:8,33,83)
