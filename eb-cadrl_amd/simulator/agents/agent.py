from ebcsim.agents import Agent  # noqa: F401  (simulator/agents/agent.py)
