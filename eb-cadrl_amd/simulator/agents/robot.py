from ebcsim.agents import Robot  # noqa: F401  (simulator/agents/robot.py)
