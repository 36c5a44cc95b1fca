from ebcsim.info import *  # noqa: F401,F403  (simulator/utils/info.py)
from ebcsim.info import __all__  # noqa: F401
