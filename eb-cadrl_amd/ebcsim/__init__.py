"""ebcsim — MI355X-native batched crowd-navigation simulator (host side).

Mirror of the reference's simulator package for the step hot path; all compute
goes through the C ABI of libebcsim.so (include/ebcsim.h).  There is no CPU
fallback: constructing an environment without the HIP library raises.
"""
from . import _abi  # noqa: F401
