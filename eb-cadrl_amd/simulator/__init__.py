"""Drop-in module paths of the reference's `simulator` package, backed by ebcsim.

Put `eb-cadrl_amd/` ahead of the reference on sys.path and `from simulator.utils.info import *`,
`from simulator.utils.action import ActionXY`, `gym.make("EntityBasedCollisionAvoidance-v0")` in
rl/*.py resolve to the MI355X-native implementation (simulator/__init__.py:1-7 of the reference
registers the same id)."""
try:  # gym is optional: ebcsim.env.make() does the same without it
    from gym.envs.registration import register

    register(id="EntityBasedCollisionAvoidance-v0",
             entry_point="ebcsim.env:EntityBasedCollisionAvoidance")
except ImportError:
    pass
