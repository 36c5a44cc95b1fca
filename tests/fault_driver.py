#!/usr/bin/env python3
"""Mailbox give-up path of the fused ORCA step, on the GPU, in a process of its own (it loads the test build
libebcsim_fault.so: `make -C eb-cadrl_amd/csrc fault` = the product sources with a 2048-poll give-up and
a hook that makes one ORCA group skip its hand-off).  Run ONCE by tests/test_gpu_parity.py; prints a JSON line."""
import ctypes as C
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
os.environ["EBCSIM_LIB"] = os.path.join(ROOT, "eb-cadrl_amd", "lib", "libebcsim_fault.so")
sys.path[:0] = [os.path.join(ROOT, "eb-cadrl_amd"), ROOT, HERE]

import numpy as np  # noqa: E402

from ebcsim import _abi, _capi  # noqa: E402
from ebcsim.batched import BatchedEnv  # noqa: E402
from helpers import batch_from_init, load, params_of  # noqa: E402
from oracle import oracle  # noqa: E402


def main():
    L = _capi.lib()
    L.ebc_debug_withhold.restype, L.ebc_debug_withhold.argtypes = C.c_int, [C.c_int]
    z = load("traj_a5_linear_orcasub")
    params = params_of(z)
    E = 70
    b = batch_from_init(z, copies=E)
    g = BatchedEnv(params, E, b.N, b.S)
    o = oracle.OracleEnv(params, E, b.N, b.S)
    g.reset(b)
    o.reset(b)
    kw = dict(human_policy=_abi.HUMAN_ORCA, robot_policy=_abi.ROBOT_LINEAR)
    out = {"healthy_steps": 0}
    for t in range(3):  # the build itself is healthy
        og, oo = g.step(**kw), o.step(**kw)
        assert (og["info"] == oo["info"]).all() and np.abs(og["human_action"] - oo["human_action"]).max() <= 1e-9
        out["healthy_steps"] += 1
    assert L.ebc_debug_withhold(33 * b.N + 2) == 0  # human 2 of env 33 never publishes
    try:
        g.step(**kw)  # a host-location call: its copy-back carries the fault word
        out["first"] = "no error"
    except _capi.EbcError as e:
        out["first"] = e.code
    assert L.ebc_debug_withhold(-1) == 0
    out["sync_after"] = L.ebc_synchronize(g._h)             # reported once: the flag was cleared
    try:
        g.step(**kw)
        out["step_while_faulted"] = "no error"
    except _capi.EbcError as e:
        out["step_while_faulted"] = e.code                    # refused until reset
    g.reset(b)                                                # re-arms the mailboxes
    o.reset(b)
    worst = 0.0
    for t in range(5):
        og, oo = g.step(**kw), o.step(**kw)
        assert (og["info"] == oo["info"]).all() and (og["done"] == oo["done"]).all()
        worst = max(worst, float(np.abs(og["human_action"] - oo["human_action"]).max()),
                    float(np.abs(og["obs_rotated"] - oo["obs_rotated"]).max()))
    out["after_reset_max_err"] = worst
    out["sync_end"] = L.ebc_synchronize(g._h)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
