"""Checker script (uses the CPU oracle, so it lives under tests/): long full-size runs of the HIP step against the
oracle, every env every step.  python tests/soak_parity.py (on the GPU box)."""
import os, sys, json, time
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT,"eb-cadrl_amd"), os.path.join(ROOT,"tests")): sys.path.insert(0,p)
import numpy as np, torch, bench
from ebcsim import _abi
from ebcsim.batched import BatchedEnv
from oracle import oracle
for workload, E, steps in (("metric", 4096, 400), ("cfg4", 16384, 120), ("metric", 1000, 600)):
    params, b = bench.build_batch(workload, E, 1)
    params.time_limit = 7
    g = BatchedEnv(params, E, b.N, b.S); o = oracle.OracleEnv(params, E, b.N, b.S)
    g.reset(b); o.reset(b); g.use_torch_stream()
    outs = g.alloc_step_outputs(("reward","done","info","obs_rotated","human_action"))
    oracle.set_threads(16)
    worst = 0.0; restarts = 0; t0=time.time()
    for t in range(steps):
        g.step_device(outs, human_policy=_abi.HUMAN_ORCA, robot_policy=_abi.ROBOT_LINEAR, flags=_abi.FLAG_AUTO_RESET)
        ref = o.step(human_policy=_abi.HUMAN_ORCA, robot_policy=_abi.ROBOT_LINEAR, flags=_abi.FLAG_AUTO_RESET)
        g.synchronize()
        assert (outs["done"].cpu().numpy() == ref["done"]).all(), (workload, t)
        assert (outs["info"].cpu().numpy() == ref["info"]).all(), (workload, t)
        worst = max(worst, float(np.abs(outs["reward"].cpu().numpy()-ref["reward"]).max()), float(np.abs(outs["human_action"].cpu().numpy()-ref["human_action"]).max()))
        d = float(np.abs(outs["obs_rotated"].cpu().numpy()-ref["obs_rotated"]).max()); assert d <= 1e-5, (workload,t,d)
        restarts += int(ref["done"].sum())
    sg, so = g.get_state(), o.get_state()
    for k in sg: assert np.allclose(sg[k], so[k], atol=1e-9, rtol=0), k
    print(workload, E, steps, "steps: masks equal, max |d| f64", worst, "restarts", restarts, "%.1fs"%(time.time()-t0), flush=True)
