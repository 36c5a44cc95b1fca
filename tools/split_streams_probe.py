#!/usr/bin/env python3
"""The bench batch as S independent sub-batches, each on its handle's own HIP stream: do the launches of different
sub-batches overlap (one's tail under another's head)?   python3 tools/split_streams_probe.py [envs] [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "eb-cadrl_amd")):
    sys.path.insert(0, p)
import torch, bench
from ebcsim import _abi
from ebcsim.batched import BatchedEnv

E = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
K = int(sys.argv[2]) if len(sys.argv) > 2 else 500
params, batch = bench.build_batch("metric", E, 0)
kw = dict(human_policy=_abi.HUMAN_ORCA, robot_policy=_abi.ROBOT_LINEAR, flags=_abi.FLAG_AUTO_RESET)
for S in (1, 2, 4, 8):
    n = E // S
    envs, outs = [], []
    for s in range(S):
        sub = bench.ebc_scene_slice_range(batch, s * n, (s + 1) * n)
        e = BatchedEnv(params, n, batch.N, batch.S)   # keeps its private stream
        e.reset(sub)
        envs.append(e)
        outs.append(e.alloc_step_outputs(("reward", "done", "info", "obs_rotated")))
    for _ in range(50):
        for e, o in zip(envs, outs):
            e.step_device(o, **kw)
    for e in envs:
        e.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        for e, o in zip(envs, outs):
            e.step_device(o, **kw)
    for e in envs:
        e.synchronize()
    dt = time.perf_counter() - t0
    print("sub-batches %d x %d envs: %.2f us per step of %d envs = %.2f G agent-steps/s" % (
        S, n, dt / K * 1e6, E, float(batch.n_humans.sum()) * K / dt / 1e9))
    del envs, outs

# the same with the steps enqueued by one C call per sub-batch (ebc_step_k), one host thread per sub-batch:
# the host cost of a launch (Python + ctypes) no longer limits 4 and 8 sub-batches
import threading
for S in (1, 2, 3, 4, 6, 8):
    n = E // S
    envs, outs = [], []
    for s in range(S):
        sub = bench.ebc_scene_slice_range(batch, s * n, (s + 1) * n)
        e = BatchedEnv(params, n, batch.N, batch.S)
        e.reset(sub)
        envs.append(e)
        outs.append(e.alloc_step_k_outputs(K, ("reward", "done", "info")))
    def run(e, o):
        e.step_k_device(o, K, human_policy=_abi.HUMAN_ORCA, robot_policy=_abi.ROBOT_LINEAR, flags=_abi.FLAG_AUTO_RESET)
        e.synchronize()
    for rep in range(2):
        ths = [threading.Thread(target=run, args=(e, o)) for e, o in zip(envs, outs)]
        t0 = time.perf_counter()
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        dt = time.perf_counter() - t0
    tot = float(sum(bench.ebc_scene_slice_range(batch, s * n, (s + 1) * n).n_humans.sum() for s in range(S)))
    print("threads: sub-batches %d x %d envs, %d steps per C call (no observation rows): %.2f us per step = %.2f G agent-steps/s" % (
        S, n, K, dt / K * 1e6, tot * K / dt / 1e9))
    del envs, outs
