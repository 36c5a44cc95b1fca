#!/usr/bin/env python3
"""Time each kernel role over a sweep of batch sizes (scenes tiled from 256 unique ones), with
HIP events on the launch stream; prints one line per (E, variant).  Diagnostic tool."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "eb-cadrl_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch
import bench
from ebcsim import _abi, actions, scene as ebc_scene
from ebcsim.batched import BatchedEnv

KEYS = ("n_humans", "px", "py", "vx", "vy", "gx", "gy", "radius", "v_pref", "type", "n_static",
        "spx", "spy", "sradius", "grid", "robot")


def tiled(base, E):
    reps = (E + base.n - 1) // base.n
    cut = lambda a: None if a is None else np.concatenate([a] * reps, 0)[:E]  # noqa: E731
    return ebc_scene.SceneBatch(E, base.N, base.S, *[cut(getattr(base, k)) for k in KEYS])


def timed(fn, n=60):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    workload = sys.argv[1] if len(sys.argv) > 1 else "metric"
    sizes = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [512, 1024, 2048, 4096, 8192, 16384, 32768]
    params, base = bench.build_batch(workload, 256, 0)
    space = actions.build_action_space(float(base.robot[0, 7]))
    for E in sizes:
        b = tiled(base, E)
        env = BatchedEnv(params, E, b.N, b.S)
        env.reset(b)
        env.use_torch_stream()
        outs = env.alloc_step_outputs(("reward", "done", "info", "obs_rotated"))
        fl = _abi.FLAG_AUTO_RESET
        t_orca = timed(lambda: env.step_device(outs, human_policy=_abi.HUMAN_ORCA, robot_policy=_abi.ROBOT_LINEAR, flags=fl))
        t_lin = timed(lambda: env.step_device(outs, human_policy=_abi.HUMAN_LINEAR, robot_policy=_abi.ROBOT_LINEAR, flags=fl))
        t_ext = timed(lambda: env.step_device(outs, human_policy=_abi.HUMAN_CACHED, robot_policy=_abi.ROBOT_LINEAR, flags=fl))
        acts = torch.tensor(space, dtype=torch.float64, device="cuda")
        lo = env.alloc_lookahead_outputs(len(space), ("reward", "done", "info"))
        t_la = timed(lambda: env.lookahead_device(acts, lo, human_policy=_abi.HUMAN_ORCA), n=30)
        t_rows = float("nan")
        if E * len(space) * env.R * env.T * 4 < 3e9:
            lr = env.alloc_lookahead_outputs(len(space), ("reward", "done", "info", "rows_rotated"))
            t_rows = timed(lambda: env.lookahead_device(acts, lr, human_policy=_abi.HUMAN_ORCA), n=20)
            gb = E * len(space) * env.R * env.T * 4 / 1e9
            print("E %6d lookahead[81, no rows] %8.2f us   lookahead[81, rows %.3f GB] %8.2f us = %.0f GB/s"
                  % (E, t_la, gb, t_rows, gb / (t_rows * 1e-6)), flush=True)
            del lr
        print("E %6d N %d  step[orca] %8.2f us  step[linear] %8.2f us  step[cached] %8.2f us  -> orca role ~%8.2f us"
              % (E, b.N, t_orca, t_lin, t_ext, t_orca - t_ext), flush=True)
        env.close()


if __name__ == "__main__":
    main()
