#!/usr/bin/env python3
"""Time the 81-action look-ahead sweep with rotated rows left in HBM (the one HBM-bound output of
the path): python3 tools/lookahead_bench.py [envs]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "eb-cadrl_amd")):
    sys.path.insert(0, p)


def main():
    import torch
    import bench
    from ebcsim import _abi, actions
    from ebcsim.batched import BatchedEnv
    E = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    params, batch = bench.build_batch("metric", E, 0)
    env = BatchedEnv(params, E, batch.N, batch.S)
    env.reset(batch)
    env.use_torch_stream()
    space = actions.build_action_space(float(batch.robot[0, 7]))
    acts = torch.tensor(space, dtype=torch.float64, device="cuda")
    bufs = env.alloc_lookahead_outputs(len(space), ("reward", "rows_rotated"))
    for _ in range(5):
        env.lookahead_device(acts, bufs, human_policy=_abi.HUMAN_ORCA)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 30
    e0.record()
    for _ in range(n):
        env.lookahead_device(acts, bufs, human_policy=_abi.HUMAN_ORCA)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    nbytes = bufs["rows_rotated"].numel() * 4
    print("%d envs x %d actions: %.1f us per sweep (ORCA prelude + look-ahead), rows %.1f MB -> %.2f TB/s" % (
        E, len(space), ms * 1e3, nbytes / 1e6, nbytes / (ms * 1e-3) / 1e12))


if __name__ == "__main__":
    main()
