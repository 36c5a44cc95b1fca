#!/usr/bin/env python3
"""Run the step in its variants under `rocprofv3 --kernel-trace` so that each kernel role can be
timed alone: ORCA+ENV (normal step), ENV only (linear humans), ORCA only (look-ahead prelude).
Dispatches are told apart by grid size in the trace.

    rocprofv3 --kernel-trace --output-format csv -d out -- python3 tools/profile_variants.py [workload]
    python3 tools/profile_variants.py --summarize out/*/*kernel_trace.csv
"""
import collections
import csv
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "eb-cadrl_amd")):
    sys.path.insert(0, p)


def summarize(paths):
    agg = collections.defaultdict(list)
    for path in paths:
        for r in csv.DictReader(open(path)):
            name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
            if "rocclr" in name or "at::native" in name:
                continue
            agg[(name, int(r.get("Grid_Size") or r["Grid_Size_X"]) // 64)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for (name, waves), v in sorted(agg.items()):
        v = sorted(v)[len(v) // 10:]  # drop the cold first decile
        print("%-34s waves %6d  n %4d  mean %8.2f us  min %8.2f us" % (name, waves, len(v), sum(v) / len(v) / 1e3, v[0] / 1e3))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--summarize":
        return summarize(sys.argv[2:])
    import torch
    import bench
    from ebcsim import _abi, actions
    from ebcsim.batched import BatchedEnv
    workload = sys.argv[1] if len(sys.argv) > 1 else "metric"
    E = int(sys.argv[2]) if len(sys.argv) > 2 else bench.WORKLOADS[workload][2]
    params, batch = bench.build_batch(workload, E, 0)
    env = BatchedEnv(params, E, batch.N, batch.S)
    env.reset(batch)
    env.use_torch_stream()
    outs = env.alloc_step_outputs(("reward", "done", "info", "obs_rotated"))
    fl = _abi.FLAG_AUTO_RESET
    for hp in (_abi.HUMAN_ORCA, _abi.HUMAN_LINEAR, _abi.HUMAN_ORCA):
        for _ in range(100):
            env.step_device(outs, human_policy=hp, robot_policy=_abi.ROBOT_LINEAR, flags=fl)
        torch.cuda.synchronize()
    space = actions.build_action_space(float(batch.robot[0, 7]))
    for _ in range(20):
        env.lookahead(space, human_policy=_abi.HUMAN_ORCA, want_rows=False)
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
