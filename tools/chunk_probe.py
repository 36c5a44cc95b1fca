#!/usr/bin/env python3
"""Decision time against the row-chunk size of the value network (are the activations between the blocks served
from the 256 MB Infinity Cache when a chunk is small enough?).  python tools/chunk_probe.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "eb-cadrl_amd")):
    sys.path.insert(0, p)


def main():
    import torch
    import bench
    from ebcsim import actions as ebc_actions
    from ebcsim.batched import BatchedEnv
    from ebcsim.sarl import DeviceSarlPolicy, SarlValueNet
    from ebcsim.train import SarlModule
    dev = torch.device("cuda", 0)
    params, batch = bench.build_batch("metric", 1024, 0)
    env = BatchedEnv(params, 1024, batch.N, batch.S)
    env.reset(batch)
    env.use_torch_stream()
    torch.manual_seed(0)
    mod = SarlModule(env.T, [300, 200], [200, 100], [300, 200, 200, 1], [200, 200, 1])
    net = SarlValueNet({k: v.detach() for k, v in mod.state_dict().items()}, device=str(dev))
    space = ebc_actions.build_action_space(float(batch.robot[0, 7]))
    for shift in (21, 20, 19, 18, 17, 16, 21):
        pol = DeviceSarlPolicy(net, space, 0.9, chunk_rows=1 << shift)
        for _ in range(2):
            pol.decide(env)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(6):
            pol.decide(env)
        e1.record()
        torch.cuda.synchronize()
        print("chunk_rows 2^%d: %.3f ms per decision batch" % (shift, e0.elapsed_time(e1) / 6), flush=True)
        del pol


if __name__ == "__main__":
    main()
