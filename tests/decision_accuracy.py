#!/usr/bin/env python3
"""Checker script (GPU): the value network's matrix-core path (split-bf16 blocks with the pair reductions in their
epilogues) against plain float32 GEMMs on a full decision batch — 1024 envs x 81 actions x 18 rows of the bench
workload, the shipped eb-cadrl weights and the random-init x2 network: largest value difference, and how many envs
take the float32 network's action with and without the top-2 refinement.  python tests/decision_accuracy.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "eb-cadrl_amd")):
    sys.path.insert(0, p)


def main():
    import torch
    import bench
    from ebcsim import _abi, actions as ebc_actions
    from ebcsim.batched import BatchedEnv
    from ebcsim.sarl import SarlValueNet
    from ebcsim.train import SarlModule
    dev = torch.device("cuda", 0)
    E = 1024
    params, batch = bench.build_batch("metric", E, 0)
    env = BatchedEnv(params, E, batch.N, batch.S)
    env.reset(batch)
    env.use_torch_stream()
    space = ebc_actions.build_action_space(float(batch.robot[0, 7]))
    acts = torch.tensor(space, dtype=torch.float64, device=dev)
    A = len(space)
    outs = env.alloc_step_outputs(("reward", "done"))
    torch.manual_seed(0)
    mod = SarlModule(env.T, [300, 200], [200, 100], [300, 200, 200, 1], [200, 200, 1])
    nets = {"random-init x2 network": SarlValueNet({k: v.detach() for k, v in mod.state_dict().items()}, device=str(dev)),
            "shipped eb-cadrl weights": SarlValueNet.load(os.path.join(ROOT, "tests", "golden", "weights", "sarl_n10_ebcadrl.pth"), device=str(dev))}
    for step in range(3):  # a few states along the episodes
        for _ in range(10 * step):
            env.step_device(outs, human_policy=_abi.HUMAN_ORCA, robot_policy=_abi.ROBOT_LINEAR, flags=_abi.FLAG_AUTO_RESET)
        bufs = env.alloc_lookahead_outputs(A, ("reward", "done", "info", "rows_rotated"))
        env.lookahead_device(acts, bufs, human_policy=_abi.HUMAN_ORCA)
        rows, reward = bufs["rows_rotated"], bufs["reward"]
        for name, net in nets.items():
            exact = torch.empty((E, A), dtype=torch.float32, device=dev)
            for e0 in range(0, E, 128):
                exact[e0:e0 + 128] = net.forward(rows[e0:e0 + 128].reshape(-1, env.R, env.T), exact=True).view(-1, A)
            want = (reward + 0.9 * exact.double())
            raw = net.action_values(rows, reward, 0.9, refine=0)
            ref = net.action_values(rows, reward, 0.9, refine=2)
            torch.cuda.synchronize()
            gap = torch.topk(want, 2, dim=1).values
            # where the coarse pass ranks the float32 network's best action (1 = first): the refinement looks at the top two
            best = want.argmax(1)
            rank = 1 + (raw > raw.gather(1, best[:, None])).sum(1)
            err = (raw - want).abs()
            print("   coarse rank of the float32-best action: max %d, envs with rank > 1: %d; |value - float32| median %.1e, 99.9th percentile %.1e" % (
                int(rank.max()), int((rank > 1).sum()), float(err.median()), float(torch.quantile(err.flatten()[:1000000].float(), 0.999))), flush=True)
            print("%s, after %d steps: max |value - float32| %.2e; same action as float32: %d / %d without refinement, "
                  "%d / %d with the top-2 refinement; smallest top-2 gap %.1e" % (
                      name, 10 * step, float((raw - want).abs().max()), int((raw.argmax(1) == want.argmax(1)).sum()), E,
                      int((ref.argmax(1) == want.argmax(1)).sum()), E, float((gap[:, 0] - gap[:, 1]).min())), flush=True)


if __name__ == "__main__":
    main()
