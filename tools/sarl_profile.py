#!/usr/bin/env python3
"""Robot decisions on device (look-ahead sweep + SARL value network + argmax) for profiling:

    rocprofv3 --kernel-trace --stats --output-format csv -d out -o s -- python3 tools/sarl_profile.py [envs] [net]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "eb-cadrl_amd")):
    sys.path.insert(0, p)


def main():
    import torch
    import bench
    from ebcsim import _abi, actions
    from ebcsim.batched import BatchedEnv
    from ebcsim.sarl import DeviceSarlPolicy, SarlValueNet
    E = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    which = sys.argv[2] if len(sys.argv) > 2 else "sarl_n10_ebcadrl"
    params, batch = bench.build_batch("metric", E, 0)
    env = BatchedEnv(params, E, batch.N, batch.S)
    env.reset(batch)
    env.use_torch_stream()
    if which == "random":  # the bench's decision leg: the x2 architecture with random-init weights
        from ebcsim.train import SarlModule
        torch.manual_seed(0)
        mod = SarlModule(env.T, [300, 200], [200, 100], [300, 200, 200, 1], [200, 200, 1])
        net = SarlValueNet({k: v.detach() for k, v in mod.state_dict().items()}, device="cuda")
    else:
        net = SarlValueNet.load(os.path.join(ROOT, "tests", "golden", "weights", which + ".pth"), device="cuda")
    net.frag_handoff = os.environ.get("EBCSIM_FRAG_HANDOFF", "1") != "0"
    net.CHUNK_STREAMS = int(os.environ.get("EBCSIM_CHUNK_STREAMS", net.CHUNK_STREAMS))  # 1: per-kernel times without overlap
    space = actions.build_action_space(float(batch.robot[0, 7]))
    pol = DeviceSarlPolicy(net, space, 0.9, chunk_rows=int(os.environ.get("EBCSIM_CHUNK_ROWS", 1 << 19)))
    outs = env.alloc_step_outputs(("reward", "done"))
    for it in range(8):
        if it == 3:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        act, _ = pol.decide(env)
        env.step_device(outs, robot_action=act, human_policy=_abi.HUMAN_CACHED, flags=_abi.FLAG_AUTO_RESET)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print("%d envs x %d actions, net %s: %.2f ms per decision batch = %.0f decisions/s" % (E, len(space), which, dt * 1e3, E / dt))
    print("  coarse_eps %.3g, refine_stats %s" % (net.coarse_eps, getattr(net, "refine_stats", None)))


if __name__ == "__main__":
    main()
