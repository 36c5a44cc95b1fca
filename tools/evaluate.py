#!/usr/bin/env python3
"""What rl/test.py does with a trained model (rl/test.py:99-151 -> Explorer.run_k_episodes over the test
cases, rl/utils/explorer.py:116-131), on the device: `--cases` test cases (seed 1000 + case, the reference's
"test" phase) as one batch, the SARL value network from a reference .pth driving every robot, the
reference's metrics at the end.

    python3 tools/evaluate.py --weights tests/golden/weights/sarl_n10_ebcadrl.pth \
        --env-config eb-cadrl_amd/configs/bench_metric.config --policy-config eb-cadrl_amd/configs/policy_agent_type.config \
        --cases 1000 [--policy orca]     (orca: the imitation-learning demonstrator instead of the network)
"""
import argparse
import configparser
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "eb-cadrl_amd")):
    sys.path.insert(0, p)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--weights", default=os.path.join(ROOT, "tests", "golden", "weights", "sarl_n10_ebcadrl.pth"))
    ap.add_argument("--env-config", default=os.path.join(ROOT, "eb-cadrl_amd", "configs", "bench_metric.config"))
    ap.add_argument("--policy-config", default=os.path.join(ROOT, "eb-cadrl_amd", "configs", "policy_agent_type.config"))
    ap.add_argument("--cases", type=int, default=1000)
    ap.add_argument("--first-case", type=int, default=0)
    ap.add_argument("--gamma", type=float, default=0.9)
    ap.add_argument("--policy", default="sarl", choices=["sarl", "orca"])
    ap.add_argument("--safety-space", type=float, default=0.15)
    ap.add_argument("--device-scenes", action="store_true",
                    help="generate the test scenes on the device (ebc_generate_reset) instead of on the host")
    args = ap.parse_args()
    import torch
    from ebcsim import _abi, actions as ebc_actions, config as ebc_config, scene as ebc_scene
    from ebcsim.batched import BatchedEnv
    from ebcsim.sarl import DeviceSarlPolicy, SarlValueNet
    from ebcsim.train import evaluate
    cfg, pol = configparser.RawConfigParser(), configparser.RawConfigParser()
    cfg.read(args.env_config)
    pol.read(args.policy_config)
    params = ebc_config.params_from_config(cfg, pol)
    sc = ebc_scene.SceneConfig.from_config(cfg)
    t0 = time.perf_counter()
    seeds = [ebc_scene.COUNTER_OFFSET["test"] + args.first_case + c for c in range(args.cases)]
    if args.device_scenes:
        gen = ebc_scene.gen_struct(sc, "test")
        env = BatchedEnv(params, args.cases, sum(gen.count), ebc_scene.max_static_rows(sc))
        env.generate_reset(gen, seeds[0])
        env.synchronize()
        v_pref = sc.robot.v_pref
    else:
        batch = ebc_scene.SceneBatch.from_scenes([ebc_scene.generate_scene(sc, s, "test") for s in seeds])
        env = BatchedEnv(params, args.cases, batch.N, batch.S)
        env.reset(batch)
        v_pref = float(batch.robot[0, 7])
    t1 = time.perf_counter()
    env.use_torch_stream()
    if args.policy == "sarl":
        net = SarlValueNet.load(args.weights, device="cuda:0")
        policy = DeviceSarlPolicy(net, ebc_actions.build_action_space(v_pref), args.gamma)
        decide, hp = (lambda e: policy.decide(e)[0]), _abi.HUMAN_CACHED
    else:
        act = torch.zeros((args.cases, 2), dtype=torch.float64, device="cuda:0")

        def decide(e):
            e.robot_orca_device(act, args.safety_space)
            return act
        hp = _abi.HUMAN_ORCA
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    m = evaluate(env, decide, args.gamma, human_policy=hp)
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    print("TEST  has success rate: %.2f, collision rate adult / bicycle / child / obstacle: %.2f / %.2f / %.2f / %.4f, "
          "timeout: %d, nav time: %.2f, total reward: %.4f" % (
              m["success_rate"], m["collision_rate_adult"], m["collision_rate_bicycle"], m["collision_rate_child"],
              m["collision_rate_obstacle"], m["timeout"], m["avg_nav_time"], m["total_reward:"]))
    print("Frequency of being in danger: %.2f and average min separate distance in danger: %.2f" % (
        m["Frequency of being in danger"] or 0.0, m["average min separate distance in danger"]))
    print(json.dumps({"cases": args.cases, "policy": args.policy, "scenes": "device" if args.device_scenes else "host", "scene_generation_s": t1 - t0, "episodes_s": t3 - t2,
                      "episodes_per_s": args.cases / (t3 - t2),
                      "metrics": {k: v for k, v in m.items() if not isinstance(v, list)}}))


if __name__ == "__main__":
    main()
