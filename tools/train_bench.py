#!/usr/bin/env python3
"""BASELINE config 5 in small: the training schedule (imitation learning with the robot on ORCA, then
epsilon-greedy RL rounds) on the rank's env slice, gradients averaged over ranks with one flat all-reduce
per optimizer step.  One process per GPU:

    python3 tools/train_bench.py [--envs 1024] [--iterations 5]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/train_bench.py

Prints per-stage wall times of rank 0 (MAX over ranks for the RL rounds) as one JSON line."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "eb-cadrl_amd")):
    sys.path.insert(0, p)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=1024, help="envs per GPU")
    ap.add_argument("--il-steps", type=int, default=150)
    ap.add_argument("--il-epochs", type=int, default=2)
    ap.add_argument("--iterations", type=int, default=5)
    ap.add_argument("--steps-per-iteration", type=int, default=4)
    ap.add_argument("--train-batches", type=int, default=20)
    ap.add_argument("--batch-size", type=int, default=100)
    ap.add_argument("--device-scenes", action="store_true",
                    help="every episode on a scene generated on the device from the train seeds (run_training scene_gen)")
    args = ap.parse_args()
    import torch
    import torch.distributed as dist
    import bench
    from ebcsim import actions as ebc_actions
    from ebcsim.batched import BatchedEnv
    from ebcsim.train import SarlModule, collect_il, run_training, DeviceReplay
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)
    params, batch = bench.build_batch("metric", args.envs, rank)
    env = BatchedEnv(params, args.envs, batch.N, batch.S, device=local)
    env.reset(batch)
    env.use_torch_stream()
    torch.manual_seed(0)  # the same initial weights on every rank
    model = SarlModule(input_dim=env.T, mlp1_dims=[300, 200], mlp2_dims=[200, 100], mlp3_dims=[300, 200, 200, 1],
                       attention_dims=[200, 200, 1]).to(dev)
    space = ebc_actions.build_action_space(float(batch.robot[0, 7]))
    g = torch.Generator(device=dev).manual_seed(1 + rank)
    times = {}

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        return time.perf_counter()

    # stage timings: the IL rollouts alone (after one untimed warm-up call: first launches, allocations, clocks;
    # always on the HOST-generated batch, whatever --device-scenes says: the il_* fields are labelled so), then
    # the whole schedule
    mem = DeviceReplay(args.il_steps * args.envs, env.R, env.T, dev)
    collect_il(env, mem, min(args.il_steps, 20), 0.9, 0.15)
    del mem
    env.reset(batch)
    t0 = sync()
    mem = DeviceReplay(args.il_steps * args.envs, env.R, env.T, dev)
    stored, episodes = collect_il(env, mem, args.il_steps, 0.9, 0.15)
    t1 = sync()
    times["il_rollout_s"] = t1 - t0
    times["il_env_steps_per_s"] = args.il_steps * args.envs * world / (t1 - t0)
    times["il_fields"] = "il_* measured on the host-generated batch after a 20-step warm-up call (not on device-generated scenes)"
    del mem
    env.reset(batch)
    t0 = sync()
    marks = []
    scene_kw = {}
    if args.device_scenes:
        from ebcsim import scene as ebc_scene
        cfg_s = bench.scene_config("metric")
        if ebc_scene.max_static_rows(cfg_s) > batch.S:
            raise SystemExit("--device-scenes: the bench batch has fewer static rows than a generated map can need")
        scene_kw = dict(scene_gen=ebc_scene.gen_struct(cfg_s, "train"), scene_seed0=2000 + rank * (1 << 24))
    hist = run_training(env, model, space, 0.9, il_steps=args.il_steps, il_epochs=args.il_epochs, train_iterations=args.iterations, **scene_kw,
                        steps_per_iteration=args.steps_per_iteration, train_batches=args.train_batches,
                        batch_size=args.batch_size, capacity=max(100000, args.il_steps * args.envs), generator=g,
                        log=lambda line: marks.append((line.split(":")[0], sync())))
    t1 = sync()
    times["schedule_s"] = t1 - t0
    decisions = args.iterations * args.steps_per_iteration * args.envs * world
    rl = [t for name, t in marks if name.startswith("iteration")]
    if len(rl) >= 2:  # the RL rounds after the first (its first decision allocates the sweep's buffers)
        per_round = (rl[-1] - rl[0]) / (len(rl) - 1)
        times["rl_round_s"] = per_round
        times["rl_decisions_per_s"] = args.steps_per_iteration * args.envs * world / per_round
        times["rl_round"] = "%d decisions per env + %d optimizer steps of batch %d" % (
            args.steps_per_iteration, args.train_batches, args.batch_size)
    if rank == 0:
        print(json.dumps({"world": world, "envs_per_gpu": args.envs, "humans": int(batch.N), "il_states_stored": stored,
                          "il_episodes": episodes, "il_loss": hist["il_loss"], "rl_loss": hist["rl_loss"],
                          "rl_decisions": decisions, "scenes": "device" if args.device_scenes else "host batch",
                          "scene_pools": hist.get("scene_pools", 0), **times}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
