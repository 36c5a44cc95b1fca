"""N > 1 path on CPU: two gloo ranks each own an env slice; together they must reproduce the
single-process result exactly (no data-path collective), and the timing reduction must take
the max over ranks and the sum of units.  The per-rank compute here is the CPU oracle — the
sharding logic under test is host code shared with bench.py."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "eb-cadrl_amd")


def _worker(rank, world, port, total, steps, ret):
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    from ebcsim import _abi, config as ebc_config, scene as ebc_scene, shard
    from oracle import oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = ebc_config.read_config(os.path.join(PKG, "configs", "bench_cfg2.config"))
    params = ebc_config.params_from_config(cfg)
    sc = ebc_scene.SceneConfig.from_config(cfg)
    start, count = shard.shard_range(total, rank, world)
    scenes = [ebc_scene.generate_scene(sc, s) for s in shard.scene_seeds(1000, start, count)]
    b = ebc_scene.SceneBatch.from_scenes(scenes, 5, 0)
    env = oracle.OracleEnv(params, count, 5, 0)
    env.reset(b)
    rewards = []
    for _ in range(steps):
        out = env.step(human_policy=_abi.HUMAN_ORCA, robot_policy=_abi.ROBOT_LINEAR)
        rewards.append(out["reward"].copy())
    local = torch.from_numpy(np.stack(rewards, 1))           # [count, steps]
    sizes = [shard.shard_range(total, r, world)[1] for r in range(world)]
    parts = [torch.zeros(sz, steps, dtype=torch.float64) for sz in sizes]
    dist.all_gather(parts, local) if len(set(sizes)) == 1 else _gather_ragged(dist, parts, local, rank)
    elapsed, units = shard.job_rate(0.5 + rank, count * 5 * steps)
    if rank == 0:
        ret["rewards"] = torch.cat(parts, 0).numpy()
        ret["elapsed"], ret["units"] = elapsed, units
    dist.barrier()
    dist.destroy_process_group()


def _gather_ragged(dist, parts, local, rank):
    for r in range(len(parts)):
        if r == rank:
            parts[r].copy_(local)
        dist.broadcast(parts[r], src=r)


@pytest.mark.parametrize("total", [12, 13])
def test_two_ranks_equal_one(total):
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    from ebcsim import _abi, config as ebc_config, scene as ebc_scene, shard
    from oracle import oracle
    steps = 15
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29500 + (os.getpid() + total) % 2000
    mp.spawn(_worker, args=(2, port, total, steps, ret), nprocs=2, join=True)
    cfg = ebc_config.read_config(os.path.join(PKG, "configs", "bench_cfg2.config"))
    params = ebc_config.params_from_config(cfg)
    sc = ebc_scene.SceneConfig.from_config(cfg)
    b = ebc_scene.SceneBatch.from_scenes(
        [ebc_scene.generate_scene(sc, s) for s in shard.scene_seeds(1000, 0, total)], 5, 0)
    env = oracle.OracleEnv(params, total, 5, 0)
    env.reset(b)
    ref = np.stack([env.step(human_policy=_abi.HUMAN_ORCA, robot_policy=_abi.ROBOT_LINEAR)["reward"]
                    for _ in range(steps)], 1)
    np.testing.assert_array_equal(ret["rewards"], ref)
    assert ret["elapsed"] == 1.5                      # max over ranks (0.5, 1.5)
    assert ret["units"] == total * 5 * steps          # sum over ranks


def test_shard_ranges_cover_and_are_disjoint():
    from ebcsim import shard
    for total in (1, 7, 8, 4096, 131072, 131075):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                s, c = shard.shard_range(total, r, world)
                seen += list(range(s, s + c)) if total < 10000 else [(s, c)]
            if total < 10000:
                assert seen == list(range(total))
            else:
                assert sum(c for _, c in seen) == total
                assert all(seen[i][0] + seen[i][1] == seen[i + 1][0] for i in range(world - 1))
    assert shard.weak_range(4096, 3) == (12288, 4096)
