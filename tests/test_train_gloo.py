"""BASELINE config 5 on CPU: two gloo ranks, each with its own replay shard, one flat-bucket
gradient all-reduce per step -> the same parameters as one process on the union batch; value
targets follow Explorer.update_memory; state_dicts are interchangeable with the reference's."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "eb-cadrl_amd")
DIMS = dict(input_dim=13, mlp1_dims=[150, 100], mlp2_dims=[100, 50], mlp3_dims=[150, 100, 100, 1],
            attention_dims=[100, 100, 1])


def _data(seed, n, R=5, T=13):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(n, R, T, generator=g), torch.randn(n, generator=g)


def _worker(rank, world, port, ret):
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    from ebcsim.train import DataParallelTrainer, DeviceReplay, SarlModule
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    model = SarlModule(**DIMS)
    x, y = _data(100 + rank, 32)
    mem = DeviceReplay(32, 5, 13, "cpu")
    mem.push(x, y)

    class All(object):  # deterministic "sample": the whole shard
        def sample(self, bs, generator=None):
            return mem.states, mem.values
    tr = DataParallelTrainer(model, All(), 32, "sgd", 0.01)
    for _ in range(3):
        tr.optimize_batch(1)
    if rank == 0:
        ret["params"] = {k: v.detach().clone() for k, v in model.named_parameters()}
    dist.barrier()
    dist.destroy_process_group()


def test_flat_bucket_allreduce_equals_single_process():
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    from ebcsim.train import DataParallelTrainer, SarlModule
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, 29700 + os.getpid() % 1000, ret), nprocs=2, join=True)
    torch.manual_seed(0)
    model = SarlModule(**DIMS)
    xs, ys = zip(_data(100, 32), _data(101, 32))
    X, Y = torch.cat(xs), torch.cat(ys)

    class All(object):
        def sample(self, bs, generator=None):
            return X, Y
    tr = DataParallelTrainer(model, All(), 64, "sgd", 0.01)
    for _ in range(3):
        tr.optimize_batch(1)
    for k, v in model.named_parameters():
        torch.testing.assert_close(ret["params"][k], v.detach(), atol=2e-6, rtol=1e-5)


def test_state_dict_roundtrip_and_value_targets():
    from helpers import GOLDEN
    from ebcsim.sarl import SarlValueNet
    from ebcsim.train import DeviceReplay, SarlModule, value_targets
    sd = torch.load(os.path.join(GOLDEN, "weights", "sarl_a5_baseline.pth"), map_location="cpu")
    m = SarlModule(**DIMS)
    m.load_reference_state_dict(sd)
    back = m.reference_state_dict()
    assert set(back) == set(sd)
    for k in sd:
        assert torch.equal(back[k], sd[k])
    rows = torch.randn(7, 5, 13)
    net = SarlValueNet(sd)
    torch.testing.assert_close(m(rows), net.forward(rows))
    # explorer.py:171-184
    reward = torch.tensor([0.1, -0.25, 1.0], dtype=torch.float64)
    done = torch.tensor([0, 1, 1], dtype=torch.uint8)
    nxt = torch.randn(3, 5, 13)
    t = value_targets(reward, done, nxt, net, 0.9)
    v = net.forward(nxt).double()
    torch.testing.assert_close(t, torch.stack([reward[0] + 0.9 * v[0], reward[1], reward[2]]))
    mem = DeviceReplay(4, 5, 13, "cpu")
    mem.push(torch.randn(3, 5, 13), torch.arange(3.0))
    mem.push(torch.randn(3, 5, 13), torch.arange(3.0) + 10)   # wraps: capacity 4
    assert len(mem) == 4 and mem.position == 2
    assert sorted(mem.values.tolist()) == [2.0, 10.0, 11.0, 12.0]


@pytest.mark.gpu
def test_collect_and_train_on_device():
    """Roll out on the HIP env with the device SARL policy, fill the replay, take optimizer
    steps: the loss is finite and decreases on the fixed replay."""
    import json
    from helpers import GOLDEN, batch_from_init, load, params_of
    from ebcsim.batched import BatchedEnv
    from ebcsim.sarl import DeviceSarlPolicy
    from ebcsim.train import DataParallelTrainer, DeviceReplay, SarlModule, collect
    z = load("sarl_a5_baseline")
    meta = json.loads(str(z["meta"]))
    E = 64
    b = batch_from_init(z, copies=E)
    env = BatchedEnv(params_of(z), E, b.N, b.S)
    env.reset(b)
    env.use_torch_stream()
    model = SarlModule(**DIMS).to("cuda:0")
    model.load_reference_state_dict(torch.load(os.path.join(GOLDEN, "weights", meta["weights"]), map_location="cpu"))
    target = SarlModule(**DIMS).to("cuda:0")
    target.load_state_dict(model.state_dict())
    pol = DeviceSarlPolicy(model.as_value_net(), z["action_space"], meta["gamma"])
    mem = DeviceReplay(4096, env.R, env.T, "cuda:0")
    g = torch.Generator(device="cuda:0").manual_seed(1)
    mean_r = collect(env, pol, target.as_value_net(), mem, 30, meta["gamma"], epsilon=0.2, generator=g)
    assert len(mem) == 30 * E and np.isfinite(mean_r)
    tr = DataParallelTrainer(model, mem, 100, "sgd", 0.001)
    first = tr.optimize_batch(20, g)
    last = tr.optimize_batch(20, g)
    assert np.isfinite(first) and np.isfinite(last) and last < first * 1.5


def test_imitation_learning_targets_match_reference():
    """Explorer.update_memory(imitation_learning=True) (explorer.py:159-170) on the reference's own IL
    episodes (golden, robot on ORCA): the discounted return from every step on; and the batched form:
    episodes of different lengths following one another, the unfinished tail left out."""
    from helpers import load
    from ebcsim.train import il_value_targets
    for name in ("traj_a5_il_orcasub", "traj_n10_walls_il_orcasub"):
        z = load(name)
        gb = float(z["il_gamma"]) ** (0.25 * float(z["robot_v_pref"]))
        r = torch.tensor(z["reward"])[:, None]
        d = torch.tensor(z["done"].astype(np.uint8))[:, None]
        v, c = il_value_targets(r, d, gb)
        assert bool(c.all())
        np.testing.assert_allclose(v[:, 0].numpy(), z["il_value"], atol=1e-6)
    # two envs: env 0 = episode of 3 steps, then 2 steps, then an unfinished one; env 1 never ends
    r = torch.tensor([[1.0, 0.5], [0.0, 0.5], [2.0, 0.5], [1.0, 0.5], [3.0, 0.5], [7.0, 0.5]], dtype=torch.float64)
    d = torch.tensor([[0, 0], [0, 0], [1, 0], [0, 0], [1, 0], [0, 0]], dtype=torch.uint8)
    v, c = il_value_targets(r, d, 0.5)
    np.testing.assert_allclose(v[:, 0].numpy(), [1 + 0.25 * 2, 0.5 * 2, 2.0, 1 + 0.5 * 3, 3.0, 7.0])
    assert c[:, 0].tolist() == [True, True, True, True, True, False] and not c[:, 1].any()


def test_optimize_epoch_walks_the_memory():
    """Trainer.optimize_epoch (trainer.py:45-72): every stored pair once per epoch; the loss falls."""
    from ebcsim.train import DataParallelTrainer, DeviceReplay, SarlModule
    torch.manual_seed(0)
    model = SarlModule(**DIMS)
    x, y = _data(5, 96)
    mem = DeviceReplay(128, 5, 13, "cpu")
    mem.push(x, 0.1 * y)
    tr = DataParallelTrainer(model, mem, 32, "sgd", 0.01)
    g = torch.Generator().manual_seed(3)
    first = tr.optimize_epoch(1, g)
    last = tr.optimize_epoch(5, g)
    assert np.isfinite(first) and last < first


@pytest.mark.gpu
def test_run_training_imitation_then_rl_on_device():
    """rl/train.py:99-260 in small: the robot on ORCA fills the memory with whole episodes and their
    returns, epochs over it, then epsilon-greedy RL rounds against a target network."""
    import json
    from helpers import GOLDEN, batch_from_init, load, params_of
    from ebcsim.batched import BatchedEnv
    from ebcsim.train import SarlModule, run_training
    z = load("sarl_a5_baseline")
    meta = json.loads(str(z["meta"]))
    E = 64
    b = batch_from_init(z, copies=E)
    env = BatchedEnv(params_of(z), E, b.N, b.S)
    env.reset(b)
    env.use_torch_stream()
    model = SarlModule(**DIMS).to("cuda:0")
    g = torch.Generator(device="cuda:0").manual_seed(1)
    lines = []
    hist = run_training(env, model, z["action_space"], meta["gamma"], il_steps=150, il_epochs=3,
                        il_learning_rate=0.01, il_safety_space=0.15, rl_learning_rate=0.001,
                        train_iterations=3, steps_per_iteration=4, train_batches=5, batch_size=100,
                        capacity=20000, epsilon_start=0.5, epsilon_end=0.1, epsilon_decay=2,
                        target_update_interval=2, generator=g, log=lines.append)
    assert hist["il_episodes"] >= E and hist["il_stored"] > 0
    assert hist["il_loss"] is not None and np.isfinite(hist["il_loss"])
    assert len(hist["rl_loss"]) == 3 and all(np.isfinite(v) for v in hist["rl_loss"])
    assert len(lines) == 4
    # the rollouts decided on the matrix-core blocks, re-packed on the device after every round's optimizer steps
    assert hist["native_refreshes"] == 3 and hist["native_forwards"] >= 3 * 4
    # ... and those blocks hold the FINAL weights: a fresh host pack of the trained module gives the same values
    from ebcsim.sarl import SarlValueNet
    fresh = SarlValueNet({k: v.detach().clone() for k, v in model.state_dict().items()}, device="cuda:0")
    used = model.as_value_net(native=True)
    rows = torch.randn(50, env.R, env.T, device="cuda:0")
    assert torch.equal(fresh.forward(rows), used.forward(rows))


@pytest.mark.gpu
def test_run_training_checkpoints_and_validation(tmp_path):
    """The reference's weight files (il_model.pth, rl_model_<n>.pth, its state_dict layout), the IL stage
    skipped when il_model.pth is there (train.py:113-116), validation rounds through evaluate()."""
    import json
    from helpers import batch_from_init, load, params_of
    from ebcsim.batched import BatchedEnv
    from ebcsim.sarl import SarlValueNet
    from ebcsim.train import SarlModule, run_training
    z = load("sarl_a5_baseline")
    meta = json.loads(str(z["meta"]))
    E = 32
    b = batch_from_init(z, copies=E)
    env = BatchedEnv(params_of(z), E, b.N, b.S)
    env.reset(b)
    env.use_torch_stream()
    val_env = BatchedEnv(params_of(z), 4, b.N, b.S)
    val_env.use_torch_stream()
    val_scenes = batch_from_init(z, copies=4)
    out = str(tmp_path / "out")
    kw = dict(il_learning_rate=0.01, train_iterations=2, steps_per_iteration=3, train_batches=3, capacity=20000,
              epsilon_decay=2, target_update_interval=1, output_dir=out, checkpoint_interval=1, evaluation_interval=1,
              val_env=val_env, val_scenes=val_scenes)
    model = SarlModule(**DIMS).to("cuda:0")
    h1 = run_training(env, model, z["action_space"], meta["gamma"], il_steps=120, il_epochs=1, **kw)
    assert not h1["il_loaded"] and sorted(os.listdir(out)) == ["il_model.pth", "rl_model_1.pth", "rl_model_2.pth"]
    assert len(h1["val"]) == 2 and h1["val"][0][1]["num_episodes"] == 4
    sd = torch.load(os.path.join(out, "rl_model_2.pth"), map_location="cpu")
    assert "mlp1.0.weight" in sd and "attention.4.bias" in sd      # the reference's ValueNetwork keys
    net = SarlValueNet(sd)                                          # loads like a shipped .pth
    torch.testing.assert_close(net.forward(torch.zeros(1, b.N + b.S, 13)), model.cpu()(torch.zeros(1, b.N + b.S, 13)))
    model2 = SarlModule(**DIMS).to("cuda:0")
    env.reset(b)
    h2 = run_training(env, model2, z["action_space"], meta["gamma"], il_steps=120, il_epochs=1, **kw)
    assert h2["il_loaded"] and h2["il_stored"] == 0


@pytest.mark.gpu
def test_run_training_on_device_generated_scenes():
    """The schedule with `scene_gen`: every episode on a scene generated on the device from the train seeds
    (simulator/env.py:153-169: 2000 + episode number), the pool refreshed from fresh seeds between rounds."""
    import configparser
    import json
    from helpers import load, params_of
    from ebcsim import scene as ebc_scene
    from ebcsim.batched import BatchedEnv
    from ebcsim.train import SarlModule, run_training
    z = load("scenes")
    meta = json.loads(str(z["meta_0"]))  # env_adults_5: circle crossing, free map
    cfg = configparser.RawConfigParser()
    cfg.read_string(meta["config_text"])
    sc = ebc_scene.SceneConfig.from_config(cfg)
    gen = ebc_scene.gen_struct(sc, "train")
    zz = load("sarl_a5_baseline")
    E = 32
    env = BatchedEnv(params_of(zz), E, 5, 0)
    env.use_torch_stream()
    model = SarlModule(**DIMS).to("cuda:0")
    hist = run_training(env, model, zz["action_space"], 0.9, il_steps=60, il_epochs=1, train_iterations=3,
                        steps_per_iteration=8, train_batches=2, capacity=20000, epsilon_decay=2, scene_gen=gen,
                        scene_seed0=2000, scene_pool_factor=2)
    assert hist["il_episodes"] > 0 and hist["scene_pools"] >= 3
    # every env now runs a scene of the train sequence: its goals are those of one of the seeds drawn so far
    st = env.get_state()
    drawn = E + 2 * E + 15 * E + hist["scene_pools"] * 2 * E
    ref = env.generate_scenes(gen, 2000, drawn)
    goals = {(float(a), float(b)) for a, b in zip(ref.gx[:, 0], ref.gy[:, 0])}
    assert all((float(st["gx"][e, 0]), float(st["gy"][e, 0])) in goals for e in range(E))
    assert len({float(st["gx"][e, 0]) for e in range(E)}) > E // 2  # and not all the same scene


RCCL_DRIVER = r"""
import json, os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "eb-cadrl_amd")); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import torch, torch.distributed as dist
from helpers import batch_from_init, load, params_of
from ebcsim.batched import BatchedEnv
from ebcsim.train import SarlModule, run_training
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
if os.environ.get("EBCSIM_FORCE_COLLECTIVES") == "1":
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
z = load("sarl_a5_baseline")
E = 32
b = batch_from_init(z, copies=E)
env = BatchedEnv(params_of(z), E, b.N, b.S)
env.reset(b)
env.use_torch_stream()
torch.manual_seed(3)
model = SarlModule(input_dim=13, mlp1_dims=[150, 100], mlp2_dims=[100, 50], mlp3_dims=[150, 100, 100, 1], attention_dims=[100, 100, 1]).to(dev)
g = torch.Generator(device=dev).manual_seed(5)
hist = run_training(env, model, z["action_space"], 0.9, il_steps=80, il_epochs=1, train_iterations=2, steps_per_iteration=3,
                    train_batches=3, capacity=20000, epsilon_decay=2, generator=g)
digest = float(sum(p.detach().double().abs().sum() for p in model.parameters()))
print(json.dumps({"il_loss": hist["il_loss"], "rl_loss": hist["rl_loss"], "digest": digest,
                  "backend": dist.get_backend() if dist.is_initialized() else None}))
if dist.is_initialized():
    dist.destroy_process_group()
"""


@pytest.mark.gpu
def test_schedule_through_rccl_on_one_rank(tmp_path):
    """Every collective of the schedule (the flat gradient all-reduce, the MIN all-reduce of the 'has data' flag, the
    broadcasts of the start) through RCCL itself: a one-rank nccl process group with EBCSIM_FORCE_COLLECTIVES=1 on
    the GPU must train exactly what the run without a process group trains (a sum over one rank is the identity)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "rccl_driver.py"
    script.write_text(RCCL_DRIVER)
    outs = []
    for force in ("0", "1"):
        env = dict(os.environ, EBCSIM_FORCE_COLLECTIVES=force, MASTER_ADDR="127.0.0.1", MASTER_PORT="29731",
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        r = subprocess.run([sys.executable, str(script), root], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(json.loads(r.stdout.strip().splitlines()[-1]))
    assert outs[0]["backend"] is None and outs[1]["backend"] == "nccl"
    # (approx: two GPU training runs are not bit-reproducible to begin with)
    assert outs[1]["il_loss"] == pytest.approx(outs[0]["il_loss"], rel=1e-4)
    assert outs[1]["rl_loss"] == pytest.approx(outs[0]["rl_loss"], rel=1e-3)
    assert outs[1]["digest"] == pytest.approx(outs[0]["digest"], rel=1e-5)


def test_episode_store_keeps_successes_and_collisions_only():
    """explorer.py:82-92: pairs reach the memory when their episode ends, and only for ReachGoal or a
    collision; a timeout's pairs are dropped; an unfinished episode keeps waiting."""
    from ebcsim import _abi
    from ebcsim.train import DeviceReplay, EpisodeStore, il_value_targets
    E, R, T = 3, 2, 13
    store, mem = EpisodeStore(E, 6, R, T, "cpu"), DeviceReplay(64, R, T, "cpu")
    tag = lambda t: torch.full((E, R, T), float(t)) + torch.arange(E)[:, None, None] * 100.0  # noqa: E731
    infos = [
        [_abi.INFO_NOTHING, _abi.INFO_DANGER, _abi.INFO_NOTHING],
        [_abi.INFO_REACH_GOAL, _abi.INFO_NOTHING, _abi.INFO_NOTHING],      # env 0 ends well after 2 steps
        [_abi.INFO_NOTHING, _abi.INFO_TIMEOUT, _abi.INFO_NOTHING],         # env 1 times out after 3 steps
        [_abi.INFO_COLLISION_CHILD, _abi.INFO_NOTHING, _abi.INFO_NOTHING],  # env 0's second episode: 2 steps
    ]
    pushed = []
    for t, inf in enumerate(infos):
        inf = torch.tensor(inf, dtype=torch.uint8)
        done = ((inf == _abi.INFO_REACH_GOAL) | (inf >= _abi.INFO_COLLISION_OBSTACLE)).to(torch.uint8)
        store.add(tag(t), torch.full((E,), float(t)))
        pushed.append(store.end(done, inf, mem))
    assert pushed == [0, 2, 0, 2] and len(mem) == 4
    assert sorted(mem.states[:4, 0, 0].tolist()) == [0.0, 1.0, 2.0, 3.0]   # env 0's four steps, nobody else's
    assert store.length.tolist() == [0, 1, 4]  # env 1 dropped its timeout and started over; env 2 still waiting
    # the same filter in the windowed IL targets
    r = torch.ones((4, E), dtype=torch.float64)
    inf = torch.tensor(infos, dtype=torch.uint8)
    done = ((inf == _abi.INFO_REACH_GOAL) | (inf >= _abi.INFO_COLLISION_OBSTACLE)).to(torch.uint8)
    v, keep = il_value_targets(r, done, 0.5, inf)
    assert keep[:, 0].all() and not keep[:, 1].any() and not keep[:, 2].any()
    np.testing.assert_allclose(v[:, 0].numpy(), [1.5, 1.0, 1.5, 1.0])


def _epoch_worker(rank, world, port, ret):
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    from ebcsim.train import DataParallelTrainer, DeviceReplay, SarlModule
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    model = SarlModule(**DIMS)
    n = 64 + 32 * rank  # shards of different sizes: every rank must take the same number of batches
    x, y = _data(200 + rank, n)
    mem = DeviceReplay(128, 5, 13, "cpu")
    mem.push(x, 0.1 * y)
    tr = DataParallelTrainer(model, mem, 32, "sgd", 0.01)
    loss = tr.optimize_epoch(2, torch.Generator().manual_seed(rank))
    ret[rank] = (loss, {k: v.detach().clone() for k, v in model.named_parameters()})
    dist.barrier()
    dist.destroy_process_group()


def test_optimize_epoch_two_ranks_unequal_shards():
    """The IL epochs under two gloo ranks with replay shards of different sizes: no rank waits for a
    batch the other does not have, and the averaged gradients keep the replicas identical."""
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_epoch_worker, args=(2, 29900 + os.getpid() % 1000, ret), nprocs=2, join=True)
    assert np.isfinite(ret[0][0]) and np.isfinite(ret[1][0])
    for k, v in ret[0][1].items():
        torch.testing.assert_close(ret[1][1][k], v, atol=1e-7, rtol=1e-6)


def test_replay_push_larger_than_capacity_keeps_pairs_consistent():
    """collect_il pushes steps x E pairs in one call (4096 envs x 50 steps against the default capacity):
    a ring keeps the newest `capacity` of them, each state still next to ITS value and row count."""
    from ebcsim.train import DeviceReplay
    cap, n, R, T = 16, 45, 3, 13
    mem = DeviceReplay(cap, R, T, "cpu")
    mem.push(torch.zeros(5, R, T), torch.zeros(5))      # something to overwrite, position 5
    tag = torch.arange(n, dtype=torch.float32)
    mem.push(tag[:, None, None].expand(n, R, T).clone(), tag, (tag % R + 1).to(torch.int64))
    assert len(mem) == cap and mem.position == (5 + n) % cap and mem.ragged
    assert sorted(mem.values.tolist()) == [float(v) for v in range(n - cap, n)]
    assert torch.equal(mem.states[:, 0, 0], mem.values)                       # pair by pair
    assert torch.equal(mem.n_valid, (mem.values % R + 1).to(torch.int64))
    s, v, rows = mem.sample(8, torch.Generator().manual_seed(0), with_rows=True)
    assert torch.equal(s[:, 0, 0], v) and torch.equal(rows, (v % R + 1).to(torch.int64))
    # the newest item sits just before the write position, like after item-by-item pushes
    assert mem.values[(mem.position - 1) % cap] == n - 1


def _schedule_worker(rank, world, port, ret):
    for p in (ROOT, PKG, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import json
    import torch.distributed as dist
    from ebcsim import _abi
    from ebcsim.train import SarlModule, run_training
    from helpers import CpuDeviceEnv, batch_from_init, load, params_of
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    z = load("sarl_a5_baseline")
    meta = json.loads(str(z["meta"]))
    params = params_of(z)
    b = batch_from_init(z, copies=2)
    # rank 0: the robot stands on its goal (one action: stay) -> every step is a ReachGoal episode: its
    # shard has data after round 0.  rank 1: one action (0.15 m straight up per step), the robot 0.45 m
    # below its goal (radius 0.2) -> ReachGoal at its second step: empty after round 0, filled after round 1.
    gx, gy = b.robot[0, 5], b.robot[0, 6]
    b.robot[:, 0], b.robot[:, 1] = gx, (gy if rank == 0 else gy - 0.45)
    actions = np.array([[0.0, 0.0]]) if rank == 0 else np.array([[0.0, 0.6]])
    env = CpuDeviceEnv(params, 2, b.N, b.S)
    env.reset(b)
    torch.manual_seed(1234 + rank)  # different constructors' draws: the schedule must broadcast rank 0's weights
    model = SarlModule(**DIMS)
    hist = run_training(env, model, actions, meta["gamma"], il_steps=0, train_iterations=3, steps_per_iteration=1,
                        train_batches=2, batch_size=4, capacity=64, epsilon_start=0.0, epsilon_end=0.0,
                        target_update_interval=2, generator=torch.Generator().manual_seed(rank), rank=rank)
    ret[rank] = (hist["rl_loss"], {k: v.detach().clone() for k, v in model.named_parameters()})
    dist.barrier()
    dist.destroy_process_group()


def test_run_training_two_ranks_fill_their_shards_at_different_rounds():
    """rl/train.py:239-259 under two gloo ranks whose replay shards become non-empty in different rounds:
    a round's optimizer steps (gradient all-reduces) are taken by both ranks or by neither, the replicas
    start from rank 0's weights and stay identical."""
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_schedule_worker, args=(2, 29300 + os.getpid() % 500, ret), nprocs=2, join=True)
    l0, l1 = ret[0][0], ret[1][0]
    assert np.isnan(l0[0]) and np.isnan(l1[0])            # round 0: rank 1's shard is still empty -> nobody trains
    assert np.isfinite(l0[1]) and np.isfinite(l1[1])      # round 1: both have data -> both train
    assert np.isfinite(l0[2]) and np.isfinite(l1[2])
    for k, v in ret[0][1].items():
        torch.testing.assert_close(ret[1][1][k], v, atol=1e-7, rtol=1e-6)


def _rl_memory_against_reference(make_env, device):
    """explorer.py:171-184 + :82-92 (imitation_learning=False): one greedy SARL episode rolled by collect()
    puts in the memory exactly the (state, value) pairs the reference's Explorer.update_memory put in its
    ReplayMemory for the same episode (golden from the reference's own policy, explorer and memory)."""
    import json
    from ebcsim import _abi
    from ebcsim.sarl import DeviceSarlPolicy, SarlValueNet
    from ebcsim.train import DeviceReplay, EpisodeStore, collect, value_targets
    from helpers import GOLDEN, batch_from_init, load, params_of
    z, g = load("sarl_a5_baseline"), load("sarl_a5_rl_memory")
    meta = json.loads(str(z["meta"]))
    params = params_of(z)
    b = batch_from_init(z, copies=1)
    env = make_env(params, 1, b.N, b.S)
    env.reset(b)
    env.use_torch_stream()
    net = SarlValueNet.load(os.path.join(GOLDEN, "weights", meta["weights"]), device=device)
    policy = DeviceSarlPolicy(net, z["action_space"], float(g["gamma"]))
    n = len(g["rl_value"])
    assert n == len(z["action"])
    mem = DeviceReplay(4 * n, env.R, env.T, device)
    store = EpisodeStore(1, n + 2, env.R, env.T, device)
    collect(env, policy, net, mem, n, float(g["gamma"]), epsilon=0.0, human_policy=_abi.HUMAN_ORCA, store=store)
    assert len(mem) == n, (len(mem), n)  # the whole episode, pushed at its end (ReachGoal) and nothing else
    np.testing.assert_allclose(mem.states[:n].cpu().numpy(), g["rl_state"], atol=1e-5)
    np.testing.assert_allclose(mem.values[:n].cpu().numpy(), g["rl_value"], atol=5e-5)
    # value_targets alone on the reference's own states: terminal -> reward, else reward + gamma_bar * V(next)
    st = torch.from_numpy(g["rl_state"]).to(device)
    rw = torch.from_numpy(g["reward"]).to(device)
    done = torch.zeros(n, dtype=torch.uint8, device=device)
    done[-1] = 1
    nxt = torch.cat([st[1:], st[-1:]])
    gb = float(g["gamma"]) ** (float(g["time_step"]) * float(g["robot_v_pref"]))
    t = value_targets(rw, done, nxt, net, gb)
    np.testing.assert_allclose(t.cpu().numpy(), g["rl_value"], atol=5e-5)
    assert t[-1] == rw[-1]


def test_rl_mode_memory_equals_the_reference_explorer_cpu():
    from helpers import CpuDeviceEnv
    _rl_memory_against_reference(CpuDeviceEnv, "cpu")


@pytest.mark.gpu
def test_rl_mode_memory_equals_the_reference_explorer_gpu():
    from ebcsim.batched import BatchedEnv
    _rl_memory_against_reference(lambda p, E, N, S: BatchedEnv(p, E, N, S), "cuda:0")


def test_optimizer_steps_equal_the_reference_trainer():
    """rl/utils/trainer.py:74-100: three SGD (momentum 0.9) steps of the reference's own Trainer on the golden
    RL memory, all pairs per batch, from the shipped weights — this trainer's losses and every parameter
    afterwards are the reference's (golden: the reference's Trainer + ValueNetwork + ReplayMemory)."""
    from helpers import GOLDEN, load
    from ebcsim.train import DataParallelTrainer, DeviceReplay, SarlModule
    g, t = load("sarl_a5_rl_memory"), load("sarl_a5_trainer_steps")
    sd = torch.load(os.path.join(GOLDEN, "weights", "sarl_a5_baseline.pth"), map_location="cpu")
    model = SarlModule(**DIMS)
    model.load_state_dict(sd)
    n = len(g["rl_value"])
    mem = DeviceReplay(n, 5, 13, "cpu")
    mem.push(torch.from_numpy(g["rl_state"]), torch.from_numpy(g["rl_value"]))

    class All(object):  # the whole memory as the batch, like the golden run
        def sample(self, bs, generator=None):
            return mem.states[:n], mem.values[:n]
    tr = DataParallelTrainer(model, All(), n, "sgd", float(t["lr"]))
    losses = [tr.optimize_batch(1) for _ in range(int(t["steps"]))]
    np.testing.assert_allclose(losses, t["loss"], rtol=2e-5)
    after = model.state_dict()
    for k in sd:
        np.testing.assert_allclose(after[k].detach().numpy(), t["p_" + k], atol=2e-7, rtol=2e-5, err_msg=k)
        assert not np.array_equal(t["p_" + k], sd[k].numpy()) or k.endswith("bias")  # the steps moved the weights


@pytest.mark.gpu
def test_il_targets_kernel_equals_the_host_scan():
    """ebc_il_targets (one thread per env, backwards over the window) against the torch scan it replaces on the
    device path — itself pinned on the reference's Explorer memory (test_imitation_learning_targets_match_reference)."""
    from ebcsim import _abi
    from ebcsim.train import il_value_targets
    g = torch.Generator().manual_seed(3)
    T, E = 57, 333
    r = torch.rand(T, E, generator=g, dtype=torch.float64) - 0.3
    codes = torch.tensor([_abi.INFO_NOTHING, _abi.INFO_DANGER, _abi.INFO_REACH_GOAL, _abi.INFO_COLLISION_ADULT,
                          _abi.INFO_COLLISION_CHILD, _abi.INFO_TIMEOUT, _abi.INFO_COLLISION_OBSTACLE], dtype=torch.uint8)
    pick = torch.randint(0, 40, (T, E), generator=g)
    info = torch.where(pick < len(codes), codes[pick.clamp(max=len(codes) - 1)], torch.zeros((), dtype=torch.uint8))
    done = ((info == _abi.INFO_REACH_GOAL) | (info >= _abi.INFO_COLLISION_OBSTACLE)).to(torch.uint8)
    v_host, k_host = il_value_targets(r, done, 0.93, info)              # CPU tensors: the torch scan
    v_dev, k_dev = il_value_targets(r.cuda(), done.cuda(), 0.93, info.cuda())  # CUDA tensors: the kernel
    assert torch.equal(k_dev.cpu(), k_host)
    torch.testing.assert_close(v_dev.cpu(), v_host, atol=1e-12, rtol=0)


def _collect_il_reproduces_reference_memory(make_env, device, name):
    """The reference's whole IL stage on one il_policy (Explorer.run_k_episodes(k, "train", update_memory=True,
    imitation_learning=True), rl/train.py:130-133) against ONE collect_il call: env 0 = that policy object, episode 0 by
    reset, the following episodes as its scene pool (restarts inside ebc_step_k), the persistent simulator of
    simulator/policy/orca.py:96-133 inside the kernel.  The replay memory ends up with the reference's states and
    discounted returns, in its order (timeouts left out, explorer.py:82-92)."""
    from ebcsim.scene import SceneBatch
    from ebcsim.train import DeviceReplay, collect_il
    from helpers import il_persistent_episodes
    z, params, N, S, batches = il_persistent_episodes(name)
    K = len(batches)
    env = make_env(params, 1, N, S)
    env.reset(batches[0])
    pool = SceneBatch(K - 1, N, S, *[None if getattr(batches[0], f) is None and f == "grid" else
                                     np.concatenate([(getattr(b, f) if getattr(b, f) is not None else
                                                      np.zeros((1, env_G(params), 2), np.uint64)) for b in batches[1:]], 0)
                                     for f in ("n_humans", "px", "py", "vx", "vy", "gx", "gy", "radius", "v_pref", "type",
                                               "n_static", "spx", "spy", "sradius", "grid", "robot")])
    env.set_scene_pool(pool, stride=1)
    env.use_torch_stream()
    steps = sum(len(z["action%d" % k]) for k in range(K))
    R, T = N + S, z["il_state"].shape[1]
    mem = DeviceReplay(steps + 8, R, T, device)
    stored, episodes = collect_il(env, mem, steps, float(z["il_gamma"]), float(z["safety_space"]))
    assert episodes == K and stored == len(z["il_value"]) == len(z["il_state_rows"])
    np.testing.assert_allclose(mem.values[:stored].cpu().numpy(), z["il_value"], atol=1e-6, rtol=0)
    rows = z["il_state_rows"]
    np.testing.assert_array_equal(mem.n_valid[:stored].cpu().numpy(), rows)
    off = np.concatenate([[0], np.cumsum(rows)])
    st = mem.states[:stored].cpu().numpy()
    for q in range(stored):
        np.testing.assert_allclose(st[q, :rows[q]], z["il_state"][off[q]:off[q + 1]], atol=1e-5, rtol=1e-5, err_msg=str(q))
        assert not st[q, rows[q]:].any()


def env_G(params):
    return int(round(params.map_size_m / params.map_resolution))


@pytest.mark.parametrize("name", ["il_persistent_const_rows", "il_persistent_wall_rows"])
def test_collect_il_reproduces_the_reference_il_stage_cpu(name):
    from helpers import CpuDeviceEnv
    _collect_il_reproduces_reference_memory(CpuDeviceEnv, "cpu", name)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["il_persistent_const_rows", "il_persistent_wall_rows"])
def test_collect_il_reproduces_the_reference_il_stage_gpu(name):
    from ebcsim.batched import BatchedEnv
    _collect_il_reproduces_reference_memory(lambda p, E, N, S: BatchedEnv(p, E, N, S), "cuda:0", name)
