"""Robot decisions = look-ahead sweep + SARL value network.  Golden: the
reference's own SARL policy (its shipped weights) driving full episodes, 81 action values per
decision (tests/golden/sarl_*.npz, humans on the oracle-substituted rvo2).  Values are float32
network outputs on O(1) numbers: 5e-5 absolute (observed <= 1.5e-5 with the split-bf16 matrix-core
blocks); EVERY decision must pick the reference's action (counted per episode)."""
import json
import os

import numpy as np
import pytest
import torch

from ebcsim import _abi
from ebcsim.sarl import DeviceSarlPolicy, SarlValueNet
from helpers import GOLDEN, batch_from_init, load, params_of

# the last three are the reference's remaining known-answer runs (tests/run_tests.py:23-41): test_basic_simulation on
# two more configs (bicycles; bicycles + a static map) and test_scene_simulation on the frozen 10-obstacle scene
RUNS = ["sarl_a5_baseline", "sarl_n10_ebcadrl", "sarl_a3b3s2_baseline", "sarl_a3b3_baseline",
        "sarl_scene_a3b3s10_baseline"]
TOL = 5e-5


def _check_values(values, z, t):
    """-> 1 when the decision picks the reference's action"""
    ref = z["values"][t]
    np.testing.assert_allclose(values, ref, atol=TOL, rtol=0, err_msg="decision %d" % t)
    best = int(np.argmax(values))
    chosen = int(np.where((z["action_space"] == z["action"][t]).all(1))[0][0])
    return int(best == chosen)


@pytest.mark.parametrize("name", RUNS)
def test_sarl_values_cpu(name):
    """SarlValueNet (torch, CPU) on the oracle's look-ahead rows."""
    from oracle import oracle
    z = load(name)
    meta = json.loads(str(z["meta"]))
    params = params_of(z)
    b = batch_from_init(z)
    env = oracle.OracleEnv(params, 1, b.N, b.S)
    env.reset(b)
    net = SarlValueNet.load(os.path.join(GOLDEN, "weights", meta["weights"]))
    pol = DeviceSarlPolicy(net, z["action_space"], meta["gamma"])
    v_pref = float(b.robot[0, 7])
    steps = len(z["action"]) if name in RUNS[2:] else min(len(z["action"]), 40)  # the known-answer runs: to the goal
    agree = decided = 0
    for t in range(steps):
        la = env.lookahead(z["action_space"], human_policy=_abi.HUMAN_ORCA)
        if not np.isnan(z["values"][t]).any():
            vals = pol.values_from(torch.from_numpy(la["rows_rotated"]), torch.from_numpy(la["reward"]),
                                   None, params.time_step, v_pref)
            agree += _check_values(vals[0].numpy(), z, t)
            decided += 1
        out = env.step(robot_action=z["action"][t][None], human_policy=_abi.HUMAN_CACHED)
        assert int(out["info"][0]) == int(z["info"][t])
    assert agree == decided, (agree, decided)
    if steps == len(z["action"]):  # pass criterion of the reference's test: terminal class ReachGoal
        assert bool(out["done"][0]) and int(out["info"][0]) == _abi.INFO_REACH_GOAL


@pytest.mark.gpu
@pytest.mark.parametrize("name", RUNS)
def test_sarl_decisions_gpu(name):
    """The whole decision on device: ebc_lookahead -> torch GEMMs -> argmax -> cached step."""
    from ebcsim.batched import BatchedEnv
    z = load(name)
    meta = json.loads(str(z["meta"]))
    params = params_of(z)
    E = 5
    b = batch_from_init(z, copies=E)
    env = BatchedEnv(params, E, b.N, b.S)
    env.reset(b)
    env.use_torch_stream()
    net = SarlValueNet.load(os.path.join(GOLDEN, "weights", meta["weights"]), device="cuda:0")
    pol = DeviceSarlPolicy(net, z["action_space"], meta["gamma"])
    outs = env.alloc_step_outputs(("reward", "done", "info"))
    agree = decided = 0
    for t in range(len(z["action"])):
        actions, values = pol.decide(env)
        torch.cuda.synchronize()
        v = values.cpu().numpy()
        assert (np.abs(v - v[0:1]) < 1e-5).all()
        if not np.isnan(z["values"][t]).any():
            agree += _check_values(v[0], z, t)
            decided += 1
        forced = torch.tensor(np.tile(z["action"][t], (E, 1)), dtype=torch.float64, device="cuda:0")
        env.step_device(outs, robot_action=forced, human_policy=_abi.HUMAN_CACHED)
        torch.cuda.synchronize()
        assert int(outs["info"][0]) == int(z["info"][t]), t
        np.testing.assert_allclose(float(outs["reward"][0]), z["reward"][t], atol=1e-9)
    assert int(z["info"][-1]) == _abi.INFO_REACH_GOAL
    assert agree == decided, (agree, decided)  # every decision of the episode picks the reference's action
    # the values above came from the HIP value-network kernels, not from a torch stand-in
    assert net._native_blocks(), "the value network did not run on the MFMA blocks"
    assert getattr(net, "native_forwards", 0) >= decided


def _ragged_pool_decisions(make_env, device, check_rows):
    """A scene pool whose scenes differ in size: after restarts the rows the policy masks with must be the
    CURRENT scene's (read from the device state at every decision), and its values must equal the network
    run on exactly the rows that exist."""
    z = load("sarl_a5_baseline")
    meta = json.loads(str(z["meta"]))
    params = params_of(z)
    params.time_limit = 1  # every env times out at step 4: restarts come quickly
    E = 4
    b = batch_from_init(z, copies=E)
    pool = batch_from_init(z, copies=2 * E)
    for c in range(2 * E):
        pool.n_humans[c] = 5 - (c % 3)       # 5, 4, 3 humans
        pool.px[c, pool.n_humans[c]:] = 0
    env = make_env(params, E, b.N, b.S)
    env.reset(b)
    env.set_scene_pool(pool, stride=E)
    env.use_torch_stream()
    net = SarlValueNet.load(os.path.join(GOLDEN, "weights", meta["weights"]), device=device)
    pol = DeviceSarlPolicy(net, z["action_space"], meta["gamma"])
    outs = env.alloc_step_outputs(("reward", "done", "info"))
    seen = set()
    for t in range(14):
        actions, values = pol.decide(env)
        rows = check_rows(env)
        assert env.ragged and pol.n_valid is not None
        np.testing.assert_array_equal(pol.n_valid.cpu().numpy(), rows)
        seen.update(rows.tolist())
        # the masked batch forward == the network on the rows that exist, env by env
        rr = pol._bufs["rows_rotated"]
        for e in range(E):
            alone = net.forward(rr[e, :, :int(rows[e])].contiguous()).to(torch.float64)
            full = (values[e] - pol._bufs["reward"][e]) / (meta["gamma"] ** (params.time_step * pol._v_pref))
            np.testing.assert_allclose(full.cpu().numpy(), alone.cpu().numpy(), atol=2e-5)
        env.step_device(outs, robot_action=actions.contiguous(), human_policy=_abi.HUMAN_CACHED, flags=_abi.FLAG_AUTO_RESET)
    assert seen == {3, 4, 5}  # the walk went through scenes of every size


def test_ragged_pool_row_counts_follow_restarts_cpu():
    from helpers import CpuDeviceEnv
    _ragged_pool_decisions(CpuDeviceEnv, "cpu", lambda env: env._o.row_counts())


@pytest.mark.gpu
def test_ragged_pool_row_counts_follow_restarts_gpu():
    from ebcsim.batched import BatchedEnv
    _ragged_pool_decisions(lambda p, E, N, S: BatchedEnv(p, E, N, S), "cuda:0", lambda env: env.row_counts())


@pytest.mark.gpu
def test_evaluate_reports_the_reference_metrics():
    """Explorer.run_k_episodes' statistics (explorer.py:202-330) from one batched pass: the shipped SARL
    weights on the golden episode's scene reach the goal in the golden episode's time, with its
    discounted reward; and the imitation-learning demonstrator through the same function."""
    from ebcsim.batched import BatchedEnv
    from ebcsim.train import evaluate
    z = load("sarl_a5_baseline")
    meta = json.loads(str(z["meta"]))
    params = params_of(z)
    E = 6
    b = batch_from_init(z, copies=E)
    env = BatchedEnv(params, E, b.N, b.S)
    env.reset(b)
    env.use_torch_stream()
    net = SarlValueNet.load(os.path.join(GOLDEN, "weights", meta["weights"]), device="cuda:0")
    pol = DeviceSarlPolicy(net, z["action_space"], meta["gamma"])
    m = evaluate(env, lambda e: pol.decide(e)[0], meta["gamma"], human_policy=_abi.HUMAN_CACHED)
    T = len(z["action"])
    assert m["success_rate"] == 1.0 and m["success"] == E and m["timeout"] == 0 and m["num_episodes"] == E
    assert abs(m["avg_nav_time"] - T * params.time_step) <= 2 * params.time_step
    gb = meta["gamma"] ** (params.time_step * float(b.robot[0, 7]))
    ref_total = float(sum(gb ** t * r for t, r in enumerate(z["reward"])))
    assert abs(m["total_reward:"] - ref_total) < 0.05
    danger_steps = int((z["info"] == _abi.INFO_DANGER).sum())
    if abs(m["avg_nav_time"] - T * params.time_step) < 1e-9:
        np.testing.assert_allclose(m["Frequency of being in danger"], danger_steps / T, atol=1e-9)
    # the demonstrator (robot on ORCA): every episode ends, the rates add up to one
    env.reset(b)
    act = torch.zeros((E, 2), dtype=torch.float64, device="cuda:0")

    def demo(e):
        e.robot_orca_device(act, 0.15)
        return act
    d = evaluate(env, demo, meta["gamma"])
    total = (d["success_rate"] + d["collision_rate_adult"] + d["collision_rate_bicycle"] + d["collision_rate_child"]
             + d["collision_rate_obstacle"] + d["timeout"] / E)
    assert abs(total - 1.0) < 1e-12


@pytest.mark.gpu
def test_decisions_are_the_float32_decisions_on_the_bench_workload():
    """Where the split-bf16 error is largest (the shipped eb-cadrl weights: large attention scores) and the batch is the
    bench's: 1024 envs x 81 actions x 18 rows at three points of the episodes.  (a) The matrix-core values stay within
    the bound the refinement is built on (SarlValueNet.COARSE_EPS); (b) after the bound-driven refinement EVERY env
    takes the float32 network's action (multi_human_rl.py:61-80 maximises float32 values) and (c) the refined values
    are float32-GEMM-grade (2e-6 of torch's float32 GEMMs); (d) the refinement ran inside the library
    (ebc_mlp2_forward_f32), not through torch."""
    import bench
    from ebcsim import actions as ebc_actions
    from ebcsim.batched import BatchedEnv
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
    net = SarlValueNet.load(os.path.join(GOLDEN, "weights", "sarl_n10_ebcadrl.pth"), device=str(dev))
    net64 = SarlValueNet.load(os.path.join(GOLDEN, "weights", "sarl_n10_ebcadrl.pth"), device=str(dev), dtype=torch.float64)
    bufs = env.alloc_lookahead_outputs(A, ("reward", "done", "info", "rows_rotated"))
    worst_coarse = worst_refined = worst_native64 = worst_torch64 = 0.0
    for point in range(3):
        for _ in range(10 * point):
            env.step_device(outs, human_policy=_abi.HUMAN_ORCA, robot_policy=_abi.ROBOT_LINEAR, flags=_abi.FLAG_AUTO_RESET)
        env.lookahead_device(acts, bufs, human_policy=_abi.HUMAN_ORCA)
        rows, reward = bufs["rows_rotated"], bufs["reward"]
        net.native_exact = False  # the yard-stick: torch's float32 GEMMs
        f32 = torch.empty((E, A), dtype=torch.float32, device=dev)
        for e0 in range(0, E, 128):
            f32[e0:e0 + 128] = net.forward(rows[e0:e0 + 128].reshape(-1, env.R, env.T), exact=True).view(-1, A)
        net.native_exact = True
        want = reward + 0.9 * f32.double()
        coarse = net.action_values(rows, reward, 0.9, refine=0)
        before = getattr(net, "native_exact_forwards", 0)
        got = net.action_values(rows, reward, 0.9)
        torch.cuda.synchronize()
        assert net.native_exact_forwards > before
        worst_coarse = max(worst_coarse, float((coarse - want).abs().max()) / 0.9)
        changed = got != coarse
        worst_refined = max(worst_refined, float((got - want).abs()[changed].max()))
        # the yard-stick of both float32 forms: the same network in float64 on the refined candidates
        ei, ai = torch.nonzero(changed, as_tuple=True)
        truth = reward[ei, ai] + 0.9 * net64.forward(rows[ei, ai]).double()
        worst_native64 = max(worst_native64, float((got[ei, ai] - truth).abs().max()))
        worst_torch64 = max(worst_torch64, float((want[ei, ai] - truth).abs().max()))
        assert int((got.argmax(1) == want.argmax(1)).sum()) == E, point
    assert worst_coarse <= net.coarse_eps, (worst_coarse, net.coarse_eps)  # the measured bound the refinement set is built on
    # float32-GEMM-grade: as close to the float64 network as torch's float32 GEMMs are (two float32 evaluations of this
    # network differ by ~2e-6 from one another: its attention scores are large)
    assert worst_native64 <= max(1.5 * worst_torch64, 2e-6), (worst_native64, worst_torch64)
    assert worst_refined <= 5e-6, worst_refined
    st = net.refine_stats
    assert st["capped"] == 0 and st["decisions"] == 3 * E and st["bound_violations"] == 0
    print("refinement: %.2f candidates per decision, %d of %d decisions with more than 2, largest set %d; coarse error %.2e; "
          "refined vs torch float32 %.2e; vs the float64 network: native %.2e, torch float32 %.2e" % (
              st["candidates"] / st["decisions"], st["over2"], st["decisions"], st["max_set"], worst_coarse, worst_refined,
              worst_native64, worst_torch64))


@pytest.mark.gpu
def test_chunks_on_two_streams_give_the_values_of_one_stream():
    """action_values spreads the chunks of a decision batch over two streams (one chunk's small per-pair kernels beside
    the other's wide ones): every value must be the value of the single-stream pass, bit for bit — a pair's result does
    not depend on where its rows lie in a batch — and the caller's stream must see them complete."""
    import torch
    from ebcsim.sarl import SarlValueNet
    dev = torch.device("cuda", 0)
    net = SarlValueNet.load(os.path.join(GOLDEN, "weights", "sarl_n10_ebcadrl.pth"), device=str(dev))
    g = torch.Generator().manual_seed(21)
    E, A, R, T = 37, 81, 18, 17
    rows = torch.randn(E, A, R, T, generator=g).to(dev)
    rows[..., 13:] = 0
    rows[..., 13] = 1
    reward = torch.randn(E, A, generator=g, dtype=torch.float64).to(dev)
    nv = torch.randint(12, R + 1, (E,), generator=g).to(dev)
    out = {}
    for streams in (1, 2):
        net.CHUNK_STREAMS = streams
        for _ in range(2):  # the second pass re-uses the streams and the allocator's blocks of the first
            out[streams] = net.action_values(rows, reward, 0.9, n_valid=nv, refine=0, chunk_pairs=5 * A)
    assert torch.equal(out[1], out[2])
    assert bool(torch.isfinite(out[2]).all())


def test_batches_whose_robots_differ_in_v_pref_are_refused():
    """One action space and one discount gamma^(dt v_pref) per batch (the reference has them per robot,
    multi_human_rl.py:36-60, 72-76): a batch of robots with different v_pref must not silently take env 0's."""
    from ebcsim.sarl import uniform_v_pref

    class Env:
        def __init__(self, v):
            self.v = v

        def get_state(self):
            robot = np.zeros((len(self.v), 9))
            robot[:, 7] = self.v
            return {"robot": robot}
    assert uniform_v_pref(Env([1.0, 1.0, 1.0])) == 1.0
    with pytest.raises(ValueError, match="v_pref"):
        uniform_v_pref(Env([1.0, 1.2, 1.0]))


@pytest.mark.gpu
def test_decision_rank_kernel_against_torch():
    """ebc_decision_rank (one launch: values, every env's actions by value, the size of the near-best set) against the
    torch expressions it replaces: the same float64 values bit for bit, a permutation per env that sorts them (equal
    values: the lower action first), the same counts — with exact ties in the rows."""
    import torch
    from ebcsim import _capi
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(5)
    for E, A in ((1024, 81), (3, 1), (17, 200)):
        v = torch.randn(E, A, generator=g)
        v[:, A // 2] = v[:, 0]                      # an exact tie in every row
        reward = torch.randn(E, A, generator=g, dtype=torch.float64) * 0.1
        reward[:, A // 2] = reward[:, 0]
        v, reward = v.to(dev), reward.to(dev)
        discount, bound = 0.9 ** 0.25, 0.05
        values = torch.empty((E, A), dtype=torch.float64, device=dev)
        order = torch.empty((E, A), dtype=torch.int32, device=dev)
        count = torch.empty((E,), dtype=torch.int32, device=dev)
        _capi.check(_capi.lib().ebc_decision_rank(torch.cuda.current_stream(dev).cuda_stream, v.data_ptr(), reward.data_ptr(), discount, bound,
                                                  E, A, values.data_ptr(), order.data_ptr(), count.data_ptr()))
        ref = reward + discount * v.to(torch.float64)
        assert torch.equal(values, ref)
        o = order.to(torch.int64)
        assert torch.equal(o.sort(1).values, torch.arange(A, device=dev).expand(E, A))
        ranked = ref.gather(1, o)
        assert bool((ranked[:, 1:] <= ranked[:, :-1]).all())
        tie = ranked[:, 1:] == ranked[:, :-1]
        assert bool((o[:, 1:][tie] > o[:, :-1][tie]).all()) and int(tie.sum()) >= (E if A > 1 else 0)
        assert torch.equal(count.to(torch.int64), (ref >= (ref.max(1, keepdim=True).values - bound)).sum(1))


@pytest.mark.gpu
def test_decision_apply_kernel_against_torch():
    """ebc_decision_apply against the torch expressions it replaces: the candidates' values written where they belong
    (bit for bit reward + discount * exact), everything else untouched, and the largest |exact - coarse| among them."""
    import torch
    from ebcsim import _capi
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(6)
    E, A = 300, 81
    v = torch.randn(E, A, generator=g).to(dev)
    reward = (torch.randn(E, A, generator=g, dtype=torch.float64) * 0.1).to(dev)
    for n in (1, 77, 5000):
        flat = torch.randperm(E * A, generator=g)[:n].sort().values.to(dev)
        env_i, act_i = flat // A, flat % A
        exact = (v[env_i, act_i] + 1e-3 * torch.randn(n, generator=g).to(dev)).contiguous()
        values = (reward + 0.9 * v.to(torch.float64)).contiguous()
        ref = values.clone()
        ref[env_i, act_i] = reward[env_i, act_i] + 0.9 * exact.to(torch.float64)
        worst = torch.zeros(1, dtype=torch.float32, device=dev)
        _capi.check(_capi.lib().ebc_decision_apply(torch.cuda.current_stream(dev).cuda_stream, exact.data_ptr(), v.data_ptr(), env_i.data_ptr(),
                                                   act_i.data_ptr(), reward.data_ptr(), 0.9, A, n, values.data_ptr(), worst.data_ptr()))
        assert torch.equal(values, ref)
        assert float(worst) == float((exact - v[env_i, act_i]).abs().max())
