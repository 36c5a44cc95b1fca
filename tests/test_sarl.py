"""Robot decisions = look-ahead sweep + SARL value network.  Golden: the
reference's own SARL policy (its shipped weights) driving full episodes, 81 action values per
decision (tests/golden/sarl_*.npz, humans on the oracle-substituted rvo2).  Values are float32
network outputs on O(1) numbers: 2e-4 absolute; the chosen action must be the reference's
unless the two best values are closer than that."""
import json
import os

import numpy as np
import pytest
import torch

from ebcsim import _abi
from ebcsim.sarl import DeviceSarlPolicy, SarlValueNet
from helpers import GOLDEN, batch_from_init, load, params_of

RUNS = ["sarl_a5_baseline", "sarl_n10_ebcadrl"]
TOL = 2e-4


def _check_values(values, z, t):
    ref = z["values"][t]
    np.testing.assert_allclose(values, ref, atol=TOL, rtol=0, err_msg="decision %d" % t)
    best = int(np.argmax(values))
    chosen = int(np.where((z["action_space"] == z["action"][t]).all(1))[0][0])
    if best != chosen:
        top = np.sort(ref)[-2:]
        assert top[1] - top[0] < TOL, (t, best, chosen)


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
    steps = min(len(z["action"]), 40)
    for t in range(steps):
        la = env.lookahead(z["action_space"], human_policy=_abi.HUMAN_ORCA)
        if not np.isnan(z["values"][t]).any():
            vals = pol.values_from(torch.from_numpy(la["rows_rotated"]), torch.from_numpy(la["reward"]),
                                   None, params.time_step, v_pref)
            _check_values(vals[0].numpy(), z, t)
        out = env.step(robot_action=z["action"][t][None], human_policy=_abi.HUMAN_CACHED)
        assert int(out["info"][0]) == int(z["info"][t])


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
    for t in range(len(z["action"])):
        actions, values = pol.decide(env)
        torch.cuda.synchronize()
        v = values.cpu().numpy()
        assert (np.abs(v - v[0:1]) < 1e-5).all()
        if not np.isnan(z["values"][t]).any():
            _check_values(v[0], z, t)
        forced = torch.tensor(np.tile(z["action"][t], (E, 1)), dtype=torch.float64, device="cuda:0")
        env.step_device(outs, robot_action=forced, human_policy=_abi.HUMAN_CACHED)
        torch.cuda.synchronize()
        assert int(outs["info"][0]) == int(z["info"][t]), t
        np.testing.assert_allclose(float(outs["reward"][0]), z["reward"][t], atol=1e-9)
    assert int(z["info"][-1]) == _abi.INFO_REACH_GOAL


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
