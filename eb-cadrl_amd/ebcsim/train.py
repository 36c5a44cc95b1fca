"""Data-parallel value-network training on device-resident rollouts (BASELINE config 5).

Mirrors, batched: the replay memory (rl/utils/memory.py:4-28), the value targets of
Explorer.update_memory in RL mode (rl/utils/explorer.py:151-200: terminal -> reward, else
reward + gamma^(dt * v_pref) * V_target(next state)), and Trainer.optimize_batch
(rl/utils/trainer.py:74-100: MSE regression, SGD momentum 0.9 or Adam).

Multi-GPU: every rank rolls out its own env slice (no collective on the sim path) and holds its
own replay shard; the only exchange is ONE all-reduce of the flattened gradient per optimizer
step (SARL has 0.39-1.5 MB of fp32 parameters: latency-bound on xGMI, so a single bucket, not
per-tensor calls).  torch.distributed backend "nccl" is RCCL on ROCm; tests use gloo."""
import numpy as np
import torch
import torch.distributed as dist

from . import _abi
from .sarl import SarlValueNet, _mlp


class SarlModule(torch.nn.Module):
    """Trainable twin of SarlValueNet: same parameter names as the reference's ValueNetwork
    (rl/policy/sarl.py:9-36), so state_dicts move both ways."""

    def __init__(self, input_dim, mlp1_dims, mlp2_dims, mlp3_dims, attention_dims,
                 with_global_state=True, self_state_dim=6):
        super().__init__()

        def stack(name, dims):
            idx = 0
            for a, b in zip(dims[:-1], dims[1:]):
                lin = torch.nn.Linear(a, b)
                self.register_parameter("%s_%d_weight" % (name, idx), lin.weight)
                self.register_parameter("%s_%d_bias" % (name, idx), lin.bias)
                idx += 2
        self.with_global_state = with_global_state
        self.self_state_dim = self_state_dim
        att_in = mlp1_dims[-1] * (2 if with_global_state else 1)
        self._layout = {"mlp1": [input_dim] + mlp1_dims, "mlp2": [mlp1_dims[-1]] + mlp2_dims,
                        "attention": [att_in] + attention_dims,
                        "mlp3": [mlp2_dims[-1] + self_state_dim] + mlp3_dims}
        for name, dims in self._layout.items():
            stack(name, dims)

    def _stack(self, name):
        n = len(self._layout[name]) - 1
        return [(getattr(self, "%s_%d_weight" % (name, 2 * i)), getattr(self, "%s_%d_bias" % (name, 2 * i)))
                for i in range(n)]

    def reference_state_dict(self):
        return {k.replace("_weight", ".weight").replace("_bias", ".bias").replace("_", ".", 1): v.detach()
                for k, v in self.named_parameters()}

    def load_reference_state_dict(self, sd):
        with torch.no_grad():
            for k, v in self.named_parameters():
                v.copy_(sd[k.replace("_weight", ".weight").replace("_bias", ".bias").replace("_", ".", 1)])

    def as_value_net(self):
        net = SarlValueNet.__new__(SarlValueNet)
        net.mlp1, net.mlp2 = self._stack("mlp1"), self._stack("mlp2")
        net.attention, net.mlp3 = self._stack("attention"), self._stack("mlp3")
        net.with_global_state, net.self_state_dim = self.with_global_state, self.self_state_dim
        net.input_dim = self._layout["mlp1"][0]
        net.device = next(self.parameters()).device
        net._native = ()  # the weights change under training: no packed copies of them
        return net

    def forward(self, rows, n_valid=None):
        net = self.as_value_net()
        return net._forward(rows, n_valid)


class DeviceReplay(object):
    """Ring buffer of (rotated joint state [R, T], value) pairs in device memory."""

    def __init__(self, capacity, R, T, device):
        self.states = torch.zeros((capacity, R, T), dtype=torch.float32, device=device)
        self.values = torch.zeros(capacity, dtype=torch.float32, device=device)
        self.capacity, self.position, self.size = int(capacity), 0, 0

    def push(self, states, values):
        n = states.shape[0]
        idx = (torch.arange(n, device=states.device) + self.position) % self.capacity
        self.states[idx] = states
        self.values[idx] = values.to(torch.float32)
        self.position = (self.position + n) % self.capacity
        self.size = min(self.capacity, self.size + n)

    def sample(self, batch_size, generator=None):
        idx = torch.randint(0, self.size, (batch_size,), device=self.states.device, generator=generator)
        return self.states[idx], self.values[idx]

    def __len__(self):
        return self.size


def value_targets(reward, done, next_states, target_net, gamma_bar):
    """explorer.py:171-184: terminal -> reward; else reward + gamma_bar * V_target(s')."""
    with torch.no_grad():
        nxt = target_net.forward(next_states)
    return torch.where(done.bool(), reward, reward + gamma_bar * nxt.to(reward.dtype))


def allreduce_flat_(params):
    """Average the gradients of `params` over ranks with ONE all-reduce of a flat buffer."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    grads = [p.grad for p in params if p.grad is not None]
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat /= dist.get_world_size()
    off = 0
    for g in grads:
        g.copy_(flat[off:off + g.numel()].view_as(g))
        off += g.numel()


class DataParallelTrainer(object):
    """Trainer.optimize_batch with the gradient averaged over ranks."""

    def __init__(self, model, memory, batch_size, optimizer_algorithm="sgd", learning_rate=0.001):
        self.model, self.memory, self.batch_size = model, memory, int(batch_size)
        params = list(model.parameters())
        if optimizer_algorithm == "adam":
            self.optimizer = torch.optim.Adam(params, lr=learning_rate)
        else:
            self.optimizer = torch.optim.SGD(params, lr=learning_rate, momentum=0.9)
        self.criterion = torch.nn.MSELoss()

    def optimize_batch(self, num_batches, generator=None):
        losses = 0.0
        params = list(self.model.parameters())
        for _ in range(num_batches):
            inputs, values = self.memory.sample(self.batch_size, generator)
            self.optimizer.zero_grad()
            loss = self.criterion(self.model(inputs), values)
            loss.backward()
            allreduce_flat_(params)
            self.optimizer.step()
            losses += float(loss.detach())
        return losses / num_batches


def collect(env, policy, target_net, memory, steps, gamma, epsilon=0.0, generator=None,
            human_policy=_abi.HUMAN_ORCA):
    """Roll the rank's env slice `steps` decisions forward with an epsilon-greedy SARL policy
    (multi_human_rl.py:31-33, :84-85) and push (state, value target) pairs.  env: BatchedEnv on
    the policy's device with auto-reset; returns the mean reward."""
    dev = policy.net.device
    E, R, T = env.E, env.R, env.T
    cur = torch.zeros((E, R, T), dtype=torch.float32, device=dev)
    outs = env.alloc_step_outputs(("reward", "done", "info", "obs_rotated"))
    env.observe_device(cur)
    v_pref = float(env.get_state()["robot"][0, 7])
    gamma_bar = gamma ** (env.params.time_step * v_pref)
    A = len(policy.actions_np)
    total = 0.0
    for _ in range(steps):
        actions, _ = policy.decide(env, human_policy=human_policy)
        if epsilon > 0:
            explore = torch.rand(E, device=dev, generator=generator) < epsilon
            rnd = torch.randint(0, A, (E,), device=dev, generator=generator)
            actions = torch.where(explore[:, None], policy._acts[rnd], actions)
        env.step_device(outs, robot_action=actions.contiguous(), human_policy=_abi.HUMAN_CACHED,
                        flags=_abi.FLAG_AUTO_RESET)
        targets = value_targets(outs["reward"], outs["done"], outs["obs_rotated"], target_net, gamma_bar)
        memory.push(cur.clone(), targets)
        total += float(outs["reward"].mean())
        # the next decision's state: the returned observation, or the reset scene after a terminal step
        env.observe_device(cur)
    return total / steps
