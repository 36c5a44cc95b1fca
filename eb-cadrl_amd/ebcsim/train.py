"""Data-parallel value-network training on device-resident rollouts (BASELINE config 5).

Mirrors, batched: the replay memory (rl/utils/memory.py:4-28); the value targets of
Explorer.update_memory (rl/utils/explorer.py:151-200) in both modes — imitation learning: the
episode's discounted return from each step on; RL: terminal -> reward, else reward +
gamma^(dt * v_pref) * V_target(next state) —; Trainer.optimize_epoch / optimize_batch
(rl/utils/trainer.py:45-100: MSE regression, SGD momentum 0.9 or Adam); and the schedule of
rl/train.py:99-260 (imitation learning with the robot on ORCA, then epsilon-greedy RL with a
target network) as `run_training`.

Multi-GPU: every rank rolls out its own env slice (no collective on the sim path) and holds its
own replay shard; the only exchange is ONE all-reduce of the flattened gradient per optimizer
step (SARL has 0.39-1.5 MB of fp32 parameters: latency-bound on xGMI, so a single bucket, not
per-tensor calls).  torch.distributed backend "nccl" is RCCL on ROCm; tests use gloo."""
import numpy as np
import torch
import torch.distributed as dist

from . import _abi
from .sarl import SarlValueNet, _mlp, uniform_v_pref


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

    @staticmethod
    def _ref_name(k):
        return k.replace("_weight", ".weight").replace("_bias", ".bias").replace("_", ".", 1)

    def state_dict(self, *args, **kw):
        """Keys in the reference's layout (`mlp1.0.weight`, ...): what rl/train.py saves and
        `policy.get_model().load_state_dict(torch.load(path))` (rl/test.py:90) loads."""
        return {self._ref_name(k): v for k, v in super().state_dict(*args, **kw).items()}

    def load_state_dict(self, sd, strict=True, **kw):
        own = {self._ref_name(k): k for k, _ in self.named_parameters()}
        return super().load_state_dict({own.get(k, k): v for k, v in sd.items()}, strict=strict, **kw)

    def _stack(self, name):
        n = len(self._layout[name]) - 1
        return [(getattr(self, "%s_%d_weight" % (name, 2 * i)), getattr(self, "%s_%d_bias" % (name, 2 * i)))
                for i in range(n)]

    def reference_state_dict(self):
        return {self._ref_name(k): v.detach() for k, v in self.named_parameters()}

    def load_reference_state_dict(self, sd):
        with torch.no_grad():
            for k, v in self.named_parameters():
                v.copy_(sd[self._ref_name(k)])

    def as_value_net(self, native=False):
        """A SarlValueNet whose tensors ARE this module's parameters (it follows every optimizer step).  native: also
        build the matrix-core blocks for inference from the current weights — the caller re-packs them with
        net.refresh_native() after the weights have changed (run_training: once per round)."""
        net = SarlValueNet.__new__(SarlValueNet)
        net.mlp1, net.mlp2 = self._stack("mlp1"), self._stack("mlp2")
        net.attention, net.mlp3 = self._stack("attention"), self._stack("mlp3")
        net.with_global_state, net.self_state_dim = self.with_global_state, self.self_state_dim
        net.input_dim = self._layout["mlp1"][0]
        net.device = next(self.parameters()).device
        net.dtype = torch.float32
        net._native = ()  # the weights change under training: packed copies only on request (native=True)
        if native and net.device.type == "cuda":
            net.enable_native()
        return net

    def forward(self, rows, n_valid=None):
        net = self.as_value_net()
        return net._forward(rows, n_valid)


class DeviceReplay(object):
    """Ring buffer of (rotated joint state [R, T], value, rows that exist) triples in device memory.
    `n_valid` is what the reference's variable-length state tensors carry implicitly (an env with fewer
    humans + static rows than R has padding rows the network must not see, multi_human_rl.py:128-149)."""

    def __init__(self, capacity, R, T, device):
        self.states = torch.zeros((capacity, R, T), dtype=torch.float32, device=device)
        self.values = torch.zeros(capacity, dtype=torch.float32, device=device)
        self.n_valid = torch.full((capacity,), R, dtype=torch.int64, device=device)
        self.capacity, self.position, self.size, self.R = int(capacity), 0, 0, int(R)
        self.ragged = False  # any pair with fewer than R rows

    def push(self, states, values, n_valid=None):
        n = states.shape[0]
        if n == 0:
            return
        total = n
        if n > self.capacity:  # only the newest `capacity` items survive a ring: no duplicate slots in one write
            states, values = states[-self.capacity:], values[-self.capacity:]
            n_valid = None if n_valid is None else n_valid[-self.capacity:]
            self.position = (self.position + n - self.capacity) % self.capacity
            n = self.capacity
        idx = (torch.arange(n, device=states.device) + self.position) % self.capacity
        self.states[idx] = states
        self.values[idx] = values.to(torch.float32)
        if n_valid is None:
            self.n_valid[idx] = self.R
        else:  # callers pass row counts only for handles that may run ragged scenes (no device sync here)
            self.n_valid[idx] = n_valid.to(torch.int64)
            self.ragged = True
        self.position = (self.position + n) % self.capacity
        self.size = min(self.capacity, self.size + total)

    def rows_of(self, idx):
        """n_valid for model(states[idx], n_valid) — None while every stored pair has all R rows."""
        return self.n_valid[idx] if self.ragged else None

    def sample(self, batch_size, generator=None, with_rows=False):
        idx = torch.randint(0, self.size, (batch_size,), device=self.states.device, generator=generator)
        if with_rows:
            return self.states[idx], self.values[idx], self.rows_of(idx)
        return self.states[idx], self.values[idx]

    def __len__(self):
        return self.size


def value_targets(reward, done, next_states, target_net, gamma_bar, n_valid=None):
    """explorer.py:171-184: terminal -> reward; else reward + gamma_bar * V_target(s').  n_valid: rows of
    each next state that exist (None = all R)."""
    with torch.no_grad():
        nxt = target_net.forward(next_states, n_valid)
    return torch.where(done.bool(), reward, reward + gamma_bar * nxt.to(reward.dtype))


def stored_episode(info):
    """explorer.py:82-92: an episode's experience is stored only when it ended in ReachGoal or in a
    collision — not on a timeout.  info: EBC_INFO_* codes of terminal steps."""
    return ((info == _abi.INFO_REACH_GOAL) | (info == _abi.INFO_COLLISION_OBSTACLE) | (info == _abi.INFO_COLLISION_ADULT)
            | (info == _abi.INFO_COLLISION_BICYCLE) | (info == _abi.INFO_COLLISION_CHILD))


def il_value_targets(rewards, done, gamma_bar, info=None):
    """Imitation-learning targets (explorer.py:159-170): value_i = sum over the REST OF ITS EPISODE
    of gamma_bar^(t - i) * reward_t, gamma_bar = gamma^(dt * v_pref).  rewards, done: [T, E] in time
    order, episodes of an env following one another (auto-reset).  Returns (values [T, E], keep
    [T, E]): `keep` marks the steps whose episode ended inside the window — the reference only ever
    stores whole episodes (explorer.py:33-92) — and, when the steps' info codes are given, ended in
    ReachGoal or a collision (`stored_episode`)."""
    T = rewards.shape[0]
    if (rewards.is_cuda and info is not None and rewards.dtype == torch.float64 and rewards.dim() == 2
            and done.dtype == torch.uint8 and info.dtype == torch.uint8):
        # one thread per env walks its window backwards (libebcsim ebc_il_targets): the T-iteration torch loop
        # below was most of an imitation-learning rollout's wall time
        from . import _capi
        rewards, done, info = rewards.contiguous(), done.contiguous(), info.contiguous()
        values = torch.empty_like(rewards)
        keep8 = torch.empty_like(done)
        _capi.check(_capi.lib().ebc_il_targets(torch.cuda.current_stream(rewards.device).cuda_stream, rewards.data_ptr(),
                                               done.data_ptr(), info.data_ptr(), int(T), int(rewards.shape[1]), float(gamma_bar),
                                               values.data_ptr(), keep8.data_ptr()))
        return values, keep8.bool()
    values = torch.zeros_like(rewards)
    keep = torch.zeros_like(done, dtype=torch.bool)
    run = torch.zeros_like(rewards[0])
    closed = torch.zeros_like(done[0], dtype=torch.bool)
    for t in range(T - 1, -1, -1):
        d = done[t].bool()
        run = torch.where(d, rewards[t], rewards[t] + gamma_bar * run)
        closed = torch.where(d, torch.ones_like(d) if info is None else stored_episode(info[t]), closed)
        values[t] = run
        keep[t] = closed
    return values, keep


class EpisodeStore(object):
    """Where an env's (state, value) pairs wait for their episode's end: the reference pushes an episode
    to the replay memory when it is over, and only if it ended in ReachGoal or a collision
    (explorer.py:82-92).  [T_max][E] slots in device memory, one write cursor per env."""

    def __init__(self, E, T_max, R, T, device):
        self.states = torch.zeros((T_max, E, R, T), dtype=torch.float32, device=device)
        self.values = torch.zeros((T_max, E), dtype=torch.float32, device=device)
        self.n_valid = torch.full((T_max, E), R, dtype=torch.int64, device=device)
        self.length = torch.zeros(E, dtype=torch.int64, device=device)
        self.T_max, self.E = int(T_max), int(E)
        self._env = torch.arange(E, device=device)
        self._t = torch.arange(T_max, device=device)[:, None]

    def add(self, states, values, n_valid=None):
        t = self.length.clamp(max=self.T_max - 1)  # an episode cannot outlast time_limit / time_step steps
        self.states[t, self._env] = states
        self.values[t, self._env] = values.to(torch.float32)
        if n_valid is not None:
            self.n_valid[t, self._env] = n_valid.to(torch.int64)
        self.length = (self.length + 1).clamp(max=self.T_max)

    def end(self, done, info, memory):
        """After a step: episodes that ended go to `memory` (or are dropped); returns how many pairs went."""
        done = done.bool()
        go = done & stored_episode(info)
        n = 0
        if bool(go.any()):
            mask = (self._t < self.length[None, :]) & go[None, :]
            n = int(mask.sum())
            memory.push(self.states[mask], self.values[mask], self.n_valid[mask])
        self.length = torch.where(done, torch.zeros_like(self.length), self.length)
        return n


def _multi_rank():
    """A process group with more than one rank — or with one rank and EBCSIM_FORCE_COLLECTIVES=1: every collective of
    the schedule then runs through the backend (RCCL on a one-GPU box: tests/test_train_gloo.py) instead of being
    skipped as the identity it is."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    import os
    return dist.get_world_size() > 1 or os.environ.get("EBCSIM_FORCE_COLLECTIVES") == "1"


def all_ranks(flag, device=None):
    """True on every rank iff `flag` holds on EVERY rank (one MIN all-reduce).  Branches that contain
    collectives (the gradient all-reduce of an optimizer step) must be taken by all ranks or by none."""
    if not _multi_rank():
        return bool(flag)
    t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(int(t.item()))


def broadcast_parameters_(model, src=0):
    """Every replica starts from rank `src`'s weights (one flat broadcast), whatever seeds the ranks ran
    their constructors with; the reference's workers get theirs from the parent process
    (rl/utils/parallel_explorer.py:59-70)."""
    if not _multi_rank():
        return
    params = list(model.parameters())
    flat = torch.cat([p.detach().reshape(-1) for p in params])
    dist.broadcast(flat, src=src)
    off = 0
    with torch.no_grad():
        for p in params:
            p.copy_(flat[off:off + p.numel()].view_as(p))
            off += p.numel()


def allreduce_flat_(params):
    """Average the gradients of `params` over ranks with ONE all-reduce of a flat buffer."""
    if not _multi_rank():
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

    def set_learning_rate(self, learning_rate):
        """Trainer.set_optimizer (trainer.py:24-43): a new optimizer at the stage's learning rate."""
        params = list(self.model.parameters())
        if isinstance(self.optimizer, torch.optim.Adam):
            self.optimizer = torch.optim.Adam(params, lr=learning_rate)
        else:
            self.optimizer = torch.optim.SGD(params, lr=learning_rate, momentum=0.9)

    def optimize_epoch(self, num_epochs, generator=None):
        """Trainer.optimize_epoch (trainer.py:45-72, imitation learning): `num_epochs` shuffled passes
        over the memory in batches.  Every rank walks its own shard; the gradient of each batch is
        averaged over ranks, so all ranks take the same number of batches (the smallest shard's)."""
        params = list(self.model.parameters())
        n = len(self.memory)
        batches = (n + self.batch_size - 1) // self.batch_size
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            t = torch.tensor([batches], dtype=torch.int64, device=self.memory.states.device)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            batches = int(t.item())
        average = 0.0
        for _ in range(num_epochs):
            perm = torch.randperm(n, device=self.memory.states.device, generator=generator)
            epoch_loss = 0.0
            for b in range(batches):
                idx = perm[b * self.batch_size:(b + 1) * self.batch_size]
                self.optimizer.zero_grad()
                loss = self.criterion(self.model(self.memory.states[idx], self.memory.rows_of(idx)),
                                      self.memory.values[idx])
                loss.backward()
                allreduce_flat_(params)
                self.optimizer.step()
                epoch_loss += float(loss.detach())
            average = epoch_loss / max(n, 1)  # trainer.py:69: divided by len(memory), as the reference does
        return average

    def optimize_batch(self, num_batches, generator=None):
        losses = 0.0
        params = list(self.model.parameters())
        for _ in range(num_batches):
            if getattr(self.memory, "ragged", False):
                inputs, values, rows = self.memory.sample(self.batch_size, generator, with_rows=True)
            else:
                (inputs, values), rows = self.memory.sample(self.batch_size, generator), None
            self.optimizer.zero_grad()
            loss = self.criterion(self.model(inputs, rows), values)
            loss.backward()
            allreduce_flat_(params)
            self.optimizer.step()
            losses += float(loss.detach())
        return losses / num_batches


def collect(env, policy, target_net, memory, steps, gamma, epsilon=0.0, generator=None,
            human_policy=_abi.HUMAN_ORCA, store=None):
    """Roll the rank's env slice `steps` decisions forward with an epsilon-greedy SARL policy
    (multi_human_rl.py:31-33, :84-85) and push (state, value target) pairs.  env: BatchedEnv on
    the policy's device with auto-reset; returns the mean reward.  With an EpisodeStore the pairs
    reach the memory as the reference's do: per finished episode, successes and collisions only;
    without one every step's pair goes in at once."""
    dev = policy.net.device
    E, R, T = env.E, env.R, env.T
    cur = torch.zeros((E, R, T), dtype=torch.float32, device=dev)
    outs = env.alloc_step_outputs(("reward", "done", "info", "obs_rotated"))
    env.observe_device(cur)
    v_pref = uniform_v_pref(env)
    gamma_bar = gamma ** (env.params.time_step * v_pref)
    A = len(policy.actions_np)
    total = 0.0
    ragged = bool(getattr(env, "ragged", False))
    for _ in range(steps):
        actions, _ = policy.decide(env, human_policy=human_policy)
        nv_cur = policy.n_valid.clone() if ragged else None  # rows of `cur` (the state the decision saw)
        if epsilon > 0:
            explore = torch.rand(E, device=dev, generator=generator) < epsilon
            rnd = torch.randint(0, A, (E,), device=dev, generator=generator)
            actions = torch.where(explore[:, None], policy._acts[rnd], actions)
        env.step_device(outs, robot_action=actions.contiguous(), human_policy=_abi.HUMAN_CACHED,
                        flags=_abi.FLAG_AUTO_RESET)
        # the returned observation belongs to the scene that was stepped (a terminal env's target is its
        # reward, its restart scene shows up in `cur` below): its row count is the decision's
        targets = value_targets(outs["reward"], outs["done"], outs["obs_rotated"], target_net, gamma_bar, nv_cur)
        if store is None:
            memory.push(cur.clone(), targets, nv_cur)
        else:
            store.add(cur, targets, nv_cur)
            store.end(outs["done"], outs["info"], memory)
        total += float(outs["reward"].mean())
        # the next decision's state: the returned observation, or the reset scene after a terminal step
        env.observe_device(cur)
    env.synchronize()  # surfaces a mailbox fault of any step above (EBC_ERR_DEVICE) instead of training on it
    return total / steps


def collect_il(env, memory, steps, gamma, safety_space=0.0, human_policy=_abi.HUMAN_ORCA, persistent_sim=True):
    """The imitation-learning stage's rollouts (rl/train.py:124-133, explorer.py:33-92 with
    imitation_learning=True): the robot of every env of the rank's slice on ORCA (ebc_robot_orca ->
    ebc_step, auto-reset) for `steps` steps; the states of every episode that ended inside the window in
    ReachGoal or a collision (explorer.py:82-92) go to `memory` with their discounted returns.  Returns
    (steps stored, episodes ended).
    Demonstrator semantics (persistent_sim=True, the reference's): every env is ONE il_policy object playing its
    episodes one after the other, as rl/train.py:130-133 builds one for the whole stage — its rvo2 simulator is
    built at the env's first step of this call and again whenever the row count of its scene changes, and in between
    keeps the radii / maxSpeed it was built with (simulator/policy/orca.py:96-133), across restarts too: an env's
    episode sequence equals the reference's serial sequence on the same scenes (tests: il_persistent_* goldens).
    persistent_sim=False: the demonstrator sees every state's own radii (a fresh policy object per step)."""
    E, R, T = env.E, env.R, env.T
    env.robot_orca_sim(bool(persistent_sim))
    v_pref = uniform_v_pref(env)
    gamma_bar = gamma ** (env.params.time_step * v_pref)
    ragged = bool(getattr(env, "ragged", False))
    # the whole window in ONE call through the C ABI (ebc_step_k): per step the library enqueues the state the
    # policy sees (policy.last_state transformed, explorer.py:43, :162), the robot's ORCA action and the step,
    # every output written at its step index; no host work between the steps
    keys = ("state_rotated", "reward", "done", "info") + (("n_rows",) if ragged else ())
    outs = env.alloc_step_k_outputs(steps, keys)
    env.step_k_device(outs, steps, human_policy=human_policy, robot_policy=_abi.ROBOT_ORCA,
                      flags=_abi.FLAG_AUTO_RESET, robot_safety_space=safety_space)
    states, rewards, dones, infos = outs["state_rotated"], outs["reward"], outs["done"], outs["info"]
    rows = outs["n_rows"] if ragged else None
    values, keep = il_value_targets(rewards, dones, gamma_bar, infos)
    keep = keep.reshape(-1)
    memory.push(states.reshape(steps * E, R, T)[keep], values.reshape(-1)[keep],
                rows.reshape(-1)[keep] if ragged else None)
    env.synchronize()  # a natural sync point: also surfaces a mailbox fault of any step above
    return int(keep.sum()), int(dones.sum())


def run_training(env, model, actions, gamma, il_steps=0, il_epochs=0, il_learning_rate=0.01, il_safety_space=0.15,
                 rl_learning_rate=0.001, train_iterations=0, steps_per_iteration=1, train_batches=100,
                 batch_size=100, capacity=100000, epsilon_start=0.5, epsilon_end=0.1, epsilon_decay=4000,
                 target_update_interval=50, optimizer_algorithm="sgd", generator=None, log=None,
                 output_dir=None, checkpoint_interval=0, evaluation_interval=0, val_env=None, val_scenes=None,
                 rank=0, scene_gen=None, scene_seed0=2000, scene_pool_factor=4, scene_pool_steps=None):
    """The schedule of rl/train.py:99-260 on one rank's env slice (every rank calls it; gradients are
    averaged over ranks inside the trainer): imitation learning with the robot on ORCA
    (`il_steps` steps of every env, then `il_epochs` passes over the memory), then `train_iterations`
    rounds of [epsilon-greedy rollout of `steps_per_iteration` decisions per env -> `train_batches`
    optimizer steps], the target network refreshed every `target_update_interval` rounds, epsilon
    decayed linearly over `epsilon_decay` rounds (train.py:211-218).  env: BatchedEnv with auto-reset
    semantics, on the model's device.  Returns a dict of what happened.
    output_dir: the reference's weight files, in its state_dict layout (they load into its ValueNetwork):
    `il_model.pth` after imitation learning — found there, it is loaded and the stage skipped
    (train.py:113-116) — and `rl_model_<round>.pth` every `checkpoint_interval` rounds and at the end
    (train.py:146-150, :262-270); written by `rank` 0 only.  evaluation_interval: every so many rounds the
    greedy policy runs `val_scenes` (a SceneBatch) on `val_env` once through (train.py:222-236, `evaluate`);
    the metrics go to hist["val"].
    scene_gen: an EbcSceneGen (scene.gen_struct(cfg, "train")): the episodes then run on scenes generated on the
    device from consecutive seeds starting at `scene_seed0` (the reference: seed = 2000 + the number of train
    episodes so far, simulator/env.py:153-169; here every env draws from its own run of that sequence — give each
    rank its own `scene_seed0`): the envs are reset from the first E seeds, the auto-reset pool holds the next
    `scene_pool_factor` x E and is regenerated from fresh seeds every `scene_pool_steps` env steps (default: before
    an env can have walked through its share), with nothing but the seed crossing PCIe."""
    import os
    from .sarl import DeviceSarlPolicy
    dev = next(model.parameters()).device
    memory = DeviceReplay(capacity, env.R, env.T, dev)
    trainer = DataParallelTrainer(model, memory, batch_size, optimizer_algorithm, il_learning_rate)
    hist = {"il_stored": 0, "il_episodes": 0, "il_loss": None, "rl_loss": [], "mean_reward": [], "val": [],
            "il_loaded": False}
    il_file = os.path.join(output_dir, "il_model.pth") if output_dir else None

    def save(path):
        if rank == 0 and output_dir:
            os.makedirs(output_dir, exist_ok=True)
            torch.save({k: v.cpu() for k, v in model.reference_state_dict().items()}, path)
    # Rank 0 decides whether the imitation-learning stage is skipped (it is the rank that writes the file),
    # and its weights are every replica's starting point: the stages below contain collectives, so all
    # ranks must walk the same schedule.
    found = bool(il_file and os.path.exists(il_file)) if rank == 0 else False
    if _multi_rank():
        t = torch.tensor([1 if found else 0], dtype=torch.int32, device=dev if dev.type == "cuda" and dist.get_backend() == "nccl" else None)
        dist.broadcast(t, src=0)
        found = bool(int(t.item()))
    if found:
        if rank == 0 or os.path.exists(il_file):
            model.load_reference_state_dict(torch.load(il_file, map_location="cpu"))
        hist["il_loaded"] = True
        il_steps = 0
        if log:
            log("imitation learning: weights loaded from %s" % il_file)
    broadcast_parameters_(model, 0)
    red_dev = dev if (dev.type == "cuda" and _multi_rank() and dist.get_backend() == "nccl") else None
    next_seed, pool_age = [int(scene_seed0)], [0]
    if scene_pool_steps is None:  # an episode is at least a few steps long: a pool of f scenes per env lasts longer than 4 f steps
        scene_pool_steps = 4 * scene_pool_factor

    def fresh_pool(steps_done=0):
        if scene_gen is None:
            return
        pool_age[0] += steps_done
        if steps_done and pool_age[0] < scene_pool_steps:
            return
        P = scene_pool_factor * env.E
        env.generate_pool(scene_gen, next_seed[0], P)
        next_seed[0] += P
        pool_age[0] = 0
        hist["scene_pools"] = hist.get("scene_pools", 0) + 1
    if scene_gen is not None:
        env.generate_reset(scene_gen, next_seed[0])
        next_seed[0] += env.E
        fresh_pool()
    if il_steps > 0:
        if scene_gen is not None and il_steps > scene_pool_steps:  # a pool that lasts the whole stage
            P = min(-(-il_steps // 4), 16) * env.E  # capped: a long stage walks its scenes again
            env.generate_pool(scene_gen, next_seed[0], P)
            next_seed[0] += P
        hist["il_stored"], hist["il_episodes"] = collect_il(env, memory, il_steps, gamma, il_safety_space)
        fresh_pool(scene_pool_steps)
        if il_epochs > 0 and all_ranks(len(memory) > 0, red_dev):  # every rank's shard has data, or nobody trains
            hist["il_loss"] = trainer.optimize_epoch(il_epochs, generator)
        if il_file:
            save(il_file)
        if log:
            log("imitation learning: %d states of %d episodes, loss %s" % (hist["il_stored"], hist["il_episodes"], hist["il_loss"]))
    import copy
    target = copy.deepcopy(model)                      # explorer.update_target_model (train.py:195)
    trainer.set_learning_rate(rl_learning_rate)
    # the rollouts' decisions and the target values run on the matrix-core blocks: packed from the module's weights now,
    # re-packed on the device (ebc_mlp2_update) after every round's optimizer steps / target refresh
    policy = DeviceSarlPolicy(model.as_value_net(native=True), actions, gamma)
    target_net = target.as_value_net(native=True)
    val_policy = DeviceSarlPolicy(policy.net, actions, gamma)  # same network, its own look-ahead buffers (val_env's size)
    t_max = int(round(env.params.time_limit / env.params.time_step)) + 2
    store = EpisodeStore(env.E, t_max, env.R, env.T, dev)
    for it in range(train_iterations):
        if evaluation_interval and val_env is not None and val_scenes is not None and it % evaluation_interval == 0:
            val_env.reset(val_scenes)
            hist["val"].append((it, evaluate(val_env, lambda e: val_policy.decide(e)[0], gamma, human_policy=_abi.HUMAN_CACHED)))
            if log:
                m = hist["val"][-1][1]
                log("VAL   in round %d has success rate: %.2f, nav time: %.2f, total reward: %.4f" % (
                    it, m["success_rate"], m["avg_nav_time"], m["total_reward:"]))
        eps = epsilon_start + (epsilon_end - epsilon_start) / epsilon_decay * it if it < epsilon_decay else epsilon_end
        mean_r = collect(env, policy, target_net, memory, steps_per_iteration, gamma, epsilon=eps,
                         generator=generator, store=store)
        fresh_pool(steps_per_iteration)
        # ranks fill their shards at different times (an episode reaches the memory when it ends): the
        # optimizer steps all-reduce gradients, so a round trains on every rank or on none
        loss = trainer.optimize_batch(train_batches, generator) if all_ranks(len(memory) > 0, red_dev) else float("nan")
        policy.net.refresh_native()
        if (it + 1) % target_update_interval == 0:
            target.load_state_dict(model.state_dict())
            target_net.refresh_native()
        hist["rl_loss"].append(loss)
        hist["mean_reward"].append(mean_r)
        if checkpoint_interval and (it + 1) % checkpoint_interval == 0:
            save(os.path.join(output_dir or ".", "rl_model_%d.pth" % (it + 1)))
        if log:
            log("iteration %d: epsilon %.3f mean reward %.4f loss %.3e" % (it, eps, mean_r, loss))
    if train_iterations:
        save(os.path.join(output_dir or ".", "rl_model_%d.pth" % train_iterations))
    # how the rollouts decided: matrix-core forwards and on-device re-packs of the blocks (0 / 0 on a CPU device)
    hist["native_forwards"] = int(getattr(policy.net, "native_forwards", 0))
    hist["native_refreshes"] = int(getattr(policy.net, "native_refreshes", 0))
    return hist


def evaluate(env, decide, gamma, max_steps=None, human_policy=_abi.HUMAN_ORCA):
    """Explorer.run_k_episodes on the `val` / `test` cases (explorer.py:33-131, :202-330), batched: one
    episode per env of a freshly reset BatchedEnv (no auto-reset; an env that has ended is ignored from
    then on).  decide(env) -> robot actions, float64 CUDA tensor [E, 2] (e.g. `lambda e:
    policy.decide(e)[0]` with human_policy=EBC_HUMAN_CACHED: the decision's sweep has already worked out
    the humans' velocities).  Returns the reference's compile_metrics() dictionary (same keys; the case
    lists hold env indices) plus "num_episodes"."""
    dev = torch.device("cuda", env.device)
    E = env.E
    dt = float(env.params.time_step)
    limit = float(env.params.time_limit)
    steps = int(max_steps or round(limit / dt) + 2)
    outs = env.alloc_step_outputs(("reward", "done", "info", "dmin"))
    v_pref = uniform_v_pref(env)
    gamma_bar = gamma ** (dt * v_pref)
    dd = torch.tensor(list(env.params.discomfort_dist), dtype=torch.float64, device=dev)  # adult, bicycle, child
    alive = torch.ones(E, dtype=torch.bool, device=dev)
    final = torch.full((E,), -1, dtype=torch.int64, device=dev)
    end_time = torch.zeros(E, dtype=torch.float64, device=dev)
    cumulative = torch.zeros(E, dtype=torch.float64, device=dev)
    too_close = torch.zeros((), dtype=torch.int64, device=dev)
    min_sum = torch.zeros((), dtype=torch.float64, device=dev)
    for t in range(steps):
        actions = decide(env)
        env.step_device(outs, robot_action=actions.contiguous(), human_policy=human_policy)
        info = outs["info"].to(torch.int64)
        cumulative += torch.where(alive, (gamma_bar ** t) * outs["reward"], torch.zeros_like(cumulative))
        danger = alive & (info == _abi.INFO_DANGER)
        dm = outs["dmin"]  # Danger.min_dist: the first type under its discomfort distance, child > bicycle > adult
        md = torch.where(dm[:, 2] < dd[2], dm[:, 2], torch.where(dm[:, 1] < dd[1], dm[:, 1], dm[:, 0]))
        too_close += danger.sum()
        min_sum += torch.where(danger, md, torch.zeros_like(md)).sum()
        ended = alive & outs["done"].bool()
        final = torch.where(ended, info, final)
        end_time = torch.where(ended, torch.where(info == _abi.INFO_TIMEOUT, torch.full_like(end_time, limit),
                                                  torch.full_like(end_time, (t + 1) * dt)), end_time)
        alive = alive & ~ended
        if t % 8 == 7 and not bool(alive.any()):
            break
    env.synchronize()  # before the metrics: a broken step must not be counted
    final, end_time = final.cpu().numpy(), end_time.cpu().numpy()
    if (final < 0).any():
        raise ValueError("Invalid end signal from environment")  # explorer.py:80: every episode must end
    n = float(E)
    count = lambda code: int((final == code).sum())  # noqa: E731
    cases = lambda code: [str(i) for i in np.nonzero(final == code)[0]]  # noqa: E731
    ok = final == _abi.INFO_REACH_GOAL
    num_step = end_time.sum() / dt
    tc = int(too_close)
    return {
        "success_rate": count(_abi.INFO_REACH_GOAL) / n, "collision_rate": 0.0,
        "collision_rate_adult": count(_abi.INFO_COLLISION_ADULT) / n,
        "collision_rate_bicycle": count(_abi.INFO_COLLISION_BICYCLE) / n,
        "collision_rate_child": count(_abi.INFO_COLLISION_CHILD) / n,
        "collision_rate_obstacle": count(_abi.INFO_COLLISION_OBSTACLE) / n,
        "success": count(_abi.INFO_REACH_GOAL), "collision": 0, "timeout": count(_abi.INFO_TIMEOUT),
        "avg_nav_time": float(end_time[ok].mean()) if ok.any() else limit,
        "total_reward:": float(cumulative.mean()),
        "Frequency of being in danger": tc / num_step if num_step else None,
        "average min separate distance in danger": float(min_sum) / tc if tc else 0,
        "Collision cases:": [], "Collision Adult cases:": cases(_abi.INFO_COLLISION_ADULT),
        "Collision Bicycle cases:": cases(_abi.INFO_COLLISION_BICYCLE),
        "Collision Child cases:": cases(_abi.INFO_COLLISION_CHILD),
        "Collision Obstacle cases:": cases(_abi.INFO_COLLISION_OBSTACLE),
        "Timeout cases": cases(_abi.INFO_TIMEOUT), "num_episodes": E,
    }
