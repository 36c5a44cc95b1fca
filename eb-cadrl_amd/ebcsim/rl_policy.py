"""The value-based robot policy for the single-env facade: rl/policy/sarl.py:85-135 (SARL) with the
decision of rl/policy/multi_human_rl.py:12-87, served by the accelerated path.

The reference's predict() asks the env 81 times (`env.onestep_lookahead(action)`), rotates each answer
and runs the network 81 times.  Here one decision is ONE look-ahead sweep (`env.lookahead_all`:
ebc_lookahead leaves the rotated rows of all candidate actions) and ONE batched network forward
(ebcsim.sarl.SarlValueNet: the MFMA blocks on a HIP device, torch on the CPU).  Same attribute surface as
the reference's policy object (configure / set_phase / set_device / set_epsilon / get_model /
action_values / get_attention_weights / last_state), the same numpy draws for epsilon-greedy, the same
first-maximum tie rule, and its weight files load unchanged (`get_model().load_state_dict(torch.load(p))`)."""
import numpy as np
import torch

from .action import ActionRot, ActionXY
from .actions import build_action_space
from .policy import Policy


def _dims(config, key):
    return [int(x) for x in config.get("sarl", key).split(", ")]


class SARL(Policy):
    def __init__(self):
        Policy.__init__(self)
        self.name = "SARL"
        self.trainable = True
        self.multiagent_training = None
        self.epsilon = self.gamma = None
        self.sampling = self.speed_samples = self.rotation_samples = self.query_env = None
        self.action_space = self.speeds = self.rotations = None
        self.action_values = None
        self.with_om = None
        self.with_agent_type = False
        self.cell_num = self.cell_size = self.om_channel_size = None
        self.self_state_dim, self.agent_state_dim, self.agent_type_state_dim = 6, 7, 0
        self.joint_state_dim = 13
        self._net = None
        self._weights = None

    # ---------------------------------------------------------------- configuration
    def configure(self, config):
        """cadrl.py:72-81 + sarl.py:90-130"""
        from .train import SarlModule
        self.gamma = config.getfloat("rl", "gamma")
        self.kinematics = config.get("action_space", "kinematics")
        self.sampling = config.get("action_space", "sampling")
        self.speed_samples = config.getint("action_space", "speed_samples")
        self.rotation_samples = config.getint("action_space", "rotation_samples")
        self.query_env = config.getboolean("action_space", "query_env")
        self.cell_num = config.getint("om", "cell_num")
        self.cell_size = config.getfloat("om", "cell_size")
        self.om_channel_size = config.getint("om", "om_channel_size")
        self.with_om = config.getboolean("sarl", "with_om")
        if self.with_om:
            raise NotImplementedError("occupancy maps (OM-SARL) are outside the accelerated path")
        if not self.query_env:
            raise NotImplementedError("query_env = false ends in the reference's own NotImplementedError "
                                      "(multi_human_rl.py:89-91)")
        if config.has_option("sarl", "with_agent_type"):
            self.with_agent_type = config.getboolean("sarl", "with_agent_type")
            self.agent_type_state_dim = 4 if self.with_agent_type else 0
            self.joint_state_dim = self.self_state_dim + self.agent_state_dim + self.agent_type_state_dim
        self.model = SarlModule(self.input_dim(), _dims(config, "mlp1_dims"), _dims(config, "mlp2_dims"),
                                _dims(config, "mlp3_dims"), _dims(config, "attention_dims"),
                                config.getboolean("sarl", "with_global_state"), self.self_state_dim)
        self.multiagent_training = config.getboolean("sarl", "multiagent_training")

    def input_dim(self):
        return self.joint_state_dim

    def set_device(self, device):
        self.device = torch.device(device) if not isinstance(device, torch.device) else device
        self.model.to(self.device)
        self._net = None

    def set_epsilon(self, epsilon):
        self.epsilon = epsilon

    def get_attention_weights(self):
        """sarl.py:134-135: the weights of the network's LAST forward, i.e. of the last candidate action."""
        return self._weights

    def build_action_space(self, v_pref):
        """cadrl.py:91-116"""
        rows = build_action_space(v_pref, self.kinematics, self.speed_samples, self.rotation_samples)
        make = ActionXY if self.kinematics == "holonomic" else ActionRot
        self.action_space = [make(float(a), float(b)) for a, b in rows]
        self.speeds = sorted({float(np.hypot(a, b)) for a, b in rows[1:]}) if self.kinematics == "holonomic" \
            else sorted({float(a) for a, _ in rows[1:]})
        self.rotations = (np.linspace(0, 2 * np.pi, self.rotation_samples, endpoint=False)
                          if self.kinematics == "holonomic"
                          else np.linspace(-np.pi / 4, np.pi / 4, self.rotation_samples))
        self._action_rows = rows

    def _value_net(self):
        """Inference view of the model's current weights (packed for the matrix cores on a HIP device);
        rebuilt when the weights were replaced (load_state_dict) or moved."""
        version = tuple(p._version for p in self.model.parameters())
        if self._net is None or self._net_version != version:
            from .sarl import SarlValueNet
            self._net = SarlValueNet({k: v.detach() for k, v in self.model.state_dict().items()},
                                     device=str(self.device), with_global_state=self.model.with_global_state,
                                     self_state_dim=self.self_state_dim)
            self._net_version = version
        return self._net

    # ---------------------------------------------------------------- decision
    def predict(self, state, env=None):
        """multi_human_rl.py:12-87"""
        if self.phase is None:
            raise AttributeError("Phase attribute has to be set!")
        if self.device is None:
            raise AttributeError("Device attributes has to be set!")
        if self.phase == "train" and self.epsilon is None:
            raise AttributeError("Epsilon attribute has to be set in training phase")
        if self.reach_destination(state):
            return ActionXY(0, 0) if self.kinematics == "holonomic" else ActionRot(0, 0)
        if self.action_space is None:
            self.build_action_space(state.self_state.v_pref)
        if env is None or not hasattr(env, "lookahead_all"):
            raise ValueError("SARL.predict needs the env it acts in: robot.act(ob, env=env)")
        probability = np.random.random()
        if self.phase == "train" and probability < self.epsilon:
            chosen = self.action_space[np.random.choice(len(self.action_space))]
        else:
            sweep = env.lookahead_all(self._action_rows)
            rows = torch.from_numpy(sweep["rows_rotated"][:, :sweep["n_rows"]]).to(self.device)
            net = self._value_net()
            # the attention weights of the network's last forward in the reference = of the last action
            _, w = net.forward(rows[-1:], want_weights=True, exact=True)
            self._weights = w[0].cpu().numpy()
            discount = pow(self.gamma, self.time_step * state.self_state.v_pref)
            reward = torch.from_numpy(np.ascontiguousarray(sweep["reward"], dtype=np.float64)).to(self.device)
            values = net.action_values(rows[None], reward[None], discount)[0].cpu().numpy()
            self.action_values = [float(x) for x in values]
            if np.isnan(values).all():
                raise ValueError("Value network is not well trained. ")
            chosen = self.action_space[int(np.nanargmax(values))]  # first maximum, like `value > max_value`
        if self.phase == "train":
            self.last_state = self.transform(state, env)
        return chosen

    def transform(self, state, env=None):
        """multi_human_rl.py:128-149: the rotated joint state [rows, T].  With the env: the rows the env keeps
        for its current state (the kernels' rotate()).  Without — the reference's one-argument call, e.g.
        Explorer.update_memory's `target_policy.transform(state)` (rl/utils/explorer.py:162) — from `state`
        itself on the host, float32 like the reference's tensors."""
        if env is not None:
            return torch.from_numpy(env.observe_rotated()).to(self.device)
        rows = torch.Tensor([tuple(state.self_state + other) for other in state.agent_states])
        return self.rotate(rows).to(self.device)

    def rotate(self, rows):
        """cadrl.py:236-337 on [n, 15] float32 rows (robot FullState 9 | other ObservableState 5 + type): the
        robot-centric frame with the x axis towards the goal."""
        sx, sy, svx, svy, sr, gx, gy, vpref, theta = (rows[:, c] for c in range(9))
        ox, oy, ovx, ovy, orad = (rows[:, c] for c in range(9, 14))
        ex, ey = gx - sx, gy - sy
        ang = torch.atan2(ey, ex)
        c, s_ = torch.cos(ang), torch.sin(ang)
        col = lambda v: v.reshape(-1, 1)  # noqa: E731
        heading = col(theta - ang) if self.kinematics == "unicycle" else torch.zeros_like(col(vpref))
        rx, ry = ox - sx, oy - sy
        parts = [torch.norm(torch.cat([col(ex), col(ey)], 1), 2, dim=1, keepdim=True), col(vpref), heading, col(sr),
                 col(svx * c + svy * s_), col(svy * c - svx * s_),
                 col(rx * c + ry * s_), col(ry * c - rx * s_),
                 col(ovx * c + ovy * s_), col(ovy * c - ovx * s_), col(orad),
                 torch.norm(torch.cat([col(sx - ox), col(sy - oy)], 1), 2, dim=1, keepdim=True), col(sr) + col(orad)]
        if self.with_agent_type:
            parts.append(torch.nn.functional.one_hot(rows[:, 14].long(), num_classes=self.agent_type_state_dim))
        return torch.cat(parts, dim=1)
