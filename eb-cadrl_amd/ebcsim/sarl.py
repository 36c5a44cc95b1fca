"""The robot's decision on device: look-ahead sweep + SARL value network + argmax.

What MultiHumanRL.predict does per decision in the reference (rl/policy/multi_human_rl.py:12-87):
for each of the 81 actions, env.onestep_lookahead -> rotate -> model -> value.  Here the sweep is
one ebc_lookahead launch that leaves rows_rotated[E][A][R][T] in HBM, and the value network
(rl/policy/sarl.py:38-82) runs on those rows as batched torch GEMMs (hipBLASLt on ROCm).  The
network is rebuilt functionally from the reference's state_dict, so its .pth files load as they
are.  Rows beyond an env's n_humans + n_static are masked out of the mean / softmax (the
reference never creates them)."""
import contextlib

import numpy as np
import torch

from . import _abi


def uniform_v_pref(env):
    """The robots' preferred speed, which the whole batch must share: the reference builds the action space from it and
    discounts by gamma^(dt * v_pref) PER ROBOT (multi_human_rl.py:36-60, 72-76); a batch here has one action space and
    one discount, so robots that differ (a scene pool loaded from files can) are refused instead of silently taking
    env 0's."""
    v = np.asarray(env.get_state()["robot"], dtype=np.float64)[:, 7]
    if float(v.min()) != float(v.max()):
        raise ValueError("the robots of this batch differ in v_pref (%g .. %g): one action space and one discount per "
                         "batch — split the batch by v_pref" % (float(v.min()), float(v.max())))
    return float(v[0])


def _mlp(x, layers, last_relu):
    """Linear (+ ReLU) stack.  On the GPU the bias and the ReLU ride in the GEMM's epilogue
    (torch._addmm_activation -> hipBLASLt): same values, no separate pass over the activations."""
    for i, (w, b) in enumerate(layers):
        relu = i != len(layers) - 1 or last_relu
        if (relu and x.is_cuda and x.dim() == 2 and not torch.is_grad_enabled()
                and hasattr(torch, "_addmm_activation")):  # inference only: the fused op has no derivative
            x = torch._addmm_activation(b, x, w.t(), use_gelu=False)
        else:
            x = torch.nn.functional.linear(x, w, b)
            if relu:
                x = torch.relu(x)
    return x


class _NativeMlp2(object):
    """A two-layer block on the bf16 matrix cores with split operands (libebcsim ebc_mlp2_*)."""

    def __init__(self, layers, device_index, final=None, in_fragments=False):
        """layers: [(w1, b1), (w2, b2)]; final: optional (w3 [1, O], b3 [1]) third layer with one output.
        in_fragments: the block's input is the fragment tensor another block left with `frag_out` (ebc_mlp2_forward_ex)."""
        import ctypes as C
        from . import _capi
        (w1, b1), (w2, b2) = layers
        self._L, self._C = _capi.lib(), C
        self.K0, self.H, self.O = int(w1.shape[1]), int(w1.shape[0]), int(w2.shape[0])
        host = [t.detach().to("cpu", torch.float32).contiguous().numpy() for t in (w1, b1, w2, b2)]
        self.has_final = final is not None
        if final is not None:
            host += [final[0].detach().to("cpu", torch.float32).reshape(-1).contiguous().numpy(),
                     final[1].detach().to("cpu", torch.float32).reshape(-1).contiguous().numpy()]
        self._h = C.c_void_p()
        self.in_fragments = bool(in_fragments)
        _capi.check(self._L.ebc_mlp2_create_ex(int(device_index), self.K0, self.H, self.O, host[0].ctypes.data,
                                               host[1].ctypes.data, host[2].ctypes.data, host[3].ctypes.data,
                                               host[4].ctypes.data if final is not None else None,
                                               host[5].ctypes.data if final is not None else None,
                                               _abi.MLP_IN_FRAGMENTS if in_fragments else 0, C.byref(self._h)))

    @staticmethod
    def frag_buffer(M, width, device):
        """Storage for an [M, width] activation handed on in fragment order: [row tiles][column tiles][2][2][64] x 16 B."""
        return torch.empty(((M + 31) // 32, (width + 31) // 32, 2, 2, 64, 4), dtype=torch.int32, device=device)

    def forward_ex(self, M, relu_out, x=None, frag_in=None, row_bias=None, group_rows=0, want_y=True, seg_rows=0,
                   row_weight=None, want_partial=False, frag_out=None, general=False):
        """Every form of the forward in one call (ebc_mlp2_forward_ex): rows or fragments in; rows, per-group partial
        sums and / or fragments out.  -> (y or None, partial or None)."""
        from . import _capi
        dev = (x if x is not None else frag_in).device
        a = _abi.EbcMlpArgs()
        a.struct_size = self._C.sizeof(a)
        a.M, a.relu_out, a.group_rows, a.seg_rows = int(M), int(bool(relu_out)), int(group_rows), int(seg_rows)
        a.flags = _abi.MLP_GENERAL_KERNEL if general else 0  # the general block where a specialised kernel exists
        keep = []
        if x is not None:
            x = x.contiguous(); keep.append(x); a.x = x.data_ptr()
        if frag_in is not None:
            a.frag_in = frag_in.data_ptr()
        if row_bias is not None:
            row_bias = row_bias.contiguous(); keep.append(row_bias); a.row_bias = row_bias.data_ptr()
        if row_weight is not None:
            row_weight = row_weight.contiguous(); keep.append(row_weight); a.row_weight = row_weight.data_ptr()
        y = partial = None
        if want_y:
            y = torch.empty((M,) if self.has_final else (M, self.O), dtype=torch.float32, device=dev)
            a.y = y.data_ptr()
        if want_partial:
            partial = torch.empty(((M + 31) // 32, 3, self.O), dtype=torch.float64, device=dev)
            a.partial = partial.data_ptr()
        if frag_out is not None:
            a.frag_out = frag_out.data_ptr()
        _capi.check(self._L.ebc_mlp2_forward_ex(self._h, torch.cuda.current_stream(dev).cuda_stream, self._C.addressof(a)))
        return y, partial

    def __call__(self, x, relu_out, row_bias=None, group_rows=0):
        from . import _capi
        x = x.contiguous()
        shape = (x.shape[0],) if self.has_final else (x.shape[0], self.O)
        y = torch.empty(shape, dtype=torch.float32, device=x.device)
        if row_bias is not None:
            row_bias = row_bias.contiguous()
        _capi.check(self._L.ebc_mlp2_forward(self._h, torch.cuda.current_stream(x.device).cuda_stream,
                                             x.data_ptr(), int(x.shape[0]), int(bool(relu_out)),
                                             None if row_bias is None else row_bias.data_ptr(), int(group_rows),
                                             y.data_ptr()))
        return y

    def update(self, layers, final=None):
        """Refresh the packed weights from DEVICE tensors of the shapes the block was created with (ebc_mlp2_update)."""
        from . import _capi
        (w1, b1), (w2, b2) = layers
        t = [x.detach().to(torch.float32).contiguous() for x in (w1, b1, w2, b2)]
        assert tuple(t[0].shape) == (self.H, self.K0) and tuple(t[2].shape) == (self.O, self.H) and t[0].is_cuda
        f = None
        if final is not None:
            f = [final[0].detach().to(torch.float32).reshape(-1).contiguous(), final[1].detach().to(torch.float32).reshape(-1).contiguous()]
        _capi.check(self._L.ebc_mlp2_update(self._h, torch.cuda.current_stream(t[0].device).cuda_stream, t[0].data_ptr(),
                                            t[1].data_ptr(), t[2].data_ptr(), t[3].data_ptr(),
                                            None if f is None else f[0].data_ptr(), None if f is None else f[1].data_ptr()))

    def f32(self, x, relu_out, row_bias=None, group_rows=0):
        """The same block in plain float32 on the vector ALUs (ebc_mlp2_forward_f32): float32-GEMM-grade values for the
        few rows that decide an argmax."""
        from . import _capi
        x = x.contiguous()
        shape = (x.shape[0],) if self.has_final else (x.shape[0], self.O)
        y = torch.empty(shape, dtype=torch.float32, device=x.device)
        if row_bias is not None:
            row_bias = row_bias.contiguous()
        _capi.check(self._L.ebc_mlp2_forward_f32(self._h, torch.cuda.current_stream(x.device).cuda_stream,
                                                 x.data_ptr(), int(x.shape[0]), int(bool(relu_out)),
                                                 None if row_bias is None else row_bias.data_ptr(), int(group_rows),
                                                 y.data_ptr()))
        return y

    def reduce(self, x, relu_out, seg_rows, row_weight=None, store=True):
        """The block with the row-group sums folded into its epilogue (ebc_mlp2_forward_reduce): -> (y or None,
        partial [ceil(M / 32)][3][O]) — ebc_pair_combine turns the partials into per-pair means / weighted sums."""
        from . import _capi
        x = x.contiguous()
        M = int(x.shape[0])
        y = torch.empty((M, self.O), dtype=torch.float32, device=x.device) if store else None
        partial = torch.empty(((M + 31) // 32, 3, self.O), dtype=torch.float64, device=x.device)
        if row_weight is not None:
            row_weight = row_weight.contiguous()
        _capi.check(self._L.ebc_mlp2_forward_reduce(self._h, torch.cuda.current_stream(x.device).cuda_stream,
                                                    x.data_ptr(), M, int(bool(relu_out)), None, 0,
                                                    None if y is None else y.data_ptr(), int(seg_rows),
                                                    None if row_weight is None else row_weight.data_ptr(),
                                                    partial.data_ptr()))
        return y, partial

    def __del__(self):
        try:
            self._L.ebc_mlp2_destroy(self._h)
        except Exception:
            pass


class SarlValueNet(object):
    """rl/policy/sarl.py:9-82 (ValueNetwork), weights from its state_dict."""

    def __init__(self, state_dict, device="cpu", with_global_state=True, self_state_dim=6,
                 dtype=torch.float32):
        """dtype: torch.float32 reproduces the reference's values (2e-4); torch.bfloat16 runs the
        GEMMs on the bf16 matrix cores: measured on MI355X 2.3x faster per decision, but values move by
        up to 0.2 and only 63 % of the envs keep the fp32 action (positions lose their digits in bf16
        inputs) -- an experiment, not a parity path."""
        self.dtype = dtype

        def stack(prefix):
            idx = sorted({int(k.split(".")[1]) for k in state_dict if k.startswith(prefix + ".")})
            return [(state_dict["%s.%d.weight" % (prefix, i)].to(device=device, dtype=dtype),
                     state_dict["%s.%d.bias" % (prefix, i)].to(device=device, dtype=dtype))
                    for i in idx]
        self.mlp1, self.mlp2 = stack("mlp1"), stack("mlp2")
        self.attention, self.mlp3 = stack("attention"), stack("mlp3")
        self.with_global_state = with_global_state
        self.self_state_dim = self_state_dim
        self.input_dim = self.mlp1[0][0].shape[1]
        self.device = torch.device(device)
        self._native = None  # built on first use: (mlp1, mlp2, attention[1:]) as fused two-layer blocks

    def _block_specs(self):
        """(layers, final) of every libebcsim block of this network, from its CURRENT tensors, or None when the blocks
        do not apply: [mlp1, mlp2, attention (h1 half of layer 0 + layer 1, layer 2 as the one-output tail),
        mlp3[:2], mlp3[2:4]] and the mean-state half of attention layer 0 as a block of its own."""
        ok = (self.device.type == "cuda" and self.dtype == torch.float32 and len(self.mlp1) == 2
              and len(self.mlp2) == 2 and len(self.attention) == 3 and self.with_global_state)
        if not ok:
            return None
        H = self.mlp1[-1][0].shape[0]
        # attention = layer 0 on cat([h1, g]) | layer 1 | layer 2 (one output): the h1 half of layer 0
        # and layer 1 form the block, g's half enters as a per-pair term, layer 2 is the block's tail
        att = [(self.attention[0][0][:, :H], torch.zeros_like(self.attention[0][1])), self.attention[1]]
        stacks = (self.mlp1, self.mlp2, att)
        if not (self.attention[2][0].shape[0] == 1 and all(
                st[0][0].shape[1] <= 224 and st[1][0].shape[0] <= 224 for st in stacks)):
            return None
        specs = [(self.mlp1, None), (self.mlp2, None), (att, self.attention[2])]
        # mlp3's first two layers (the joint vector's widest ones) as a fourth block; its tail
        # (the reference's 200 -> 200 -> 1) stays with torch unless it is a two-layer block too
        if len(self.mlp3) >= 3 and self.mlp3[0][0].shape[1] <= 224 and self.mlp3[1][0].shape[0] <= 224:
            specs.append((self.mlp3[:2], None))
            if len(self.mlp3) == 4 and self.mlp3[2][0].shape[1] <= 224 and self.mlp3[3][0].shape[0] <= 224:
                specs.append((self.mlp3[2:4], None))
        # the mean state's half of attention layer 0 (one 200 x 200 layer per PAIR) as a block too: the mean of
        # ReLU outputs is >= 0, so relu(I g) = g and [I | w0[:, H:]] is that layer in the two-layer form
        gterm = None
        if self.attention[0][0].shape[0] <= 224:
            eye = torch.eye(H, dtype=torch.float32, device=self.device)
            gterm = ([(eye, torch.zeros(H, dtype=torch.float32, device=self.device)),
                      (self.attention[0][0][:, H:], self.attention[0][1])], None)
        return specs, gterm

    def _native_blocks(self):
        """The two-layer stacks of the network as libebcsim blocks (inference on a HIP device,
        float32 weights, the reference's layer counts); None when that does not apply."""
        if getattr(self, "_native", ()) is None:  # nets assembled by hand for training opt in with enable_native()
            spec = self._block_specs()
            if spec is not None:
                idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
                # the shapes were checked above: a failure from here on is a HIP error and is raised, never
                # turned into a silent torch path
                specs, gterm = spec
                self._gterm_block = None if gterm is None else _NativeMlp2(gterm[0], idx)
                self._native = tuple(_NativeMlp2(layers, idx, final=final) for layers, final in specs)
                # the two consumers of h1 once more, taking it as the fragments mlp1 leaves (no transposition, no splitting
                # in their input phase): mlp2 and the attention stack
                self._native_frag = tuple(_NativeMlp2(layers, idx, final=final, in_fragments=True) for layers, final in specs[1:3])
            else:
                self._native = ()
        return getattr(self, "_native", ()) or None

    def enable_native(self):
        """For a network whose tensors are a training module's parameters (SarlModule.as_value_net): build the blocks
        from the current weights; refresh_native() re-packs them after the weights have changed."""
        self._native = None
        return self._native_blocks() is not None

    def refresh_native(self):
        """Re-pack every block from the network's current tensors, on the device (ebc_mlp2_update: no host copy): the
        training loop calls this once per round after its optimizer steps (rl/train.py:239-259), so its rollouts decide
        on the matrix-core blocks like every other decision.  Returns False when the network has no blocks."""
        nat = getattr(self, "_native", ()) or None
        if nat is None:
            return False
        specs, gterm = self._block_specs()
        for blk, (layers, final) in zip(nat, specs):
            blk.update(layers, final)
        for blk, (layers, final) in zip(getattr(self, "_native_frag", ()), specs[1:3]):
            blk.update(layers, final)
        if self._gterm_block is not None:
            self._gterm_block.update(gterm[0], None)
        self.native_refreshes = getattr(self, "native_refreshes", 0) + 1
        self.coarse_eps = None  # other weights: the refinement measures its bound again at the next decision
        return True

    @staticmethod
    def _pair_mean(h1, nv64, B, R):
        """sarl.py:56-58 in one pass over h1 (libebcsim ebc_pair_mean)."""
        from . import _capi
        h1 = h1.contiguous()
        g = torch.empty((B, h1.shape[1]), dtype=torch.float32, device=h1.device)
        _capi.check(_capi.lib().ebc_pair_mean(torch.cuda.current_stream(h1.device).cuda_stream, h1.data_ptr(),
                                              None if nv64 is None else nv64.data_ptr(), B, R, int(h1.shape[1]),
                                              g.data_ptr()))
        return g

    @staticmethod
    def _pair_attend(scores, feat, nv64, B, R):
        """sarl.py:69-76 in one pass over the features (libebcsim ebc_pair_attend)."""
        from . import _capi
        scores, feat = scores.contiguous(), feat.contiguous()
        out = torch.empty((B, feat.shape[2]), dtype=torch.float32, device=feat.device)
        _capi.check(_capi.lib().ebc_pair_attend(torch.cuda.current_stream(feat.device).cuda_stream, scores.data_ptr(),
                                                feat.data_ptr(), None if nv64 is None else nv64.data_ptr(), B, R,
                                                int(feat.shape[2]), out.data_ptr()))
        return out

    @staticmethod
    def _pair_combine(partial, nv64, B, R, mean):
        """Per-pair sums of a block's tile partials (libebcsim ebc_pair_combine): the pair mean (sarl.py:56-58) or,
        with the attention weights as row weights, the weighted feature sum (sarl.py:73-76)."""
        from . import _capi
        O = int(partial.shape[2])
        out = torch.empty((B, O), dtype=torch.float32, device=partial.device)
        _capi.check(_capi.lib().ebc_pair_combine(torch.cuda.current_stream(partial.device).cuda_stream, partial.data_ptr(),
                                                 None if nv64 is None else nv64.data_ptr(), B, R, O, int(bool(mean)),
                                                 out.data_ptr()))
        return out

    @staticmethod
    def _pair_weights(scores, nv64, B, R):
        """softmax' of the pair's scores (sarl.py:69-71) as row weights [B * R] (libebcsim ebc_pair_weights)."""
        from . import _capi
        scores = scores.contiguous()
        w = torch.empty((B * R,), dtype=torch.float32, device=scores.device)
        _capi.check(_capi.lib().ebc_pair_weights(torch.cuda.current_stream(scores.device).cuda_stream, scores.data_ptr(),
                                                 None if nv64 is None else nv64.data_ptr(), B, R, w.data_ptr()))
        return w

    @staticmethod
    def _pair_mask(nv64, B, R):
        from . import _capi
        w = torch.empty((B * R,), dtype=torch.float32, device=nv64.device)
        _capi.check(_capi.lib().ebc_pair_mask(torch.cuda.current_stream(nv64.device).cuda_stream, nv64.data_ptr(), B, R,
                                              w.data_ptr()))
        return w

    @classmethod
    def load(cls, path, device="cpu", **kw):
        return cls(torch.load(path, map_location="cpu"), device=device, **kw)

    def forward(self, rows, n_valid=None, want_weights=False, exact=False):
        """rows [B, R, T] float32; n_valid [B] (rows that exist) or None = all -> values [B]
        (with want_weights: (values, attention weights [B, R]), sarl.py:69-71).  exact: plain float32
        GEMMs instead of the split-bf16 matrix-core blocks (the arithmetic the reference's torch does)."""
        with torch.no_grad():
            return self._forward(rows, n_valid, want_weights, exact)

    # |value of the split-bf16 blocks - float32 value| of the NETWORK OUTPUT (before reward and discount) depends on the
    # weights: <= 3.3e-4 with the shipped eb-cadrl weights on the bench workload (its attention scores are large: a
    # 2^-16 relative error there is e-4 behind the exponential), <= 1.2e-6 with random-init weights, <= 1.5e-5 on the
    # reference's golden episodes.  The bound the refinement works with is therefore MEASURED per network
    # (calibrate_eps: both forms on a sample of the decision batch, times EPS_MARGIN; again after every
    # refresh_native) and CHECKED at every decision on the candidates that were re-evaluated: one of them off by more
    # than the bound widens it and the decision is taken again.  COARSE_EPS_MAX: a network whose blocks are worse than
    # this is refused (the matrix-core path is not fit to rank its values).
    EPS_MARGIN = 2.0
    EPS_FLOOR = 2e-6
    COARSE_EPS_MAX = 5e-3
    REFINE_CAP = 1 << 30  # no cap by default: every candidate within the bound is re-evaluated (an env near its goal can have dozens)

    def calibrate_eps(self, pairs, n_valid=None, sample=2048):
        """Measure |matrix-core value - float32 value| on a strided sample of `pairs` [n, R, T] and set the bound the
        refinement uses (coarse_eps).  One host sync."""
        n = pairs.shape[0]
        idx = torch.arange(0, n, max(1, n // int(sample)), device=pairs.device)[:int(sample)]
        sel = pairs[idx]
        nv = None if n_valid is None else n_valid[idx]
        err = float((self.forward(sel, nv) - self.forward(sel, nv, exact=True)).abs().max())
        if not err <= self.COARSE_EPS_MAX:
            raise RuntimeError("SarlValueNet: the matrix-core blocks are off by %.2e on this network: not fit to rank its values" % err)
        self.coarse_eps = max(self.EPS_MARGIN * err, self.EPS_FLOOR)
        self.measured_coarse_err = err
        return self.coarse_eps

    CHUNK_STREAMS = 2  # streams the chunks of a decision batch alternate between (1: the caller's stream only)

    def _chunk_streams(self, device):
        if self.CHUNK_STREAMS < 2:
            return ()
        pool = self.__dict__.setdefault("_side_streams", {})
        key = (str(device), self.CHUNK_STREAMS)
        if key not in pool:
            pool[key] = tuple(torch.cuda.Stream(device=device) for _ in range(self.CHUNK_STREAMS))
        return pool[key]

    def action_values(self, rows, reward, discount, n_valid=None, refine=None, chunk_pairs=None, eps=None):
        """reward + discount * V(rows) for every candidate action (multi_human_rl.py:72-76): rows
        [E, A, R, T] float32, reward [E, A] float64, n_valid [E] or None -> values [E, A] float64.
        The matrix-core blocks' values carry a split-bf16 error (coarse_eps, measured per network) and the float32
        network's best two actions can be as close as 4e-7, so the DECISION is made on float32 values: every candidate
        whose coarse value lies within 2 * discount * eps of the env's coarse best — any action whose float32 value could
        be the maximum — is re-evaluated with the float32 form of the same blocks (ebc_mlp2_forward_f32) before the
        caller takes the argmax (an env with a single such candidate is decided already and keeps its matrix-core
        value); the bound is checked on those candidates (a violation widens it and repeats the
        selection).  refine: None = this bound-driven set; an int k = the best k candidates of every env (0 = the
        matrix-core values as they are).  refine_stats counts decisions, re-evaluated candidates, envs with more than two
        of them, the largest set, and bound violations."""
        E, A, R, T = rows.shape
        step = E if not chunk_pairs else max(1, int(chunk_pairs) // A)
        v = torch.empty((E, A), dtype=torch.float32, device=rows.device)
        n_chunks = -(-E // step)
        side = self._chunk_streams(rows.device) if (n_chunks > 1 and rows.is_cuda and self._native_blocks() is not None) else ()
        if side:
            # Chunks are independent, and a third of a chunk's kernels are small (the per-pair blocks: ~110 workgroups on
            # 256 CUs, each as long as its own critical path): chunks alternate between two streams, so one chunk's
            # small kernels run beside the other's wide ones.  An even number of equal chunks, no larger than asked for.
            n_chunks += n_chunks & 1
            step = -(-E // n_chunks)
            main = torch.cuda.current_stream(rows.device)
            ready = torch.cuda.Event()
            ready.record(main)
        for i, e0 in enumerate(range(0, E, step)):
            e1 = min(E, e0 + step)
            if side:
                side[i % len(side)].wait_event(ready)
            with torch.cuda.stream(side[i % len(side)]) if side else contextlib.nullcontext():
                nv = None if n_valid is None else n_valid[e0:e1].repeat_interleave(A)
                v[e0:e1] = self.forward(rows[e0:e1].reshape(-1, R, T), nv).view(e1 - e0, A)
        for s_ in side:
            main.wait_stream(s_)
        native_select = rows.is_cuda and self._native_blocks() is not None and refine is None
        if native_select:
            if eps is not None:
                self.coarse_eps = float(eps)
            elif getattr(self, "coarse_eps", None) is None:
                self.calibrate_eps(rows.reshape(E * A, R, T), None if n_valid is None else n_valid.repeat_interleave(A))
            # values, every env's actions by value and the size of its near-best set in ONE launch (ebc_decision_rank)
            from . import _capi
            values = torch.empty((E, A), dtype=torch.float64, device=rows.device)
            order = torch.empty((E, A), dtype=torch.int32, device=rows.device)
            count_dev = torch.empty((E,), dtype=torch.int32, device=rows.device)
            reward = reward.to(torch.float64).contiguous()
            _capi.check(_capi.lib().ebc_decision_rank(torch.cuda.current_stream(rows.device).cuda_stream, v.data_ptr(), reward.data_ptr(),
                                                      float(discount), 2.0 * float(discount) * self.coarse_eps, E, A,
                                                      values.data_ptr(), order.data_ptr(), count_dev.data_ptr()))
        else:
            values = reward + discount * v.to(torch.float64)
        if not (rows.is_cuda and self._native_blocks() is not None) or refine == 0:
            return values
        if refine is not None:
            k = min(int(refine), A)
            top = torch.topk(values, k, dim=1).indices
            env_i = torch.arange(E, device=rows.device)[:, None].expand(E, k).reshape(-1)
            act_i = top.reshape(-1)
            nv = None if n_valid is None else n_valid[env_i]
            values[env_i, act_i] = reward[env_i, act_i] + discount * self.forward(rows[env_i, act_i], nv, exact=True).to(torch.float64)
            return values
        if eps is not None:
            self.coarse_eps = float(eps)
        elif getattr(self, "coarse_eps", None) is None:
            self.calibrate_eps(rows.reshape(E * A, R, T), None if n_valid is None else n_valid.repeat_interleave(A))
        st = self.__dict__.setdefault("_refine_host", {"decisions": 0, "candidates": 0, "over2": 0, "capped": 0, "max_set": 0,
                                                       "bound_violations": 0, "contested": 0})
        k = min(self.REFINE_CAP, A)
        ar = np.arange(E)
        for attempt in range(4):
            bound = 2.0 * float(discount) * self.coarse_eps
            # the candidates that could be the float32 best: a PREFIX of every env's sorted row, so the set is its length.
            # The lengths go to the host in one 8 KB copy — the one host round trip of the selection — and the index
            # lists and the counters are made there: torch.nonzero plus four .item() reads were five round trips with an
            # idle GPU and a dozen launch-bound little kernels behind each.
            if attempt == 0:
                count = count_dev.cpu().numpy().astype(np.int64)
            else:  # the bound was widened: the sizes again, from the ranked values
                ranked = values.gather(1, order.to(torch.int64))
                count = (ranked >= (ranked[:, :1] - bound)).sum(1).cpu().numpy()
            count = np.minimum(count, k)
            # an env with ONE candidate is decided: every other action's float32 value lies below that one's
            sizes = np.where(count > 1, count, 0)
            n_cand = int(sizes.sum())
            if n_cand == 0:
                exact = None
                break
            env_h = np.repeat(ar, sizes)
            slot_h = np.arange(n_cand) - np.repeat(np.cumsum(sizes) - sizes, sizes)
            idx = torch.from_numpy(np.stack([env_h, slot_h])).to(rows.device, non_blocking=True)
            env_i, slot = idx[0], idx[1]
            act_i = order[env_i, slot].to(torch.int64)
            nv = None if n_valid is None else n_valid[env_i]
            exact = self.forward(rows[env_i, act_i], nv, exact=True).to(torch.float32).contiguous()
            # the candidates' float32 values into `values`, and the bound checked where it matters: one launch, one read
            # (a widened bound below selects a superset and applies again)
            worst_dev = torch.zeros(1, dtype=torch.float32, device=rows.device)
            _capi.check(_capi.lib().ebc_decision_apply(torch.cuda.current_stream(rows.device).cuda_stream, exact.data_ptr(), v.data_ptr(),
                                                       env_i.contiguous().data_ptr(), act_i.contiguous().data_ptr(), reward.data_ptr(),
                                                       float(discount), A, n_cand, values.data_ptr(), worst_dev.data_ptr()))
            worst = float(worst_dev)
            if worst <= self.coarse_eps:
                break
            st["bound_violations"] += 1
            if not worst <= self.COARSE_EPS_MAX:
                raise RuntimeError("SarlValueNet: matrix-core values off by %.2e: not fit to rank this network's values" % worst)
            self.coarse_eps = self.EPS_MARGIN * worst
        st["decisions"] += E
        st["candidates"] += n_cand
        st["contested"] += int((count > 1).sum())
        st["over2"] += int((count > 2).sum())
        st["capped"] += int((count >= k).sum()) if k < A else 0
        st["max_set"] = max(st["max_set"], int(count.max()))
        return values

    @property
    def refine_stats(self):
        """Counters of the bound-driven re-evaluation since the network was made: decisions, re-evaluated candidates, envs
        with more than one / more than two candidates, sets cut at REFINE_CAP, the largest set, bound violations."""
        st = self.__dict__.get("_refine_host")
        return None if st is None else dict(st)

    def _forward(self, rows, n_valid=None, want_weights=False, exact=False):
        B, R, T = rows.shape
        rows = rows.to(getattr(self, "dtype", torch.float32))
        self_state = rows[:, 0, :self.self_state_dim]
        nat = None if torch.is_grad_enabled() or not rows.is_cuda or exact else self._native_blocks()
        nv64 = None if n_valid is None else n_valid.to(torch.int64).contiguous()
        # The pair reductions folded into the blocks that produce the rows (16 <= R <= 32: a 32-row tile touches at
        # most three pairs): mlp1 leaves the pair sums of h1 beside h1, the attention stack runs BEFORE mlp2, and
        # mlp2 multiplies its rows by the attention weights and leaves only their pair sums — the [B * R][F]
        # features are never written, nothing reads h1 a second time for the mean.
        folded = (nat is not None and self.with_global_state and 16 <= R <= 32 and not want_weights
                  and getattr(self, "fold_pairs", True) and nat[0].O % 4 == 0 and nat[1].O % 4 == 0)
        if folded:
            self.native_forwards = getattr(self, "native_forwards", 0) + 1
            self.folded_forwards = getattr(self, "folded_forwards", 0) + 1
            mask = None if nv64 is None else self._pair_mask(nv64, B, R)
            natf = getattr(self, "_native_frag", ()) if getattr(self, "frag_handoff", True) else ()
            H1 = nat[0].O
            if len(natf) == 2:
                # h1 never exists as float32 rows: mlp1 leaves it split and in fragment order, its consumers load that
                M = B * R
                h1f = _NativeMlp2.frag_buffer(M, H1, rows.device)
                _, part = nat[0].forward_ex(M, True, x=rows.reshape(M, T), want_y=False, seg_rows=R, row_weight=mask,
                                            want_partial=True, frag_out=h1f)
            else:
                h1, part = nat[0].reduce(rows.reshape(B * R, T), True, R, mask)
            g = self._pair_combine(part, nv64, B, R, True)
            if self._gterm_block is not None:
                gterm = self._gterm_block(g, False)
            else:
                w0, b0 = self.attention[0]
                gterm = torch.nn.functional.linear(g, w0[:, H1:], b0)
            if len(natf) == 2:
                scores, _ = natf[1].forward_ex(M, False, frag_in=h1f, row_bias=gterm, group_rows=R,
                                               general=getattr(self, "general_kernels", False))
                w = self._pair_weights(scores, nv64, B, R)
                _, part = natf[0].forward_ex(M, False, frag_in=h1f, want_y=False, seg_rows=R, row_weight=w, want_partial=True)
            else:
                scores = nat[2](h1, False, row_bias=gterm, group_rows=R)
                w = self._pair_weights(scores, nv64, B, R)
                _, part = nat[1].reduce(h1, False, R, w, store=False)
            attended = self._pair_combine(part, None, B, R, False)
            joint = torch.cat([self_state, attended], dim=1)
            if len(nat) > 4:
                return nat[4](nat[3](joint, True), False).squeeze(1)
            if len(nat) > 3:
                return _mlp(nat[3](joint, True), self.mlp3[2:], False).squeeze(1).to(torch.float32)
            return _mlp(joint, self.mlp3, False).squeeze(1).to(torch.float32)
        # exact on a HIP device: the float32 form of the same blocks (vector ALUs, fmaf into float32 sums) with the pair
        # kernels in between — the whole refinement stays inside the library (no hipBLASLt GEMMs, no element-wise launches)
        natx = None
        if (exact and rows.is_cuda and not torch.is_grad_enabled() and not want_weights and self.with_global_state
                and getattr(self, "native_exact", True)):
            natx = self._native_blocks()
            if natx is not None and (len(natx) < 5 or self._gterm_block is None or natx[0].O % 4 or natx[1].O % 4
                                     or natx[1].O > 256):  # what the pair kernels take (else: torch's float32 GEMMs)
                natx = None
        if natx is not None:
            self.native_exact_forwards = getattr(self, "native_exact_forwards", 0) + 1
            h1 = natx[0].f32(rows.reshape(B * R, T), True)
            feat = natx[1].f32(h1, False).view(B, R, -1)
            g = self._pair_mean(h1, nv64, B, R)
            gterm = self._gterm_block.f32(g, False)
            scores = natx[2].f32(h1, False, row_bias=gterm, group_rows=R).view(B, R)
            attended = self._pair_attend(scores, feat, nv64, B, R)
            joint = torch.cat([self_state, attended], dim=1)
            return natx[4].f32(natx[3].f32(joint, True), False).squeeze(1)
        if nat is not None:
            self.native_forwards = getattr(self, "native_forwards", 0) + 1  # tests assert the HIP path ran
            h1 = nat[0](rows.reshape(B * R, T), True)
            feat = nat[1](h1, False).view(B, R, -1)
        else:
            h1 = _mlp(rows.reshape(B * R, T), self.mlp1, True)
            feat = _mlp(h1, self.mlp2, False).view(B, R, -1)
        # the pair glue as two HIP kernels (plain float32 arithmetic: also behind the float32 GEMMs of the exact pass,
        # where they replace a dozen element-wise launches)
        fused = ((nat is not None or (exact and rows.is_cuda and not torch.is_grad_enabled()))
                 and h1.shape[1] % 4 == 0 and feat.shape[2] % 4 == 0 and feat.shape[2] <= 256)
        if fused or n_valid is None:
            valid = None
            denom = float(R)
        else:
            valid = (torch.arange(R, device=rows.device)[None, :] < n_valid[:, None])
            denom = n_valid.to(rows.dtype).clamp(min=1)[:, None, None]
        if self.with_global_state:
            if fused:
                g = self._pair_mean(h1, nv64, B, R)
            else:
                h1v = h1.view(B, R, -1)
                if valid is not None:
                    h1v = h1v * valid[:, :, None]
                g = h1v.sum(1) / (denom if isinstance(denom, float) else denom[:, 0, :])  # [B, H]: mean of the pair's rows
            # attention layer 1 on cat([h1, g]) without building the concatenation: the g half of the
            # weight acts once per pair, its result is added to every row of the pair
            H = h1.shape[1]
            w0, b0 = self.attention[0]
            gterm = torch.nn.functional.linear(g, w0[:, H:], b0)  # [B, A1]
            if nat is not None:
                scores = nat[2](h1, False, row_bias=gterm, group_rows=R).view(B, R)
            else:
                a1 = torch.nn.functional.linear(h1, w0[:, :H]).view(B, R, -1)
                a1 = torch.relu(a1 + gterm[:, None, :]).view(B * R, -1)
                scores = _mlp(a1, self.attention[1:], False).view(B, R)
        else:
            scores = _mlp(h1, self.attention, False).view(B, R)
        if fused:
            attended = self._pair_attend(scores, feat, nv64, B, R)
        else:
            e = torch.exp(scores) * (scores != 0).to(scores.dtype)  # the reference's masked softmax (sarl.py:69-70)
            if valid is not None:
                e = e * valid
            w = (e / e.sum(dim=1, keepdim=True)).unsqueeze(2)
            attended = (w * feat).sum(dim=1)
        joint = torch.cat([self_state, attended], dim=1)
        if nat is not None and len(nat) > 4:
            value = nat[4](nat[3](joint, True), False).squeeze(1)
        elif nat is not None and len(nat) > 3:
            value = _mlp(nat[3](joint, True), self.mlp3[2:], False).squeeze(1).to(torch.float32)
        else:
            value = _mlp(joint, self.mlp3, False).squeeze(1).to(torch.float32)
        if not want_weights:
            return value
        e = torch.exp(scores) * (scores != 0).to(scores.dtype)
        if n_valid is not None:
            e = e * (torch.arange(R, device=rows.device)[None, :] < n_valid[:, None])
        return value, e / e.sum(dim=1, keepdim=True)


class DeviceSarlPolicy(object):
    """Greedy SARL decisions for a whole BatchedEnv (phase "test": no epsilon draw)."""

    def __init__(self, net, actions, gamma, chunk_rows=1 << 21, refine=None):
        """refine: None = every candidate within the matrix-core blocks' error bound of the env's best is recomputed in
        float32 before the argmax (SarlValueNet.action_values); an int k = the best k; 0 = the matrix-core values as
        they are."""
        self.refine = None if refine is None else int(refine)
        self.net = net
        self.actions_np = np.ascontiguousarray(actions, dtype=np.float64)
        self.gamma = float(gamma)
        self.chunk_rows = int(chunk_rows)
        self._bufs = None

    def values_from(self, rows, reward, n_valid, dt, v_pref):
        """rows [E, A, R, T] float32, reward [E, A] float64 -> values [E, A] float64
        (multi_human_rl.py:72-76: reward + gamma^(dt * v_pref) * V)."""
        R = rows.shape[2]
        return self.net.action_values(rows, reward, self.gamma ** (dt * v_pref), n_valid, refine=self.refine,
                                      chunk_pairs=max(1, self.chunk_rows // R))

    def decide(self, env, human_policy=_abi.HUMAN_ORCA):
        """env: BatchedEnv on this net's device.  Returns (actions [E, 2] float64 CUDA tensor,
        values [E, A]); the human velocities stay cached for a following EBC_HUMAN_CACHED step.
        The sweep's buffers are taken when an env batch is first seen (a different batch re-allocates
        them).  The rows that exist per env are read from the DEVICE state at every decision when the
        handle may run scenes of different sizes (`env.ragged`: a ragged reset batch or scene pool), so a
        restart inside the step cannot leave them stale; `self.n_valid` is what the last decision used."""
        dev = self.net.device
        A = len(self.actions_np)
        key = (id(env), env.E, env.R, env.T)
        if self._bufs is None or getattr(self, "_env_key", None) != key:  # buffers are sized for ONE env batch
            self._env_key = key
            self._acts = torch.tensor(self.actions_np, dtype=torch.float64, device=dev)
            self._bufs = env.alloc_lookahead_outputs(A, ("reward", "rows_rotated"))
            self._n_valid = torch.full((env.E,), env.R, dtype=torch.int64, device=dev)
            self._v_pref = uniform_v_pref(env)
        self.n_valid = None
        if getattr(env, "ragged", False):
            env.row_counts_device(self._n_valid)
            self.n_valid = self._n_valid
        env.lookahead_device(self._acts, self._bufs, human_policy=human_policy)
        values = self.values_from(self._bufs["rows_rotated"], self._bufs["reward"], self.n_valid,
                                  env.params.time_step, self._v_pref)
        best = torch.argmax(values, dim=1)
        return self._acts[best], values
