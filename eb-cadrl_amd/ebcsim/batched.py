"""BatchedEnv — E independent scenes behind the C ABI (include/ebcsim.h).

The struct-of-arrays counterpart of EntityBasedCollisionAvoidance (simulator/env.py):
reset() uploads host-built scenes, step() is one env.step(update=True) for every env,
lookahead() the |A|-way onestep_lookahead sweep.  Host calls take/return numpy arrays;
the *_device calls take torch CUDA tensors and only enqueue work on the current stream.
"""
import ctypes as C

import numpy as np

from . import _abi, _capi


def _np_ptr(a):
    return None if a is None else a.ctypes.data


class BatchedEnv:
    def __init__(self, params, n_envs, max_humans, max_static=0, device=0):
        self._h = C.c_void_p()
        self.params = params
        L = _capi.lib()
        _capi.check(L.ebc_create(int(device), int(n_envs), int(max_humans), int(max_static),
                                 C.addressof(params), C.byref(self._h)))
        self._L = L
        self.device = int(device)
        self.E, self.N, self.S = int(n_envs), int(max_humans), int(max_static)
        self.R = self.N + self.S
        self.T = _abi.rot_width(params)
        self.G = int(round(params.map_size_m / params.map_resolution))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._L.ebc_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ reset
    def _scene_struct(self, scene, keep):
        def arr(a, dtype):
            a = np.ascontiguousarray(a, dtype=dtype)
            keep.append(a)
            return a.ctypes.data

        if scene.N != self.N or scene.S != self.S:
            raise ValueError("scene batch padded to (%d, %d), env expects (%d, %d)"
                             % (scene.N, scene.S, self.N, self.S))
        sc = _abi.EbcScene()
        sc.struct_size = C.sizeof(sc)
        sc.n = scene.n
        sc.n_humans = arr(scene.n_humans, np.int32)
        for k in ("px", "py", "vx", "vy", "gx", "gy", "radius", "v_pref"):
            setattr(sc, k, arr(getattr(scene, k), np.float64))
        sc.type = arr(scene.type, np.uint8)
        if self.S:
            sc.n_static = arr(scene.n_static, np.int32)
            sc.spx, sc.spy, sc.sradius = (arr(scene.spx, np.float64), arr(scene.spy, np.float64),
                                          arr(scene.sradius, np.float64))
        if scene.grid is not None:
            if scene.grid.shape[1:] != (self.G, 2):
                raise ValueError("grid must be [n][%d][2] uint64" % self.G)
            sc.grid = arr(scene.grid, np.uint64)
        sc.robot = arr(scene.robot, np.float64)
        return sc

    def reset(self, scene, env_ids=None):
        """scene: ebcsim.scene.SceneBatch padded to (max_humans, max_static)."""
        keep = []
        sc = self._scene_struct(scene, keep)
        ids = None
        if env_ids is not None:
            ids = np.ascontiguousarray(env_ids, dtype=np.int32)
            if len(ids) != scene.n:
                raise ValueError("len(env_ids) != scene.n")
        _capi.check(self._L.ebc_reset(self._h, _np_ptr(ids), C.addressof(sc)))
        if not hasattr(self, "n_static_host"):
            self.n_static_host = np.zeros(self.E, dtype=np.int64)
        if self.S:
            self.n_static_host[np.arange(scene.n) if ids is None else ids] = scene.n_static
        self._note_rows(scene)

    def _note_rows(self, scene):
        """`ragged`: some scene this handle may run has fewer observation rows than R (known on the host
        from what was uploaded; which env runs which scene is the device's business: row_counts_device)."""
        rows = np.asarray(scene.n_humans, dtype=np.int64) + (np.asarray(scene.n_static, dtype=np.int64) if self.S else 0)
        self.ragged = bool(getattr(self, "ragged", False) or (rows < self.R).any())

    def set_scene_pool(self, scene, stride=None):
        """Install P = scene.n host-generated scenes as the auto-reset pool; env e walks scenes
        e mod P, + stride, ... (default stride = E: disjoint walks when P is a multiple of E)."""
        keep = []
        sc = self._scene_struct(scene, keep)
        _capi.check(self._L.ebc_set_scene_pool(self._h, C.addressof(sc), int(self.E if stride is None else stride)))
        self._note_rows(scene)

    # ------------------------------------------------------------------ scenes generated on the device
    # (SceneGenerator.generate_random_scene, scene_generator.py:330-378; gen = scene.gen_struct(cfg, phase))
    @staticmethod
    def _seeds(seeds, n):
        """(seed0, pointer, keep-alive) for `seeds` = first seed (int: seed0 + i) or an array of n seeds."""
        if np.isscalar(seeds):
            return int(seeds) & 0xFFFFFFFF, None, None
        a = np.ascontiguousarray(seeds, dtype=np.uint32)
        if a.shape != (n,):
            raise ValueError("seeds must be one int or %d of them" % n)
        return 0, a.ctypes.data, a

    def _note_generated(self, gen):
        humans = sum(gen.count)
        self.ragged = bool(getattr(self, "ragged", False) or humans < self.N or gen.num_walls > 0
                           or gen.num_circles < self.S)

    def generate_scenes(self, gen, seeds, n):
        """n generated scenes copied back as a scene.SceneBatch (what generate_scene + from_scenes build on the host)."""
        from .scene import SceneBatch
        seed0, ptr, keep = self._seeds(seeds, n)
        N, S, G = self.N, max(self.S, 1), self.G
        f = lambda *sh: np.zeros(sh, dtype=np.float64)  # noqa: E731
        b = SceneBatch(n, self.N, self.S, np.zeros(n, np.int32), f(n, N), f(n, N), f(n, N), f(n, N), f(n, N), f(n, N),
                       f(n, N), f(n, N), np.zeros((n, N), np.uint8), np.zeros(n, np.int32), f(n, S), f(n, S), f(n, S),
                       np.zeros((n, G, 2), np.uint64), f(n, 9))
        out = _abi.EbcSceneOut()
        out.struct_size = C.sizeof(out)
        out.location = _abi.HOST
        for k in ("n_humans", "px", "py", "vx", "vy", "gx", "gy", "radius", "v_pref", "type", "n_static", "spx", "spy",
                  "sradius", "grid", "robot"):
            setattr(out, k, getattr(b, k).ctypes.data)
        _capi.check(self._L.ebc_generate_scenes(self._h, C.addressof(gen), seed0, ptr, int(n), C.addressof(out)))
        if gen.num_circles + gen.num_walls == 0:
            b.grid = None
        return b

    def generate_reset(self, gen, seeds, first=0, n=None):
        """env.reset of envs first..first+n-1 from scenes generated on the device (only the seeds cross PCIe)."""
        n = self.E - first if n is None else n
        seed0, ptr, keep = self._seeds(seeds, n)
        _capi.check(self._L.ebc_generate_reset(self._h, C.addressof(gen), seed0, ptr, int(first), int(n)))
        self._note_generated(gen)

    def generate_pool(self, gen, seeds, n, stride=None):
        """set_scene_pool with n scenes generated on the device."""
        seed0, ptr, keep = self._seeds(seeds, n)
        _capi.check(self._L.ebc_generate_pool(self._h, C.addressof(gen), seed0, ptr, int(n),
                                              int(self.E if stride is None else stride)))
        self._note_generated(gen)

    # ------------------------------------------------------------------ host calls
    def set_human_actions(self, act):
        a = np.ascontiguousarray(act, dtype=np.float64).reshape(self.E, self.N, 2)
        _capi.check(self._L.ebc_set_human_actions(self._h, _abi.HOST, a.ctypes.data))

    def step(self, robot_action=None, human_policy=_abi.HUMAN_ORCA,
             robot_policy=_abi.ROBOT_EXTERNAL, flags=0, border=None, outputs=None):
        E, N, R, T = self.E, self.N, self.R, self.T
        out = dict(reward=np.zeros(E), done=np.zeros(E, np.uint8), info=np.zeros(E, np.uint8),
                   dmin=np.zeros((E, 3)), dist_to_goal=np.zeros(E),
                   robot_action_out=np.zeros((E, 2)), human_action=np.zeros((E, N, 2)),
                   ob=np.zeros((E, R, 5)), obs_rotated=np.zeros((E, R, T), np.float32))
        if outputs is not None:
            out = {k: v for k, v in out.items() if k in outputs}
        args = _abi.EbcStepArgs()
        args.struct_size = C.sizeof(args)
        args.location = _abi.HOST
        args.human_policy, args.robot_policy = int(human_policy), int(robot_policy)
        ra = b = None
        if robot_action is not None:
            ra = np.ascontiguousarray(robot_action, dtype=np.float64).reshape(E, 2)
            args.robot_action = ra.ctypes.data
        if border is not None:
            b = np.ascontiguousarray(border, dtype=np.float64).reshape(4)
            args.border = b.ctypes.data
            flags |= _abi.FLAG_BORDER
        args.flags = int(flags)
        for k, v in out.items():
            setattr(args, k, v.ctypes.data)
        _capi.check(self._L.ebc_step(self._h, C.addressof(args)))
        return out

    def lookahead(self, actions, human_policy=_abi.HUMAN_ORCA, flags=0, border=None,
                  want_rows=True):
        actions = np.ascontiguousarray(actions, dtype=np.float64).reshape(-1, 2)
        A, E, R, T = len(actions), self.E, self.R, self.T
        out = dict(reward=np.zeros((E, A)), done=np.zeros((E, A), np.uint8),
                   info=np.zeros((E, A), np.uint8), dmin=np.zeros((E, A, 3)),
                   next_ob=np.zeros((E, R, 5)))
        if want_rows:
            out["rows_rotated"] = np.zeros((E, A, R, T), np.float32)
        args = _abi.EbcLookaheadArgs()
        args.struct_size = C.sizeof(args)
        args.location = _abi.HOST
        args.human_policy, args.n_actions = int(human_policy), A
        args.actions = actions.ctypes.data
        b = None
        if border is not None:
            b = np.ascontiguousarray(border, dtype=np.float64).reshape(4)
            args.border = b.ctypes.data
            flags |= _abi.FLAG_BORDER
        args.flags = int(flags)
        for k, v in out.items():
            setattr(args, k, v.ctypes.data)
        _capi.check(self._L.ebc_lookahead(self._h, C.addressof(args)))
        return out

    def observe(self):
        """(ob [E, R, 5], obs_rotated [E, R, T]) of the current state (what reset() returns)."""
        ob = np.zeros((self.E, self.R, 5))
        obs = np.zeros((self.E, self.R, self.T), np.float32)
        _capi.check(self._L.ebc_observe(self._h, _abi.HOST, ob.ctypes.data, obs.ctypes.data))
        return ob, obs

    def observe_device(self, obs_rotated):
        """Rotated observation of the current state into a torch CUDA tensor [E, R, T]."""
        _capi.check(self._L.ebc_observe(self._h, _abi.DEVICE, None, obs_rotated.data_ptr()))

    def row_counts(self):
        """Observation rows that exist per env, int64 [E] (ebc_row_counts), from the device state."""
        n = np.zeros(self.E, dtype=np.int64)
        _capi.check(self._L.ebc_row_counts(self._h, _abi.HOST, n.ctypes.data))
        return n

    def row_counts_device(self, n_rows):
        """The same into a torch CUDA int64 [E] tensor (enqueued on the handle's stream)."""
        if n_rows.dtype.itemsize != 8 or n_rows.numel() != self.E or not n_rows.is_contiguous():
            raise ValueError("n_rows must be a contiguous int64 [E] tensor")
        _capi.check(self._L.ebc_row_counts(self._h, _abi.DEVICE, n_rows.data_ptr()))

    def robot_orca(self, safety_space=0.0):
        """ORCA.predict with the robot as the agent, every env (ebc_robot_orca) -> actions [E, 2]:
        the imitation-learning demonstrator (rl/train.py:99-143)."""
        act = np.zeros((self.E, 2))
        _capi.check(self._L.ebc_robot_orca(self._h, float(safety_space), _abi.HOST, act.ctypes.data))
        return act

    def robot_orca_sim(self, enable=True):
        """The demonstrator's persistent rvo2 simulator (simulator/policy/orca.py:96-133), one per env, all "not built
        yet" (a fresh policy object); enable=False: every robot_orca call from the current state alone."""
        _capi.check(self._L.ebc_robot_orca_sim(self._h, 1 if enable else 0))
        self._robot_sim = bool(enable)

    def robot_orca_sim_state(self, state=None):
        """Read (state None) or replace the simulators: dict(rows int32 [E], radius float32 [E, R], self float32 [E, 2])."""
        if state is not None and not getattr(self, "_robot_sim", False):
            self.robot_orca_sim(True)
        rows = np.zeros(self.E, np.int32) if state is None else np.ascontiguousarray(state["rows"], np.int32).reshape(self.E)
        rad = (np.zeros((self.E, self.R), np.float32) if state is None
               else np.ascontiguousarray(state["radius"], np.float32).reshape(self.E, self.R))
        me = np.zeros((self.E, 2), np.float32) if state is None else np.ascontiguousarray(state["self"], np.float32).reshape(self.E, 2)
        _capi.check(self._L.ebc_robot_orca_sim_state(self._h, _abi.HOST, 0 if state is None else 1, rows.ctypes.data,
                                                     rad.ctypes.data, me.ctypes.data))
        return dict(rows=rows, radius=rad, self=me)

    def robot_orca_device(self, actions, safety_space=0.0):
        """The same into a torch CUDA tensor float64 [E, 2] (enqueued on the handle's stream)."""
        if actions.dtype.itemsize != 8 or actions.numel() != self.E * 2 or not actions.is_contiguous():
            raise ValueError("actions must be a contiguous float64 [E, 2] tensor")
        _capi.check(self._L.ebc_robot_orca(self._h, float(safety_space), _abi.DEVICE, actions.data_ptr()))

    def get_state(self):
        E, N = self.E, self.N
        f = lambda *s: np.zeros(s)  # noqa: E731
        out = dict(px=f(E, N), py=f(E, N), vx=f(E, N), vy=f(E, N), gx=f(E, N), gy=f(E, N),
                   radius=f(E, N), v_pref=f(E, N), type=np.zeros((E, N), np.uint8),
                   n_humans=np.zeros(E, np.int32), robot=f(E, 9), global_time=f(E),
                   arrival_time=f(E, N), done=np.zeros(E, np.uint8))
        v = _abi.EbcStateView()
        v.struct_size = C.sizeof(v)
        v.location = _abi.HOST
        for k, a in out.items():
            setattr(v, k, a.ctypes.data)
        _capi.check(self._L.ebc_get_state(self._h, C.addressof(v)))
        return out

    # ------------------------------------------------------------------ device calls
    def use_torch_stream(self):
        """Run the kernels on torch's current CUDA(HIP) stream of this device."""
        import torch
        stream = torch.cuda.current_stream(self.device).cuda_stream
        _capi.check(self._L.ebc_set_stream(self._h, C.c_void_p(stream)))

    def alloc_step_outputs(self, keys=("reward", "done", "info", "obs_rotated")):
        """torch CUDA tensors for step_device(); the caller owns them."""
        import torch
        dev = torch.device("cuda", self.device)
        E, N, R, T = self.E, self.N, self.R, self.T
        shapes = dict(reward=((E,), torch.float64), done=((E,), torch.uint8),
                      info=((E,), torch.uint8), dmin=((E, 3), torch.float64),
                      dist_to_goal=((E,), torch.float64),
                      robot_action_out=((E, 2), torch.float64),
                      human_action=((E, N, 2), torch.float64), ob=((E, R, 5), torch.float64),
                      obs_rotated=((E, R, T), torch.float32))
        return {k: torch.zeros(shapes[k][0], dtype=shapes[k][1], device=dev) for k in keys}

    def step_device(self, outputs, robot_action=None, human_policy=_abi.HUMAN_ORCA,
                    robot_policy=_abi.ROBOT_EXTERNAL, flags=0):
        """Enqueue one step; `outputs`/`robot_action` are torch CUDA tensors (not copied).  The
        argument block is cached per (tensor addresses, policies, flags): a training loop that
        re-uses its buffers pays one ctypes call per step."""
        key = (tuple((k, t.data_ptr()) for k, t in outputs.items()),
               None if robot_action is None else robot_action.data_ptr(),
               int(human_policy), int(robot_policy), int(flags))
        cache = self.__dict__.setdefault("_step_args", {})
        args = cache.get(key)
        if args is None:
            args = _abi.EbcStepArgs()
            args.struct_size = C.sizeof(args)
            args.location = _abi.DEVICE
            args.human_policy, args.robot_policy = int(human_policy), int(robot_policy)
            args.flags = int(flags)
            if robot_action is not None:
                if robot_action.dtype.itemsize != 8 or robot_action.numel() != self.E * 2:
                    raise ValueError("robot_action must be float64 [E, 2]")
                if not robot_action.is_contiguous():
                    raise ValueError("robot_action must be contiguous")
                args.robot_action = robot_action.data_ptr()
            for k, t in outputs.items():
                setattr(args, k, t.data_ptr())
            if len(cache) > 64:
                cache.clear()
            cache[key] = args
        rc = self._L.ebc_step(self._h, C.addressof(args))
        if rc:
            _capi.check(rc)

    _STEP_K_SHAPES = staticmethod(lambda E, R, T: dict(
        state_rotated=((E, R, T), "float32"), n_rows=((E,), "int64"), robot_action_out=((E, 2), "float64"),
        reward=((E,), "float64"), done=((E,), "uint8"), info=((E,), "uint8"), dmin=((E, 3), "float64"),
        dist_to_goal=((E,), "float64"), obs_rotated=((E, R, T), "float32")))

    def _step_k_args(self, K, location, ptr, human_policy, robot_policy, flags, robot_safety_space, robot_action_ptr):
        args = _abi.EbcStepKArgs()
        args.struct_size = C.sizeof(args)
        args.location, args.K = location, int(K)
        args.human_policy, args.robot_policy, args.flags = int(human_policy), int(robot_policy), int(flags)
        args.robot_safety_space = float(robot_safety_space)
        args.robot_action = robot_action_ptr
        for k, v in ptr.items():
            setattr(args, k, v)
        return args

    def step_k(self, K, keys=("reward", "done", "info", "state_rotated"), robot_action=None,
               human_policy=_abi.HUMAN_ORCA, robot_policy=_abi.ROBOT_LINEAR, flags=0, robot_safety_space=0.0):
        """K steps in one call (ebc_step_k), host buffers: returns {key: array [K, ...]}."""
        shapes = self._STEP_K_SHAPES(self.E, self.R, self.T)
        out = {k: np.zeros((K,) + shapes[k][0], dtype=shapes[k][1]) for k in keys}
        ra = None
        if robot_action is not None:
            ra = np.ascontiguousarray(robot_action, dtype=np.float64).reshape(K, self.E, 2)
        args = self._step_k_args(K, _abi.HOST, {k: v.ctypes.data for k, v in out.items()}, human_policy, robot_policy,
                                 flags, robot_safety_space, None if ra is None else ra.ctypes.data)
        _capi.check(self._L.ebc_step_k(self._h, C.addressof(args)))
        return out

    def alloc_step_k_outputs(self, K, keys=("reward", "done", "info", "state_rotated")):
        """torch CUDA tensors [K, ...] for step_k_device(); the caller owns them."""
        import torch
        dev = torch.device("cuda", self.device)
        shapes = self._STEP_K_SHAPES(self.E, self.R, self.T)
        return {k: torch.zeros((K,) + shapes[k][0], dtype=getattr(torch, shapes[k][1]), device=dev) for k in keys}

    def step_k_device(self, outputs, K, robot_action=None, human_policy=_abi.HUMAN_ORCA,
                      robot_policy=_abi.ROBOT_LINEAR, flags=0, robot_safety_space=0.0):
        """Enqueue K steps (ebc_step_k) writing into torch CUDA tensors [K, ...] (not copied)."""
        for k, t in outputs.items():
            if t.shape[0] != K or not t.is_contiguous():
                raise ValueError("%s must be a contiguous [K, ...] tensor" % k)
        if robot_action is not None and (robot_action.dtype.itemsize != 8 or robot_action.numel() != K * self.E * 2
                                         or not robot_action.is_contiguous()):
            raise ValueError("robot_action must be a contiguous float64 [K, E, 2] tensor")
        args = self._step_k_args(K, _abi.DEVICE, {k: t.data_ptr() for k, t in outputs.items()}, human_policy,
                                 robot_policy, flags, robot_safety_space,
                                 None if robot_action is None else robot_action.data_ptr())
        _capi.check(self._L.ebc_step_k(self._h, C.addressof(args)))

    def alloc_lookahead_outputs(self, n_actions, keys=("reward", "done", "info", "rows_rotated")):
        """torch CUDA tensors for lookahead_device(); the caller owns them."""
        import torch
        dev = torch.device("cuda", self.device)
        E, R, T, A = self.E, self.R, self.T, int(n_actions)
        shapes = dict(reward=((E, A), torch.float64), done=((E, A), torch.uint8),
                      info=((E, A), torch.uint8), dmin=((E, A, 3), torch.float64),
                      next_ob=((E, R, 5), torch.float64), rows_rotated=((E, A, R, T), torch.float32))
        return {k: torch.zeros(shapes[k][0], dtype=shapes[k][1], device=dev) for k in keys}

    def lookahead_device(self, actions, outputs, human_policy=_abi.HUMAN_ORCA, flags=0):
        """Enqueue the |A|-way sweep; `actions` float64 [A, 2] and `outputs` are torch CUDA tensors."""
        args = _abi.EbcLookaheadArgs()
        args.struct_size = C.sizeof(args)
        args.location = _abi.DEVICE
        args.human_policy, args.n_actions = int(human_policy), int(actions.shape[0])
        args.flags = int(flags)
        if actions.dtype.itemsize != 8 or actions.dim() != 2 or actions.shape[1] != 2:
            raise ValueError("actions must be float64 [A, 2]")
        args.actions = actions.data_ptr()
        for k, t in outputs.items():
            setattr(args, k, t.data_ptr())
        _capi.check(self._L.ebc_lookahead(self._h, C.addressof(args)))

    def synchronize(self):
        _capi.check(self._L.ebc_synchronize(self._h))

    def timing(self, enable=True):
        _capi.check(self._L.ebc_timing(self._h, int(enable)))

    def timing_read(self, reset=True):
        ms, n = C.c_double(), C.c_int64()
        _capi.check(self._L.ebc_timing_read(self._h, int(reset), C.byref(ms), C.byref(n)))
        return ms.value, n.value
