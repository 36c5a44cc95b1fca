#!/usr/bin/env python3
"""bench.py — agent-steps/s of the simulation hot path on N MI355X (one process per GPU).

    python bench.py --gpus 1 --steps 500 --warmup 20
    python bench.py --gpus N ...          (starts its own N rank processes, one per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one ebc_step call (one HIP launch) over the rank's scene batch: ORCA for every human,
kinematics, swept collisions, grid window, reward, rotated observation — inputs resident in
HBM, outputs left in HBM.  Scenes are independent, so ranks own disjoint env slices and the
data path has no collective (weak scaling: per-GPU batch fixed); torch.distributed (RCCL) only
carries the barrier and the max-over-ranks of the timing.

Prints ONE JSON line (rank 0): metric/value/unit + roofline + cpu_baseline.
"""
import argparse
import configparser
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "eb-cadrl_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

WORKLOADS = {
    # name: (env config, policy config, envs per GPU)
    "metric": ("bench_metric.config", "policy_agent_type.config", 4096),
    "cfg2": ("bench_cfg2.config", "policy_plain.config", 4096),
    "cfg3": ("bench_metric.config", "policy_agent_type.config", 16384),
    "cfg4": ("bench_cfg2.config", "policy_plain.config", 16384),
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s


def algorithmic_bytes_per_env_step(N, S, T):
    """fp32-SoA accounting the roofline is priced against."""
    reads = 32 * N + 44 + 16 * S + 24
    writes = 16 * N + 20 + 18 + 4 * T * (N + S)
    return reads + writes


def build_batch(workload, envs, rank):
    from ebcsim import config as ebc_config, scene as ebc_scene, shard
    env_cfg, pol_cfg, _ = WORKLOADS[workload]
    cfg = configparser.RawConfigParser()
    cfg.read(os.path.join(PKG, "configs", env_cfg))
    pol = configparser.RawConfigParser()
    pol.read(os.path.join(PKG, "configs", pol_cfg))
    params = ebc_config.params_from_config(cfg, pol)
    sc = ebc_scene.SceneConfig.from_config(cfg)
    base = 2000 if "metric" in env_cfg else 1000
    start, count = shard.weak_range(envs, rank)  # weak scaling: every rank adds its own slice
    scenes = [ebc_scene.generate_scene(sc, s) for s in shard.scene_seeds(base, start, count)]
    return params, ebc_scene.SceneBatch.from_scenes(scenes)


def scene_config(workload):
    from ebcsim import scene as ebc_scene
    cfg = configparser.RawConfigParser()
    cfg.read(os.path.join(PKG, "configs", WORKLOADS[workload][0]))
    return ebc_scene.SceneConfig.from_config(cfg)


def host_cores(cap=16):
    """Threads for the all-cores CPU baseline: this process's CPU share — its affinity mask, cut
    to the cgroup quota when one is set, and to `cap` (the CPU share of a one-GPU box: a mask of
    256 logical CPUs there does not mean 256 cores are this job's)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    return max(1, min(n, cap))


def cpu_baseline(params, batch, seconds_target=7.0):
    """The CPU oracle (scalar C restatement) on a bounded sample of the same workload: one
    thread, then the env loop spread over every core this process may use (OpenMP static split;
    envs are independent).  `value` is the all-cores rate, `cores` the threads it used."""
    from ebcsim import _abi, scene as ebc_scene
    from oracle import oracle
    cores = host_cores()
    flags = _abi.FLAG_AUTO_RESET
    kw = dict(human_policy=_abi.HUMAN_ORCA, robot_policy=_abi.ROBOT_LINEAR, flags=flags)

    def sample(n, threads, chunk):
        sub = ebc_scene.SceneBatch(n, batch.N, batch.S, *[
            None if getattr(batch, k) is None else getattr(batch, k)[:n] for k in (
                "n_humans", "px", "py", "vx", "vy", "gx", "gy", "radius", "v_pref", "type",
                "n_static", "spx", "spy", "sradius", "grid", "robot")])
        env = oracle.OracleEnv(params, n, batch.N, batch.S)
        env.reset(sub)
        oracle.set_threads(threads)
        try:
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < 1.0:  # thread team, page faults, clocks
                env.step(**kw)
            steps, t0 = 0, time.perf_counter()
            while True:
                for _ in range(chunk):
                    env.step(**kw)
                steps += chunk
                dt = time.perf_counter() - t0
                if dt >= seconds_target or steps >= 100000:
                    break
        finally:
            oracle.set_threads(1)
        return int(sub.n_humans.sum()) * steps / dt, steps, dt

    n1 = min(256, batch.n)
    v1, s1, t1 = sample(n1, 1, 10)
    out = {"value": v1, "unit": "agent-steps/s", "cores": 1, "kind": "port",
           "sample": "%d envs x %d humans x %d steps of the same workload, oracle/ebc_oracle.c, "
                     "1 thread, %.1f s" % (n1, batch.N, s1, t1)}
    if cores > 1:
        nc = batch.n  # the whole batch: every thread gets a contiguous env slice
        vc, sc, tc = sample(nc, cores, 2)
        out = {"value": vc, "unit": "agent-steps/s", "cores": cores, "kind": "port",
               "sample": "%d envs x %d humans x %d steps of the same workload, oracle/ebc_oracle.c, "
                         "%d OpenMP threads over envs, %.1f s" % (nc, batch.N, sc, cores, tc),
               "single_thread": {"value": v1, "sample": out["sample"]}}
    out["kind_note"] = ("port = oracle/ebc_oracle.c, this repo's scalar C restatement of the reference's arithmetic "
                        "(parity-checked against the imported reference), timed on this box in this run")
    out["reference_python"] = REFERENCE_PYTHON
    return out


def launch_ranks(n, argv, extra_env=None, timeout=1800.0):
    """Start `n` fresh rank processes (one per GPU) running `argv`, with the torch.distributed.run
    environment (RANK, LOCAL_RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT); relay rank 0's stdout.
    The reference scales out the same way — a pool of worker processes (rl/train.py:19,
    rl/utils/parallel_explorer.py:275-276).  The parent never touches a GPU and never exec()s.
    All ranks are polled together: the first rank that exits non-zero (a HIP error, a reported
    mailbox fault, OOM) ends the job — the survivors, which would sit in a barrier or an all-reduce
    waiting for it, are killed and that rank's code is returned.  `timeout` seconds for the whole
    job (exit code 124).  Returns (exit code, rank 0's stdout)."""
    import socket
    import subprocess
    import threading
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        env.update(extra_env or {})
        procs.append(subprocess.Popen(argv, env=env, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    chunks = []
    drain = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    drain.start()  # rank 0's pipe is emptied while every rank is watched: a full pipe cannot stall it
    rc, deadline = 0, (time.monotonic() + timeout) if timeout else None
    try:
        pending = list(procs)
        while pending and rc == 0:
            for p in list(pending):
                code = p.poll()
                if code is not None:
                    pending.remove(p)
                    rc = rc or code
            if pending and rc == 0:
                if deadline is not None and time.monotonic() > deadline:
                    rc = 124
                else:
                    time.sleep(0.05)
    finally:
        for p in procs:  # exact PIDs we started, nothing by pattern
            if p.poll() is None:
                p.kill()
        for p in procs:
            p.wait()
        drain.join(timeout=10)
    return rc, b"".join(chunks).decode()


# the reference's Python env.step, measured in the survey container.  Carried as
# recorded constants (the reference never runs on the GPU box); every row has the humans on the
# reference's `linear` policy because rvo2 (ORCA) is not installable there — the true reference step is slower.
REFERENCE_PYTHON = {
    "hardware": "8 vCPU Intel Xeon @ 2.10 GHz, one process / one core; Python 3.10.12, numpy 2.2.6",
    "caveat": "humans on the reference's `linear` policy, not ORCA (rvo2 absent): ORCA cost excluded",
    "source": "BASELINE.md section 2 (simulator/env.py:388-466, best of 3 x 500 steps)",
    "rows": [
        {"case": "5 humans, 0 obstacles, local map off", "us_per_env_step": 181.0, "agent_steps_per_s": 27.6e3},
        {"case": "10 humans, 0 obstacles, local map off", "us_per_env_step": 349.0, "agent_steps_per_s": 28.7e3},
        {"case": "10 humans + 4 walls (6 static rows), local map off", "us_per_env_step": 358.6,
         "agent_steps_per_s": 27.9e3},
        {"case": "10 humans + 4 walls, local map on", "us_per_env_step": 4402.7, "agent_steps_per_s": 2.27e3},
        {"case": "robot decision + step, SARL baseline weights, 5 humans, 81-action look-ahead",
         "ms_per_decision": 117.0},
    ],
    "scale_out": "8 worker processes, one episode each (rl/train.py:19, rl/utils/parallel_explorer.py:275)",
}

MFMA_BF16_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA peak


def ebc_scene_slice(batch, n):
    """The first n scenes of a SceneBatch."""
    return ebc_scene_slice_range(batch, 0, n)


def ebc_scene_slice_range(batch, lo, hi):
    """Scenes [lo, hi) of a SceneBatch."""
    from ebcsim import scene as ebc_scene
    return ebc_scene.SceneBatch(hi - lo, batch.N, batch.S, *[
        None if getattr(batch, k) is None else getattr(batch, k)[lo:hi] for k in (
            "n_humans", "px", "py", "vx", "vy", "gx", "gy", "radius", "v_pref", "type",
            "n_static", "spx", "spy", "sradius", "grid", "robot")])


def also_kernels(env, batch, dev, cfg_s=None):
    """The two other kernels of the path with a roofline of their own, measured live (HIP events on
    the launch stream; inputs resident in HBM): the 81-action look-ahead sweep whose rotated rows
    are the one HBM-bound output of the path, and the value network's first block, which consumes
    those rows on the bf16 matrix cores.  Reported beside the headline, never part of `value`."""
    import torch
    from ebcsim import _abi, actions as ebc_actions
    from ebcsim.sarl import _NativeMlp2
    out = []
    ev = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731

    def timed(fn, n):
        e0, e1 = ev(), ev()
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n

    try:
        space = ebc_actions.build_action_space(float(batch.robot[0, 7]))
        A = len(space)
        acts = torch.tensor(space, dtype=torch.float64, device=dev)
        bufs = env.alloc_lookahead_outputs(A, ("reward", "done", "info", "rows_rotated"))
        env.lookahead_device(acts, bufs, human_policy=_abi.HUMAN_ORCA)  # ORCA prelude -> cached velocities
        sweep = lambda: env.lookahead_device(acts, bufs, human_policy=_abi.HUMAN_CACHED)  # noqa: E731
        timed(sweep, 3)
        ms = timed(sweep, 20)
        # algorithmic bytes per env: the pre-step rows in (as the step), A x R x T floats + A x 10 B out
        nbytes = env.E * (32 * batch.N + 44 + 16 * batch.S + 24 + A * (4 * env.T * env.R + 10))
        gbs = nbytes / (ms * 1e-3) / 1e9
        out.append({"kernel": "lookahead_kernel: %d envs x %d actions, rotated rows [E][A][R][T] left in HBM "
                              "(one launch, human velocities cached)" % (env.E, A),
                    "launch_ms": ms,
                    "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": gbs / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": nbytes}})
        del bufs
    except Exception as e:  # the headline line must still print
        out.append({"kernel": "lookahead_kernel", "error": repr(e)})
    try:
        # K steps per call with the robot on ORCA (ebc_step_k): the imitation-learning rollouts of rl/train.py:124-133
        # — per step the policy's state, the robot's ORCA action and the step, no host work in between
        K = 50
        ko = env.alloc_step_k_outputs(K, ("state_rotated", "reward", "done", "info"))
        roll = lambda: env.step_k_device(ko, K, human_policy=_abi.HUMAN_ORCA, robot_policy=_abi.ROBOT_ORCA,  # noqa: E731
                                         flags=_abi.FLAG_AUTO_RESET, robot_safety_space=0.15)
        timed(roll, 1)
        ms = timed(roll, 4)
        out.append({"kernel": "ebc_step_k: %d envs x %d steps per call, robot on ORCA (observe + orca_robot_kernel + "
                              "orca_step_kernel per step), outputs [K][E] left in HBM" % (env.E, K),
                    "call_ms": ms, "us_per_step": ms * 1e3 / K,
                    "env_steps_per_s": env.E * K / (ms * 1e-3), "agent_steps_per_s": float(batch.n_humans.sum()) * K / (ms * 1e-3)})
        del ko
    except Exception as e:
        out.append({"kernel": "ebc_step_k", "error": repr(e)})
    try:
        # env.reset of the whole batch from scenes generated on the device (scene_gen_kernel: numpy's MT19937 stream
        # and the reference's rejection loops, one lane per scene) beside the host generator doing the same scenes
        import time as _t
        from ebcsim import scene as ebc_scene
        if cfg_s is not None:
            from ebcsim.batched import BatchedEnv as _BE
            gen = ebc_scene.gen_struct(cfg_s, "test")
            genv = _BE(env.params, env.E, sum(gen.count), ebc_scene.max_static_rows(cfg_s), device=dev.index or 0)
            genv.generate_reset(gen, 1000)  # warm: first launch, allocations
            t0 = _t.perf_counter()
            for r in range(3):
                genv.generate_reset(gen, 1000 + r * env.E)
            genv.synchronize()
            dev_ms = (_t.perf_counter() - t0) / 3 * 1e3
            n_host = min(env.E, 256)
            t0 = _t.perf_counter()
            for s_ in range(n_host):
                ebc_scene.generate_scene(cfg_s, 1000 + s_, "test")
            host_ms = (_t.perf_counter() - t0) * 1e3 * env.E / n_host
            out.append({"kernel": "scene_gen_kernel: env.reset of %d envs from scenes generated on the device "
                                  "(ebc_generate_reset, seeds only over PCIe)" % env.E,
                        "call_ms": dev_ms, "scenes_per_s": env.E / (dev_ms * 1e-3),
                        "host_generator_ms_same_scenes": host_ms, "host_sample": "%d scenes, scaled" % n_host})
            del genv
    except Exception as e:
        out.append({"kernel": "scene_gen_kernel", "error": repr(e)})
    try:
        # The same batch as TWO independent sub-batches, each a handle on its own HIP stream (scenes are independent:
        # the split is exact): launches of the two overlap, one's tail under the other's head.  Reported beside the
        # headline, which stays one launch per step of the whole batch (per-launch roofline comparable across rounds).
        from ebcsim.batched import BatchedEnv as _BE
        half = env.E // 2
        subs, souts = [], []
        for lo in (0, half):
            e2 = _BE(env.params, half, batch.N, batch.S, device=dev.index or 0)  # keeps its private stream
            e2.reset(ebc_scene_slice_range(batch, lo, lo + half))
            subs.append(e2)
            souts.append(e2.alloc_step_outputs(("reward", "done", "info", "obs_rotated")))
        kw2 = dict(human_policy=_abi.HUMAN_ORCA, robot_policy=_abi.ROBOT_LINEAR, flags=_abi.FLAG_AUTO_RESET)

        def run(n):
            for _ in range(n):
                for e2, o2 in zip(subs, souts):
                    e2.step_device(o2, **kw2)
            for e2 in subs:
                e2.synchronize()
        run(50)
        import time as _t
        t0 = _t.perf_counter()
        run(400)
        dt = _t.perf_counter() - t0
        out.append({"kernel": "two sub-batches of %d envs on two HIP streams, one orca_step_kernel launch each per step" % half,
                    "us_per_step": dt / 400 * 1e6, "agent_steps_per_s": float(batch.n_humans[:2 * half].sum()) * 400 / dt})
        del subs, souts
    except Exception as e:
        out.append({"kernel": "two sub-batches on two streams", "error": repr(e)})
    try:
        # One robot decision per env for 1024 envs: 81-action look-ahead sweep + the SARL value network (the
        # architecture of the reference's shipped eb-cadrl weights, data/eb-cadrl/policy_x2_agent_type.config;
        # random-init weights) on 1024 x 81 pairs x R rows + float32 refinement of the near-best candidates + argmax
        # (rl/policy/multi_human_rl.py:38-80, rl/policy/sarl.py:38-82)
        from ebcsim import _capi
        from ebcsim.batched import BatchedEnv
        from ebcsim.sarl import DeviceSarlPolicy, SarlValueNet
        from ebcsim.train import SarlModule
        Ed = min(1024, env.E)
        sub = ebc_scene_slice(batch, Ed)
        denv = BatchedEnv(env.params, Ed, batch.N, batch.S, device=dev.index or 0)
        denv.reset(sub)
        denv.use_torch_stream()
        torch.manual_seed(0)
        mod = SarlModule(env.T, [300, 200], [200, 100], [300, 200, 200, 1], [200, 200, 1])
        net = SarlValueNet({k: v.detach() for k, v in mod.state_dict().items()}, device=str(dev))
        space = ebc_actions.build_action_space(float(batch.robot[0, 7]))
        pol = DeviceSarlPolicy(net, space, 0.9)
        dec = lambda: pol.decide(denv)  # noqa: E731
        timed(dec, 12)  # until the caching allocator has its GB-sized activation blocks and the clocks have settled
        ms = timed(dec, 8)
        A, R = len(space), denv.R
        row_macs = env.T * 300 + 300 * 200 + 200 * 200 + 200 * 100 + 200 * 200 + 200 * 200 + 200
        pair_macs = 200 * 200 + (6 + 100) * 300 + 300 * 200 + 200 * 200 + 200
        macs = float(Ed) * A * (R * row_macs + pair_macs)
        tf = 3.0 * 2.0 * macs / (ms * 1e-3) / 1e12  # three bf16 MFMA products per float32 product
        busy = None  # matrix-pipe occupancy per block, PMC-measured (tools/collect_value_net_pmc.sh), for THESE kernel sources only
        try:
            bj = json.load(open(os.path.join(ROOT, "profiles", "value_net_mfma_busy.json")))
            busy = ({k.replace("ebc::", ""): v["mfma_busy"] for k, v in bj["kernels"].items()} if bj.get("csrc_sha256") == _capi.csrc_sha256()
                    else "profiles/value_net_mfma_busy.json was taken on other kernel sources: not reported")
        except (OSError, ValueError, KeyError):
            pass
        out.append({"kernel": "decision: %d envs x %d actions x %d rows, look-ahead sweep + SARL x2 network (split-bf16 MFMA "
                              "blocks with the pair mean / attention sum in their epilogues) + float32 re-evaluation (ebc_mlp2_forward_f32) of every candidate within the blocks' error bound of the best + argmax" % (Ed, A, R),
                    "ms_per_decision_batch": ms, "decisions_per_s": Ed / (ms * 1e-3),
                    "native_blocks": bool(net._native_blocks()), "refine_stats": getattr(net, "refine_stats", None),
                    "mfma_busy_per_block": busy,
                    "roofline": {"bound": "mfma", "achieved": tf, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                                 "frac": tf / MFMA_BF16_PEAK_TFLOPS, "f32_equivalent_tflops": tf / 3.0,
                                 "algorithmic_macs_per_batch": macs}})
        del pol, net, denv
    except Exception as e:
        out.append({"kernel": "decision", "error": repr(e)})
    try:
        K0, H, O = env.T, 300, 200  # mlp1 of the reference's shipped eb-cadrl weights (data/eb-cadrl/rl_model_val.pth)
        M = 1024 * 81 * env.R
        g = torch.Generator(device="cpu").manual_seed(0)
        w1 = torch.randn(H, K0, generator=g) / K0 ** 0.5
        w2 = torch.randn(O, H, generator=g) / H ** 0.5
        blk = _NativeMlp2([(w1, torch.randn(H, generator=g)), (w2, torch.randn(O, generator=g))], dev.index or 0)
        x = torch.randn(M, K0, device=dev)
        fwd = lambda: blk(x, True)  # noqa: E731
        timed(fwd, 2)
        ms = timed(fwd, 5)
        f32_flops = 2.0 * M * (K0 * H + H * O)
        tf = 3.0 * f32_flops / (ms * 1e-3) / 1e12  # three bf16 MFMA products per float32 product (split operands)
        out.append({"kernel": "mlp2_split_wg_kernel: %d -> %d -> %d on %d rows (1024 envs x 81 actions x %d rows), "
                              "float32 operands split into two bf16" % (K0, H, O, M, env.R),
                    "launch_ms": ms,
                    "roofline": {"bound": "mfma", "achieved": tf, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                                 "frac": tf / MFMA_BF16_PEAK_TFLOPS, "f32_equivalent_tflops": tf / 3.0}})
    except Exception as e:
        out.append({"kernel": "mlp2_split_wg_kernel", "error": repr(e)})
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="metric", choices=sorted(WORKLOADS))
    ap.add_argument("--envs", type=int, default=None, help="envs per GPU (default: workload's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--min-block-seconds", type=float, default=0.05,
                    help="a K-step block shorter than this is repeated after a clock warm-up; the median is reported")
    ap.add_argument("--no-also", action="store_true", help="skip the look-ahead / value-network side measurements")
    ap.add_argument("--human-policy", default="orca", choices=["orca", "linear"],
                    help="diagnostic only: the headline metric is ORCA")
    ap.add_argument("--rehearsal-gpu-ranks", type=int, default=0,
                    help="EBCSIM_BENCH_BACKEND=gloo only: rehearse an N-rank job on a box with fewer GPUs — ranks >= this "
                         "number build their env slice on the host and take part in every collective but never open the "
                         "GPU (a one-GPU box admits few processes on its card); the line then says timing: rehearsal")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # not under a launcher: be one.  This process has not initialised any GPU and starts children.
        rc, out = launch_ranks(args.gpus, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:])
        sys.stdout.write(out)
        sys.stdout.flush()
        sys.exit(rc)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    args.gpus = world

    import torch
    import torch.distributed as dist
    # one process per GPU; EBCSIM_BENCH_BACKEND=gloo rehearses the N > 1 path on a box with fewer
    # GPUs than ranks (ranks then share devices; timing is meaningless, the code path is not)
    backend = os.environ.get("EBCSIM_BENCH_BACKEND", "nccl")
    rehearsal = backend != "nccl"
    # a rehearsal rank past --rehearsal-gpu-ranks: its slice of the scenes on the host and every collective of
    # the job, no GPU work and no HIP context (it contributes its units and a time of zero)
    on_gpu = not (rehearsal and args.rehearsal_gpu_ranks > 0 and rank >= args.rehearsal_gpu_ranks)
    if on_gpu and not torch.cuda.is_available():
        sys.exit("bench.py needs a HIP device (no CPU fallback)")
    dev = None
    if on_gpu:
        local_rank = local_rank % torch.cuda.device_count() if rehearsal else local_rank
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    gpu_sync = torch.cuda.synchronize if on_gpu else (lambda: None)
    # EBCSIM_FORCE_COLLECTIVES=1: a one-rank job still forms its process group and runs the barriers and the
    # reductions through the backend (the RCCL path on a one-GPU box: tests/test_bench_launcher.py)
    collective = world > 1 or os.environ.get("EBCSIM_FORCE_COLLECTIVES") == "1"
    if collective:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from ebcsim import _abi
    from ebcsim.batched import BatchedEnv

    E = args.envs or WORKLOADS[args.workload][2]
    params, batch = build_batch(args.workload, E, rank)
    flags = _abi.FLAG_AUTO_RESET
    hp = _abi.HUMAN_ORCA if args.human_policy == "orca" else _abi.HUMAN_LINEAR
    kw = dict(human_policy=hp, robot_policy=_abi.ROBOT_LINEAR, flags=flags)
    if on_gpu:
        env = BatchedEnv(params, E, batch.N, batch.S, device=local_rank)
        env.reset(batch)
        env.use_torch_stream()
        outs = env.alloc_step_outputs(("reward", "done", "info", "obs_rotated"))
        step = lambda: env.step_device(outs, **kw)  # noqa: E731
    else:
        env = None
        step = lambda: None  # noqa: E731

    def group_barrier():
        # a rehearsal's barrier is an all-reduce of a CPU scalar: dist.barrier() on a gloo group picks a device for
        # itself and initialises HIP in ranks that were to stay off the card (seen: the box's process guard counted them)
        if rehearsal:
            dist.all_reduce(torch.zeros(1))
        else:
            dist.barrier()

    def barrier():
        gpu_sync()
        if collective:
            group_barrier()
        gpu_sync()

    for _ in range(args.warmup):
        step()
    barrier()

    def block():
        """EXACTLY K steps between barrier + synchronize brackets -> (this rank's seconds, stream ms)."""
        if not on_gpu:
            barrier()
            return 0.0, 0.0
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record()
        for _ in range(args.steps):
            step()
        ev1.record()
        torch.cuda.synchronize()             # this rank's K steps are done ...
        dt = time.perf_counter() - t0        # ... its time; the job's time is the MAX over ranks (below)
        barrier()                            # closing bracket: every rank is done
        return dt, ev0.elapsed_time(ev1)     # HIP events on the stream the kernels run on

    first = block()
    blocks = [first]
    n_blocks = 1
    from ebcsim import shard as _sh
    # every rank takes the same decision: the first block's time is max-reduced BEFORE it is compared
    first_max, _ = _sh.job_rate(first[0], 0.0, device=dev if backend == "nccl" else None)
    if first_max < args.min_block_seconds:
        # A K-step block shorter than ~50 ms is over before the chip has left its idle clock (round 1:
        # 20 steps = 0.43 ms read 17 % low).  Keep stepping untimed for >= 0.3 s, then repeat the same
        # bracketed K-step block an odd number of times and report the MEDIAN block: `steps` and
        # `ms_per_step x steps` still describe one block.
        n_blocks = int(min(101, max(5, 2 * int(0.5 * args.min_block_seconds / max(first_max, 1e-6)) + 1)))
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.3:
            for _ in range(50):
                step()
            gpu_sync()
        barrier()
        blocks = [block() for _ in range(n_blocks)]
    # max over ranks per block, then the median block
    bt = torch.tensor([b[0] for b in blocks], dtype=torch.float64, device=dev if backend == "nccl" else None)
    if collective:
        dist.all_reduce(bt, op=dist.ReduceOp.MAX)
    order = sorted(range(len(blocks)), key=lambda i: float(bt[i]))
    mid = order[len(order) // 2]
    elapsed, stream_ms = float(bt[mid]), blocks[mid][1]
    block_spread = (float(bt.min()), float(bt.max()))

    # per-launch kernel duration: HIP events around every launch, in a separate untimed pass
    kernel_ms = None
    if on_gpu:
        env.timing(True)
        for _ in range(min(args.steps, 200)):
            step()
        kernel_ms, n_timed = env.timing_read(reset=True)
        env.timing(False)

    also = None
    if rank == 0 and world == 1 and not args.no_also and args.human_policy == "orca":
        also = also_kernels(env, batch, dev, scene_config(args.workload))

    from ebcsim import shard
    elapsed_max, total_humans = shard.job_rate(elapsed, float(batch.n_humans.sum()),
                                              device=dev if backend == "nccl" else None)
    gpu_open = None
    if rehearsal:  # how many rank processes hold the GPU's device files (the one-GPU box's process guard counts them)
        held = 0
        for f in os.listdir("/proc/self/fd"):
            try:
                held |= "kfd" in os.readlink("/proc/self/fd/" + f)
            except OSError:
                pass
        _, gpu_open = shard.job_rate(0.0, float(held), device=None)
    try:
        if on_gpu:
            env.synchronize()  # also reports a mailbox fault of any step above (EBC_ERR_DEVICE)
    except Exception as e:
        sys.exit("bench.py: the device reported a fault during the timed steps: %r" % (e,))

    if rank == 0:
        S_mean = float(batch.n_static.mean()) if batch.S else 0.0
        bytes_launch = algorithmic_bytes_per_env_step(batch.N, S_mean, env.T) * E
        # one launch per step: the launch's average duration over the timed region = HIP events
        # around the region / K (rocprofv3 --kernel-trace --stats of this command agrees, profiles/);
        # kernel_ms = events around every single launch in a separate pass (they add ~2 us each)
        launch_ms = stream_ms / args.steps
        achieved = bytes_launch / (launch_ms * 1e-3) / 1e9 if launch_ms > 0 else 0.0
        traffic = valu = traffic_note = None  # PMC-measured, from the committed profile of this workload (profiles/)
        tpath = os.path.join(ROOT, "profiles", "traffic_%s.json" % args.workload)
        if os.path.exists(tpath) and world == 1:
            tj = json.load(open(tpath))
            # PMC counters cannot be collected inside this run (rocprofv3 owns them): the file is a committed
            # measurement of THIS library on this workload.  It names the sources it was taken on; when the
            # kernels have changed since, the numbers are refused rather than reported stale.
            from ebcsim import _capi
            if tj.get("csrc_sha256") != _capi.csrc_sha256():
                traffic_note = "profiles/traffic_%s.json was taken on other kernel sources (%s...): not reported" % (
                    args.workload, str(tj.get("csrc_sha256"))[:12])
            elif tj.get("envs_per_gpu") == E:
                traffic = tj["traffic_bytes_per_step"]
                if tj.get("SQ_INSTS_VALU_per_step"):
                    # the honest yard-stick of this launch: vector-ALU issue time.  One
                    # wave64 VALU instruction holds its SIMD for 4 cycles; 256 CUs x 4 SIMDs.
                    clock = tj.get("shader_clock_hz", 2.4e9)
                    issue_s = tj["SQ_INSTS_VALU_per_step"] * 4.0 / (1024 * clock)
                    valu = {"issue_us": issue_s * 1e6, "frac_of_launch": issue_s / (launch_ms * 1e-3),
                            "SQ_INSTS_VALU_per_launch": tj["SQ_INSTS_VALU_per_step"], "shader_clock_hz": clock,
                            "source": tj.get("valu_source", tpath)}
        line = {
            "metric": "agent-steps/sec", "value": total_humans * args.steps / elapsed_max,
            "unit": "agent-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed_max / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "timed_blocks": n_blocks, "block_seconds_min_max": block_spread,
            # every env of every rank advances all its humans once per step
            "units_per_step": int(total_humans),
            # "measured": one rank per GPU over RCCL.  "rehearsal": a gloo job whose ranks share GPUs (or, past
            # --rehearsal-gpu-ranks, stay on the host): the N-rank code path ran, `value` is NOT a scaling number
            "timing": "rehearsal" if rehearsal else "measured",
            "ranks_on_gpu": (min(world, args.rehearsal_gpu_ranks) if args.rehearsal_gpu_ranks > 0 else world) if rehearsal else world,
            "ranks_with_gpu_open": None if gpu_open is None else int(gpu_open),
            "config": {"workload": "%s: %d envs/GPU x %d humans + %d static rows, ORCA + kinematics + "
                                   "collisions + reward + rotated obs (T=%d), one ebc_step (one HIP launch) per step, "
                                   "auto-reset%s" % (args.workload, E, batch.N, batch.S, env.T,
                                                   "" if hp == _abi.HUMAN_ORCA else " [DIAGNOSTIC: linear humans]"),
                       "envs_per_gpu": E, "humans": int(batch.N), "parallelism": "env-slice x%d" % world},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_ratio": None if traffic is None else traffic / bytes_launch, "valu": valu,
                         "traffic_note": traffic_note,
                         "kernel": "orca_step_kernel (the one launch of a step)",
                         "launch_ms": launch_ms, "kernel_ms_event_pair": kernel_ms,
                         "algorithmic_bytes_per_launch": bytes_launch},
        }
        if also is not None:
            line["also"] = also
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(params, batch)
        print(json.dumps(line), flush=True)
    if collective:
        group_barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
