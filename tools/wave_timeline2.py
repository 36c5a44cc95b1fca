#!/usr/bin/env python3
"""Per-wave timeline of the SECOND form of the fused step (orca_step2_kernel: ENV1, ORCA, ENV2), trace build:

    make -C eb-cadrl_amd/csrc dev DEV_FLAGS="-DEBC_WAVE_TRACE=2" DEV_OUT=libebcsim_devtrace.so
    EBCSIM_LIB=eb-cadrl_amd/lib/libebcsim_devtrace.so python3 tools/wave_timeline2.py [workload] [envs]
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "eb-cadrl_amd")):
    sys.path.insert(0, p)


def main():
    import numpy as np
    import torch
    import bench
    from ebcsim import _abi, _capi
    from ebcsim.batched import BatchedEnv
    workload = sys.argv[1] if len(sys.argv) > 1 else "metric"
    E = int(sys.argv[2]) if len(sys.argv) > 2 else bench.WORKLOADS[workload][2]
    L = _capi.lib()
    L.ebc_debug_wave_trace.restype = C.c_int
    L.ebc_debug_wave_trace.argtypes = [C.c_void_p, C.c_uint]
    params, batch = bench.build_batch(workload, E, 0)
    env = BatchedEnv(params, E, batch.N, batch.S)
    env.reset(batch)
    env.use_torch_stream()
    blocks = 1 << 17
    buf = torch.zeros((4, blocks, 16), dtype=torch.int64, device="cuda")
    _capi.check(L.ebc_debug_wave_trace(buf.data_ptr(), blocks))
    outs = env.alloc_step_outputs(("reward", "done", "info", "obs_rotated"))
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for i in range(60):
        if i == 10:
            ev[0].record()
        env.step_device(outs, human_policy=_abi.HUMAN_ORCA, robot_policy=_abi.ROBOT_LINEAR, flags=_abi.FLAG_AUTO_RESET)
    ev[1].record()
    torch.cuda.synchronize()
    print("ORCA humans, second form: %.2f us per step over 50 back-to-back launches (events)" % (ev[0].elapsed_time(ev[1]) * 1e3 / 50))
    t = buf.cpu().numpy().astype(np.uint64)
    rows = t[2]
    idx = np.nonzero(rows[:, 1] != 0)[0]
    rows = rows[idx]
    others = batch.N - 1 + (1 if params.robot_visible else 0)
    gs = next(g for g in (2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 16, 21, 32) if g >= others)
    b1 = -(-E // 64)
    b2 = b1 + -(-E * batch.N // (64 // gs))
    base = rows[:, 0].astype(np.int64).min()
    for name, sel in (("ENV1", idx < b1), ("ORCA", (idx >= b1) & (idx < b2)), ("ENV2", idx >= b2)):
        if sel.any():
            report(np, name, rows[sel], base)


def report(np, name, rows, base):
    r0, r1 = rows[:, 0].astype(np.int64), rows[:, 1].astype(np.int64)
    cyc = (rows[:, 3] - rows[:, 2]).astype(np.int64)
    us = lambda x: (x - base) / 100.0  # noqa: E731
    life = (r1 - r0) / 100.0
    mhz = np.median(cyc[life > 0] / life[life > 0])
    print("%s: %d waves; starts %.2f..%.2f us, ends %.2f..%.2f us" % (name, len(rows), us(r0.min()), us(r0.max()), us(r1.min()), us(r1.max())))
    print("   lifetime us p10/p50/p90/max: %.2f %.2f %.2f %.2f; shader clock ~%.0f MHz" % (
        np.percentile(life, 10), np.percentile(life, 50), np.percentile(life, 90), life.max(), mhz))
    edges = np.arange(0, us(r1.max()) + 1.0, 1.0)
    print("   resident waves at t = 0, 1, 2 ... us: " + " ".join(str(int(((us(r0) <= x) & (us(r1) > x)).sum())) for x in edges))
    print("   started by t:                        " + " ".join(str(int((us(r0) <= x).sum())) for x in edges))
    c0 = rows[:, 2].astype(np.int64)
    if name == "ORCA" and rows[:, 5].any():
        marks = rows[:, 5:11].astype(np.int64) - c0[:, None]
        names = ["loads+stage", "rank", "park+lines", "LP2", "LP3", "commit stores", "frame+row+flag"]
        seg = np.concatenate([marks[:, :1], np.diff(marks, axis=1), (cyc - marks[:, 5])[:, None]], axis=1)
        order = np.argsort(life)
        for label, pick in (("median waves", order[len(order) // 2 - 50:len(order) // 2 + 50]),
                            ("fastest 10%", order[:len(order) // 10]), ("slowest 1%", order[-max(1, len(order) // 100):])):
            print("   %s: cycles per segment " % label + ", ".join("%s %.0f" % (n, seg[pick, q].mean()) for q, n in enumerate(names)))
        last = np.argsort(r1)[-8:]
        print("   the 8 waves that ended last (start us, lifetime us): " + " ".join("(%.1f, %.1f)" % (us(r0[q]), life[q]) for q in last))
    if name == "ENV2" and rows[:, 6].any():
        mk = rows[:, 6:10].astype(np.int64) - c0[:, None]  # marks 1..4
        names = ["start->action in hand", "distances+reduce", "leader tail", "static rows", "rest (restart)"]
        seg = np.concatenate([mk[:, :1], np.diff(mk, axis=1), (cyc - mk[:, 3])[:, None]], axis=1)
        print("   cycles mean: " + ", ".join("%s %.0f" % (n, seg[:, q].mean()) for q, n in enumerate(names)))
        print("   cycles p90:  " + ", ".join("%s %.0f" % (n, np.percentile(seg[:, q], 90)) for q, n in enumerate(names)))


if __name__ == "__main__":
    main()
