#!/usr/bin/env python3
"""Per-wave timeline of the step kernels: when each one-wave workgroup started and ended, on the
device's constant 100 MHz clock, and how many shader cycles it lived.  Needs the trace build:

    make -C eb-cadrl_amd/csrc trace
    EBCSIM_LIB=eb-cadrl_amd/lib/libebcsim_trace.so python3 tools/wave_timeline.py [workload] [envs]

Prints, per kernel role: first/last start, last end (us from the first start of the launch),
wave lifetime percentiles, the shader clock the lifetimes imply, and how many waves were resident
over time.
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "eb-cadrl_amd")):
    sys.path.insert(0, p)

TAGS = {0: "orca_kernel", 1: "step_kernel", 2: "orca_step_kernel", 3: "(unused)"}


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
    fl = _abi.FLAG_AUTO_RESET
    for hp in (_abi.HUMAN_LINEAR, _abi.HUMAN_ORCA):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        for i in range(60):
            if i == 10:
                ev[0].record()
            env.step_device(outs, human_policy=hp, robot_policy=_abi.ROBOT_LINEAR, flags=fl)
        ev[1].record()
        torch.cuda.synchronize()
        # the launch-to-launch period of THIS build, to set beside the last wave's end below: the difference is
        # what the launch spends before its first wave and after its last one (dispatch, end-of-kernel write-back)
        print("%s: %.2f us per step over 50 back-to-back launches (events)" % (
            "linear humans" if hp == _abi.HUMAN_LINEAR else "ORCA humans", ev[0].elapsed_time(ev[1]) * 1e3 / 50))
    t = buf.cpu().numpy().astype(np.uint64)
    launch0 = None
    for tag in (2, 3, 1):
        rows = t[tag]
        idx = np.nonzero(rows[:, 1] != 0)[0]
        rows = rows[idx]
        if not len(rows):
            continue
        if tag == 2:  # orca_step_kernel: blocks in role order ENV, ORCA, ROWS, STATE
            env_blocks = -(-E // (64 // batch.N))
            state_blocks = env_blocks
            if os.environ.get("EBCSIM_STEP_FORM", "3") == "3":  # the default form: the ENV role with four lanes per env
                env_blocks = -(-E // 16)
            others = batch.N - 1 + (1 if params.robot_visible else 0)
            gs = next(g for g in (2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 16, 21, 32) if g >= others)
            orca_blocks = -(-E * batch.N // (64 // gs))
            R = batch.N + batch.S
            epw = 64 // R if R <= 64 else 1
            rows_blocks = -(-E // epw)
            b1, b2, b3 = env_blocks, env_blocks + orca_blocks, env_blocks + orca_blocks + rows_blocks
            parts = [("ENV", idx < b1), ("ORCA", (idx >= b1) & (idx < b2)),
                     ("ROWS", (idx >= b2) & (idx < b3)), ("STATE", idx >= b3)]
        else:
            parts = [(TAGS[tag], np.ones(len(idx), bool))]
        base_all = rows[:, 0].astype(np.int64).min()
        if tag == 2:
            launch0 = base_all
        for name, sel in parts:
            if sel.any():
                report(np, name, rows[sel], launch0 if tag == 3 and launch0 is not None else base_all)
        if tag == 2 and len(idx) == b3 + state_blocks and state_blocks == env_blocks:
            # what each STATE wave waited for: the ORCA waves of its envs and its ENV wave
            end = (rows[:, 1].astype(np.int64) - base_all) / 100.0
            start = (rows[:, 0].astype(np.int64) - base_all) / 100.0
            hpw, epb = 64 // gs, 64 // batch.N
            lag, info = [], []
            for sb in range(env_blocks):
                h0, h1 = sb * epb * batch.N, min((sb + 1) * epb, E) * batch.N - 1
                o0, o1 = b1 + h0 // hpw, b1 + h1 // hpw
                ready_orca = end[o0:o1 + 1].max()
                ready = max(ready_orca, end[sb])
                lag.append(end[b3 + sb] - ready)
                info.append((end[b3 + sb], ready_orca, end[sb], start[b3 + sb]))
            lag = np.array(lag)
            print("STATE end minus (its last ORCA wave's end, its ENV wave's end): p10/p50/p90/max %.2f %.2f %.2f %.2f us" % (
                np.percentile(lag, 10), np.percentile(lag, 50), np.percentile(lag, 90), lag.max()))
            last = np.argsort([q[0] for q in info])[-6:]
            print("   the 6 STATE waves that ended last (end, last ORCA end, ENV end, own start): " +
                  " ".join("(%.1f %.1f %.1f %.1f)" % info[q] for q in last))


def report(np, name, rows, base):
    r0, r1 = rows[:, 0].astype(np.int64), rows[:, 1].astype(np.int64)
    cyc = (rows[:, 3] - rows[:, 2]).astype(np.int64)
    us = lambda x: (x - base) / 100.0
    life = (r1 - r0) / 100.0
    mhz = np.median(cyc[life > 0] / life[life > 0])
    print("%s: %d waves; starts %.2f..%.2f us, last end %.2f us" % (
        name, len(rows), us(r0.min()), us(r0.max()), us(r1.max())))
    print("   lifetime us p10/p50/p90/max: %.2f %.2f %.2f %.2f; shader clock ~%.0f MHz" % (
        np.percentile(life, 10), np.percentile(life, 50), np.percentile(life, 90), life.max(), mhz))
    edges = np.arange(0, us(r1.max()) + 1.0, 1.0)
    print("   resident waves at t = 0, 1, 2 ... us: " + " ".join(str(int(((us(r0) <= x) & (us(r1) > x)).sum())) for x in edges))
    print("   started by t:                        " + " ".join(str(int((us(r0) <= x).sum())) for x in edges))
    if name == "ENV" and rows[:, 5].any():
        c0 = rows[:, 2].astype(np.int64)
        mk = rows[:, 5:10].astype(np.int64)
        names = ["start->loads ready", "robot action", "distances", "ordered reduce", "grid window", "reward + outputs"]
        segs = [mk[:, 0] - c0] + [mk[:, q + 1] - mk[:, q] for q in range(4)] + [cyc - (mk[:, 4] - c0)]
        print("   cycles mean: " + ", ".join("%s %.0f" % (n, v.mean()) for n, v in zip(names, segs)))
    elif name == "ENV" and rows[:, 6].any():  # four lanes per env: marks 1 (action), 2 (quarters combined), 3 (grid), 4 (end)
        c0 = rows[:, 2].astype(np.int64)
        mk = rows[:, 6:10].astype(np.int64) - c0[:, None]
        names = ["start->robot action done", "publish + humans + distances + combine", "grid window", "reward + outputs"]
        segs = [mk[:, 0]] + [mk[:, q + 1] - mk[:, q] for q in range(3)]
        print("   cycles mean: " + ", ".join("%s %.0f" % (n, v.mean()) for n, v in zip(names, segs)))
    if name == "ORCA":
        last = np.argsort(r1)[-8:]
        print("   the 8 waves that ended last (start us, lifetime us): " + " ".join("(%.1f, %.1f)" % (us(r0[q]), life[q]) for q in last))
    if name == "STATE" and rows[:, 5].any():
        c0 = rows[:, 2].astype(np.int64)
        mk = rows[:, 5:9].astype(np.int64)
        mhz = cyc / np.maximum(life, 1e-9)  # this wave's shader clock
        at = lambda q: us(r0) + (mk[:, q] - c0) / mhz  # a mark as us on the launch's time axis
        ready, woke, done_ = at(0), at(1), at(2)
        pct = lambda v: "%.1f / %.1f / %.1f" % (np.percentile(v, 10), np.percentile(v, 50), np.percentile(v, 90))
        print("   on the launch's time axis, p10 / p50 / p90 us: own loads ready %s, mailboxes all full %s, state stored %s" % (
            pct(ready), pct(woke), pct(done_)))
        print("   mailboxes full -> stored: %s us" % pct(done_ - woke))
        last = np.argsort(done_)[-6:]
        print("   the 6 that stored last (start, loads ready, mailboxes full, stored): " + " ".join(
            "(%.1f %.1f %.1f %.1f)" % (us(r0[q]), ready[q], woke[q], done_[q]) for q in last))
    if name == "ORCA" and rows[:, 5].any():
        c0 = rows[:, 2].astype(np.int64)
        marks = rows[:, 5:10].astype(np.int64) - c0[:, None]
        total = cyc
        names = ["loads", "rank", "lines", "LP2", "LP3", "end"]
        seg = np.concatenate([marks[:, :1], np.diff(marks, axis=1), (total - marks[:, 4])[:, None]], axis=1)
        lp1, lp3, lp3it = (rows[:, 11 + q].astype(np.int64) for q in range(3))
        order = np.argsort(life)
        for label, pick in (("median waves", order[len(order) // 2 - 50:len(order) // 2 + 50]),
                            ("slowest 1%", order[-max(1, len(order) // 100):])):
            print("   %s: cycles per segment " % label + ", ".join(
                "%s %.0f" % (n, seg[pick, q].mean()) for q, n in enumerate(names)) +
                "; LP1 calls %.1f, LP3 entered %.2f, LP3 rounds %.1f" % (lp1[pick].mean(), lp3[pick].mean(), lp3it[pick].mean()))
        prev = rows[:, 15].astype(np.int64)
        both = ((lp3 > 0) & (prev > 0)).sum()
        print("   waves entering LP3 this launch %d, the launch before %d, both %d (of %d waves)" % ((lp3 > 0).sum(), (prev > 0).sum(), both, len(lp3)))
        print("   LP1 calls per wave: mean %.2f p50 %d p90 %d max %d; waves entering LP3: %.1f%%" % (
            lp1.mean(), np.percentile(lp1, 50), np.percentile(lp1, 90), lp1.max(), 100.0 * (lp3 > 0).mean()))


if __name__ == "__main__":
    main()
