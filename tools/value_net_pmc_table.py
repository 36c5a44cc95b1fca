#!/usr/bin/env python3
"""Per value-network kernel: launch time and the matrix pipe's share of it, from tools/collect_value_net_pmc.sh's
passes.  busy = SQ_VALU_MFMA_BUSY_CYCLES / (SIMDs x GRBM_GUI_ACTIVE): the counter sums, over the chip's 1024 SIMDs,
the cycles a matrix instruction occupies the pipe (MI355X_MICROARCH.md: cycles, 32 per 32x32x16 bf16 instruction).

    python3 tools/value_net_pmc_table.py pmc.json <kernel-trace dir> [busy.json]

busy.json: {kernel: {"us": launch time, "mfma_busy": fraction}} for the value-network kernels, stamped with the hash of
the kernel sources it was measured on (bench.py reports it in also[] only for those sources).
"""
import csv
import glob
import json
import os
import sys

pmc = json.load(open(sys.argv[1]))
stats = {}
for path in glob.glob(os.path.join(sys.argv[2], "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        name = r["Name"]
        name = name[5:] if name.startswith("void ") else name
        stats[name.split("(")[0]] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3)
SIMDS = 256 * 4
XCDS = 8
busy_out = {}
print("%-52s %5s %9s %7s %9s %9s %8s %8s" % ("kernel", "calls", "us", "mfma%", "mfma/wave", "valu/wave", "lds/wave", "wait%"))
for name, rec in sorted(pmc.items(), key=lambda kv: -stats.get(kv[0], (0, 0))[0] * stats.get(kv[0], (0, 0))[1]):
    m = rec["mean_per_dispatch"]
    if "SQ_VALU_MFMA_BUSY_CYCLES" not in m or name not in stats:
        continue
    calls, us = stats[name]
    gui = m.get("GRBM_GUI_ACTIVE", 0.0)
    waves = max(m.get("SQ_WAVES", 1.0), 1.0)
    busy = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (SIMDS * gui / XCDS) if gui else float("nan")  # (the counter sums the 8 XCDs' clocks)
    wait = m.get("SQ_WAIT_INST_ANY", 0.0) / max(m.get("SQ_WAVE_CYCLES", 1.0), 1.0)
    if name.startswith("ebc::mlp2_"):
        busy_out[name] = {"calls": calls, "us": round(us, 1), "mfma_busy": round(busy, 4),
                          "mfma_per_wave": round(m.get("SQ_INSTS_MFMA", 0) / waves), "valu_per_wave": round(m.get("SQ_INSTS_VALU", 0) / waves)}
    print("%-52s %5d %9.1f %6.1f%% %9.0f %9.0f %8.0f %7.1f%%" % (name[:52], calls, us, 100 * busy, m.get("SQ_INSTS_MFMA", 0) / waves,
                                                             m.get("SQ_INSTS_VALU", 0) / waves, m.get("SQ_INSTS_LDS", 0) / waves,
                                                             100 * wait))

if len(sys.argv) > 3:
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "eb-cadrl_amd"))
    from ebcsim import _capi  # noqa: E402
    json.dump({"csrc_sha256": _capi.csrc_sha256(), "workload": "tools/sarl_profile.py 1024 (shipped eb-cadrl weights, 0.52 M rows per chunk)",
               "definition": "SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs), rocprofv3 --pmc, counters alone",
               "kernels": busy_out}, open(sys.argv[3], "w"), indent=1)
