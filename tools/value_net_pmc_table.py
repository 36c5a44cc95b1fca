#!/usr/bin/env python3
"""Per value-network kernel: launch time and the matrix pipe's share of it, from tools/collect_value_net_pmc.sh's
passes.  busy = SQ_VALU_MFMA_BUSY_CYCLES / (SIMDs x GRBM_GUI_ACTIVE): the counter sums, over the chip's 1024 SIMDs,
the cycles a matrix instruction occupies the pipe (MI355X_MICROARCH.md: cycles, 32 per 32x32x16 bf16 instruction).

    python3 tools/value_net_pmc_table.py pmc.json <kernel-trace dir>
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
    print("%-52s %5d %9.1f %6.1f%% %9.0f %9.0f %8.0f %7.1f%%" % (name[:52], calls, us, 100 * busy, m.get("SQ_INSTS_MFMA", 0) / waves,
                                                             m.get("SQ_INSTS_VALU", 0) / waves, m.get("SQ_INSTS_LDS", 0) / waves,
                                                             100 * wait))
