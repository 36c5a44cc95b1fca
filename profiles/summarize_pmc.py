#!/usr/bin/env python3
"""Aggregate a rocprofv3 --pmc counter_collection.csv into per-kernel means (one JSON object).

    python profiles/summarize_pmc.py gpurun_out/prof_pmc1/*/*counter_collection.csv [more.csv] > out.json
"""
import collections
import csv
import json
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(list))
meta = {}
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "")
        name = name[5:] if name.startswith("void ") else name
        name = name.split("(")[0]
        agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
        meta[name] = {k: r[k] for k in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size",
                                        "Scratch_Size", "VGPR_Count", "SGPR_Count")}
out = {}
for name, counters in agg.items():
    out[name] = {"dispatch": meta[name],
                 "mean_per_dispatch": {c: sum(v) / len(v) for c, v in sorted(counters.items())},
                 "dispatches": max(len(v) for v in counters.values())}
json.dump(out, sys.stdout, indent=1)
print()
