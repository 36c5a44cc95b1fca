#!/usr/bin/env python3
"""Top rows of a rocprofv3 --kernel-trace --stats summary: python3 tools/kernel_stats_top.py <dir or csv> [n]"""
import csv
import glob
import os
import sys

path = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
if os.path.isdir(path):
    path = glob.glob(os.path.join(path, "**", "*kernel_stats.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(path)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:n]:
    print("%-74s %5s %9.1f us avg %6.2f%%  max %.0f" % (r["Name"][:74], r["Calls"], float(r["AverageNs"]) / 1e3,
                                                       100 * float(r["TotalDurationNs"]) / tot, float(r["MaxNs"]) / 1e3))
print("total ms", tot / 1e6)
