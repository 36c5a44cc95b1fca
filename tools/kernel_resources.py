#!/usr/bin/env python3
"""Table of per-kernel resources from `hipcc -Rpass-analysis=kernel-resource-usage` output (stderr file):
name (demangled template arguments), SGPRs, VGPRs, AGPRs, scratch bytes/lane, spills, waves/SIMD, LDS."""
import re
import subprocess
import sys


def main(path, pattern=""):
    rows, cur = [], None
    for line in open(path):
        m = re.search(r"remark:\s+(.*?)(?: \[-Rpass)", line)
        if not m:
            continue
        t = m.group(1).strip()
        if t.startswith("Function Name:"):
            cur = {"name": t.split(":", 1)[1].strip()}
            rows.append(cur)
        elif cur is not None and ":" in t:
            k, v = t.split(":", 1)
            cur[k.strip()] = v.strip()
    if not rows:
        print("no kernels in %s" % path)
        return
    names = subprocess.run(["c++filt"] + [r["name"] for r in rows], capture_output=True, text=True,
                           stdin=subprocess.DEVNULL).stdout.splitlines()
    print("%-64s %5s %5s %5s %7s %6s %5s %7s" % ("kernel", "SGPR", "VGPR", "AGPR", "scratch", "spillV", "occ", "LDS"))
    for r, n in zip(rows, names):
        n = re.sub(r"\(.*", "", n).replace("ebc::", "").replace("void ", "")
        if pattern and pattern not in n:
            continue
        print("%-64s %5s %5s %5s %7s %6s %5s %7s" % (n[:64], r.get("TotalSGPRs"), r.get("VGPRs"), r.get("AGPRs"),
                                                    r.get("ScratchSize [bytes/lane]"), r.get("VGPRs Spill"),
                                                    r.get("Occupancy [waves/SIMD]"), r.get("LDS Size [bytes/block]")))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
