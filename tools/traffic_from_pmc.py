#!/usr/bin/env python3
"""HBM traffic per step of the bench workload from a PMC summary (profiles/summarize_pmc.py output).

    python3 tools/traffic_from_pmc.py <pmc.json> <workload> <envs_per_gpu> <humans> <source note> > profiles/traffic_<workload>.json

MI355X_MICROARCH.md, HBM section: FETCH_SIZE and WRITE_SIZE are in KB; on gfx950 FETCH_SIZE counts
64 B per 128-B request, so the read side is doubled; the two counters come from separate --pmc
passes (tools/collect_profiles.sh).  bench.py reads traffic_bytes_per_step into roofline.traffic."""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "eb-cadrl_amd"))
from ebcsim import _capi  # noqa: E402

pmc = json.load(open(sys.argv[1]))
workload, envs, humans, source = sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
per_kernel = {}
for name, k in pmc.items():
    if "step_kernel" not in name:  # the one launch of a step (orca_step_kernel / step_kernel)
        continue
    m = k["mean_per_dispatch"]
    per_kernel[name.replace("ebc::", "").split("<")[0]] = {
        "FETCH_SIZE_KB": m.get("FETCH_SIZE", 0.0), "WRITE_SIZE_KB": m.get("WRITE_SIZE", 0.0),
        "SQ_INSTS_VALU": m.get("SQ_INSTS_VALU", 0.0), "SQ_WAVES": m.get("SQ_WAVES", 0.0),
        "SQ_WAIT_ANY": m.get("SQ_WAIT_ANY", 0.0), "SQ_WAVE_CYCLES": m.get("SQ_WAVE_CYCLES", 0.0),
        "Scratch_Size": k["dispatch"].get("Scratch_Size"), "VGPR_Count": k["dispatch"].get("VGPR_Count")}
fetch = sum(v["FETCH_SIZE_KB"] for v in per_kernel.values())
write = sum(v["WRITE_SIZE_KB"] for v in per_kernel.values())
valu = sum(v["SQ_INSTS_VALU"] for v in per_kernel.values())
json.dump({
    # vector-ALU instructions issued per step (SQ_INSTS_VALU, summed over the launch's waves): bench.py turns it
    # into roofline.valu = issue time of these at 4 cycles each on 1024 SIMDs
    "SQ_INSTS_VALU_per_step": valu, "shader_clock_hz": 2.4e9,
    "valu_source": "SQ_INSTS_VALU of the same PMC run; clock = MI355X_MICROARCH.md max shader clock",
    "csrc_sha256": _capi.csrc_sha256(),  # bench.py reports these numbers only for the sources they were measured on
    "workload": workload, "envs_per_gpu": envs, "humans": humans, "source": source,
    "per_step": {"FETCH_SIZE_KB": fetch, "WRITE_SIZE_KB": write}, "per_kernel": per_kernel,
    "correction": "MI355X_MICROARCH.md HBM section: FETCH_SIZE counts 64 B per 128-B request on gfx950 -> "
                  "doubled; WRITE_SIZE as read; both in KB. Narrow (4-8 B/lane) accesses are uncalibrated, "
                  "so this is an upper estimate of the read side.",
    "traffic_bytes_per_step": (2.0 * fetch + write) * 1024.0}, sys.stdout, indent=1)
print()
