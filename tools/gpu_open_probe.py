#!/usr/bin/env python3
"""What makes a process count as "has the GPU open" on the GPU box (its process guard allows 6)?  Starts one child per
stage and lists the device files each holds.  Run on the GPU box; prints one line per stage."""
import os
import subprocess
import sys

STAGES = {
    "import torch": "import torch",
    "import torch + device_count": "import torch; torch.cuda.device_count()",
    "gloo group": ("import torch, torch.distributed as dist, os; os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29533');"
                   "dist.init_process_group('gloo', rank=0, world_size=1)"),
    "import ebcsim.batched": "import sys; sys.path.insert(0, 'eb-cadrl_amd'); import torch; from ebcsim.batched import BatchedEnv",
    "dlopen libebcsim": "import sys; sys.path.insert(0, 'eb-cadrl_amd'); from ebcsim import _capi; _capi.lib()",
    "cuda.is_available": "import torch; torch.cuda.is_available()",
}
for name, code in STAGES.items():
    prog = code + """
import os
held = []
for f in os.listdir('/proc/self/fd'):
    try:
        t = os.readlink('/proc/self/fd/' + f)
    except OSError:
        continue
    if 'kfd' in t or '/dri/' in t:
        held.append(t)
print(sorted(set(held)))
"""
    r = subprocess.run([sys.executable, "-c", prog], capture_output=True, text=True, timeout=300)
    print("%-30s -> %s %s" % (name, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else "", r.stderr.strip()[-200:]))
