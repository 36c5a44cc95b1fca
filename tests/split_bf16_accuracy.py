#!/usr/bin/env python3
"""How much accuracy does a bf16-MFMA value network keep if every f32 operand is split into bf16
hi + lo and the cross terms are kept?  Emulated on the CPU (products of bf16 values are exact in
f32, accumulation in f32 as the MFMA does) on the golden SARL decision runs.

    python3 tests/split_bf16_accuracy.py
terms 1 = plain bf16, 3 = hi*hi + hi*lo + lo*hi, 4 = all four."""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, 'eb-cadrl_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, _p)
from helpers import GOLDEN, load, params_of, batch_from_init
from ebcsim import _abi
from ebcsim.sarl import DeviceSarlPolicy, SarlValueNet
import ebcsim.sarl as sarl
from oracle import oracle
F=torch.nn.functional
def split(x):
    h=x.to(torch.bfloat16).float(); l=(x-h).to(torch.bfloat16).float(); return h,l
MODE=[3]
orig_linear=F.linear
def lin(x,w,b=None):
    xh,xl=split(x); wh,wl=split(w)
    y=orig_linear(xh,wh)
    if MODE[0]>=3: y=y+orig_linear(xh,wl)+orig_linear(xl,wh)
    if MODE[0]>=4: y=y+orig_linear(xl,wl)
    return y if b is None else y+b
for name in ["sarl_a5_baseline","sarl_n10_ebcadrl"]:
    z=load(name); meta=json.loads(str(z["meta"])); params=params_of(z); b=batch_from_init(z)
    net=SarlValueNet.load(os.path.join(GOLDEN,"weights",meta["weights"]))
    pol=DeviceSarlPolicy(net,z["action_space"],meta["gamma"]); v_pref=float(b.robot[0,7])
    for mode in (1,3,4):
        MODE[0]=mode
        env=oracle.OracleEnv(params,1,b.N,b.S); env.reset(b)
        worst=0; flips=0; n=0
        for t in range(min(len(z["action"]),40)):
            la=env.lookahead(z["action_space"],human_policy=_abi.HUMAN_ORCA)
            rows=torch.from_numpy(la["rows_rotated"]); rew=torch.from_numpy(la["reward"])
            F.linear=orig_linear
            ref=pol.values_from(rows,rew,None,params.time_step,v_pref)[0].numpy()
            F.linear=lin
            got=pol.values_from(rows,rew,None,params.time_step,v_pref)[0].numpy()
            F.linear=orig_linear
            worst=max(worst,np.abs(got-ref).max()); flips+=int(np.argmax(got)!=np.argmax(ref)); n+=1
            env.step(robot_action=z["action"][t][None],human_policy=_abi.HUMAN_CACHED)
        print(name,"bf16 terms",mode,"max |dV| %.2e"%worst,"argmax flips %d/%d"%(flips,n))
