#!/usr/bin/env python3
"""ebc_mlp2_forward_f32 timed per launch at several row counts (EBCSIM_F32_FORM=1: four tiles per workgroup on LDS-staged
weights, =2: a workgroup per tile): python3 tools/f32_forms.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "eb-cadrl_amd")):
    sys.path.insert(0, p)


def main():
    import torch
    from ebcsim.sarl import _NativeMlp2
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(1)
    print("form", os.environ.get("EBCSIM_F32_FORM", "auto"))
    for K0, H, O, tail in ((17, 150, 100, False), (200, 100, 100, True), (200, 200, 200, True), (106, 150, 100, False)):
        w1, b1 = torch.randn(H, K0, generator=g) / K0 ** 0.5, torch.randn(H, generator=g)
        w2, b2 = torch.randn(O, H, generator=g) / H ** 0.5, torch.randn(O, generator=g)
        fin = (torch.randn(1, O, generator=g) / O ** 0.5, torch.randn(1, generator=g)) if tail else None
        blk = _NativeMlp2([(w1, b1), (w2, b2)], 0, final=fin)
        for M in (1080, 8192, 36864, 131072, 524288):
            x = torch.randn(M, K0, generator=g).to(dev)
            for _ in range(3):
                blk.f32(x, False)
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            ev[0].record()
            for _ in range(10):
                blk.f32(x, False)
            ev[1].record()
            torch.cuda.synchronize()
            print("  %3d -> %3d -> %3d%s  M %7d: %8.1f us" % (K0, H, O, " -> 1" if tail else "     ", M, ev[0].elapsed_time(ev[1]) * 100))


if __name__ == "__main__":
    main()
