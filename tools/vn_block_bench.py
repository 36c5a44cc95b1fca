#!/usr/bin/env python3
"""The attention block alone at the decision's chunk size: the streamed kernel against the general block (same bits),
microseconds per launch.  EBCSIM_LIB selects an A/B library.   python3 tools/vn_block_bench.py [rows]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "eb-cadrl_amd")):
    sys.path.insert(0, p)


def main():
    import torch
    from ebcsim.sarl import _NativeMlp2
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 359 * 81 * 18
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(3)
    K0, H, O = 200, 200, 200

    def lin(o, i):
        return torch.randn(o, i, generator=g) / i ** 0.5, torch.randn(o, generator=g)
    src = _NativeMlp2([lin(300, 17), lin(K0, 300)], 0)
    att = _NativeMlp2([lin(H, K0), lin(O, H)], 0, final=(torch.randn(1, O, generator=g) / O ** 0.5, torch.randn(1, generator=g)),
                      in_fragments=True)
    x = torch.randn(M, 17, generator=g).to(dev)
    frag = _NativeMlp2.frag_buffer(M, K0, dev)
    src.forward_ex(M, True, x=x, want_y=False, seg_rows=18, want_partial=True, frag_out=frag)
    rb = torch.randn((M + 17) // 18, H, generator=g).to(dev)
    out = {}
    for general in (False, True):
        for _ in range(3):
            y, _ = att.forward_ex(M, False, frag_in=frag, row_bias=rb, group_rows=18, general=general)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        for _ in range(20):
            y, _ = att.forward_ex(M, False, frag_in=frag, row_bias=rb, group_rows=18, general=general)
        ev[1].record()
        torch.cuda.synchronize()
        out[general] = (ev[0].elapsed_time(ev[1]) * 50, y)
    print("%s: %d rows: streamed %.1f us, general %.1f us, same bits: %s" % (
        os.path.basename(os.environ.get("EBCSIM_LIB", "libebcsim.so")), M, out[False][0], out[True][0], bool(torch.equal(out[False][1], out[True][1]))))


if __name__ == "__main__":
    main()
