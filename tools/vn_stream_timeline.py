#!/usr/bin/env python3
"""Where the streamed attention block's periods go: per period, cycles a wave spends in the counted wait, at the
barrier and computing (a -DEBC_VNS_TRACE library: tools/build_vns_variant.sh trace -DEBC_VNS_TRACE).

    EBCSIM_LIB=eb-cadrl_amd/lib/libebcsim_trace_vns.so python3 tools/vn_stream_timeline.py [rows]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "eb-cadrl_amd")):
    sys.path.insert(0, p)


def main():
    import numpy as np
    import torch
    from ebcsim.sarl import _NativeMlp2
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 359 * 81 * 18
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(3)
    K0, H, O, P = 200, 200, 200, 14

    def lin(o, i):
        return torch.randn(o, i, generator=g) / i ** 0.5, torch.randn(o, generator=g)
    src = _NativeMlp2([lin(300, 17), lin(K0, 300)], 0)
    att = _NativeMlp2([lin(H, K0), lin(O, H)], 0, final=(torch.randn(1, O, generator=g) / O ** 0.5, torch.randn(1, generator=g)),
                      in_fragments=True)
    x = torch.randn(M, 17, generator=g).to(dev)
    frag = _NativeMlp2.frag_buffer(M, K0, dev)
    src.forward_ex(M, True, x=x, want_y=False, seg_rows=18, want_partial=True, frag_out=frag)
    rb = torch.randn((M + 17) // 18, H, generator=g).to(dev)
    nw = int(os.environ.get("VNS_NW", 8))  # waves per workgroup of the library under test
    tiles = (M + 32 * nw - 1) // (32 * nw) * nw
    dbg = torch.zeros(tiles * P * 4, dtype=torch.int64, device=dev)
    for _ in range(3):
        att.forward_ex(M, False, frag_in=frag, row_bias=rb, group_rows=18, row_weight=dbg.view(torch.float32))
    torch.cuda.synchronize()
    t = dbg.cpu().numpy().reshape(tiles, P, 4).astype(np.int64)
    t = t[: (M + 31) // 32]
    wait = t[:, :, 1] - t[:, :, 0]
    barrier = t[:, :, 2] - t[:, :, 1]
    nxt = np.concatenate([t[:, 1:, 0], t[:, :1, 3]], axis=1)
    compute = nxt - t[:, :, 2]
    life = t[:, 0, 3] - t[:, 0, 0]
    print("waves %d; s_memtime ticks (shader cycles; 1.98 GHz under matrix load, tools/mfma_rate.hip) per period: median [p10 .. p90]" % len(t))
    for p in range(P):
        def q(a):
            return "%5.0f [%4.0f .. %5.0f]" % (np.median(a[:, p]), np.percentile(a[:, p], 10), np.percentile(a[:, p], 90))
        print("  period %2d (%s): wait %s   barrier %s   compute %s" % (p, "A" if p < 7 else "C", q(wait), q(barrier), q(compute)))
    print("  wave: first period top -> end: median %.0f; sums of medians: wait %.0f barrier %.0f compute %.0f" % (
        np.median(life), np.median(wait, 0).sum(), np.median(barrier, 0).sum(), np.median(compute, 0).sum()))


if __name__ == "__main__":
    main()
