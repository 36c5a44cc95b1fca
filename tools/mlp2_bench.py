#!/usr/bin/env python3
"""Time the split-bf16 fused two-layer block against torch float32 on the value network's shapes."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "eb-cadrl_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)


def main():
    import numpy as np
    import torch
    from ebcsim import _capi
    from test_value_net import _lib
    L = _lib()
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 1024 * 81 * 18
    for K0, H, O in ((17, 300, 200), (200, 200, 100), (200, 200, 1), (13, 150, 100)):
        rs = np.random.RandomState(0)
        w1 = (rs.randn(H, K0) / np.sqrt(K0)).astype(np.float32); b1 = rs.randn(H).astype(np.float32)
        w2 = (rs.randn(O, H) / np.sqrt(H)).astype(np.float32); b2 = rs.randn(O).astype(np.float32)
        h = C.c_void_p()
        _capi.check(L.ebc_mlp2_create(0, K0, H, O, w1.ctypes.data, b1.ctypes.data, w2.ctypes.data, b2.ctypes.data, None, None, C.byref(h)))
        x = torch.randn(M, K0, device="cuda"); y = torch.empty(M, O, device="cuda")
        tw1, tb1, tw2, tb2 = (torch.from_numpy(a).cuda() for a in (w1, b1, w2, b2))
        st = torch.cuda.current_stream().cuda_stream

        def native():
            _capi.check(L.ebc_mlp2_forward(h, st, x.data_ptr(), M, 1, None, 0, y.data_ptr()))

        def ref():
            return torch._addmm_activation(tb2, torch._addmm_activation(tb1, x, tw1.t()), tw2.t())
        for f in (native, ref):
            for _ in range(3):
                f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                f()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            fl = 2.0 * M * (K0 * H + H * O)
            print("%4d -> %4d -> %4d, M = %d: %-6s %7.3f ms  %6.1f TFLOP/s (f32-equivalent)" % (K0, H, O, M, f.__name__, ms, fl / ms / 1e9))


if __name__ == "__main__":
    main()
