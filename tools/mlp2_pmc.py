#!/usr/bin/env python3
"""The value network's three block shapes alone, a few launches each (for rocprofv3 --pmc / --kernel-trace)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "eb-cadrl_amd")):
    sys.path.insert(0, p)


def main():
    import numpy as np
    import torch
    from ebcsim import _capi
    L = _capi.lib()
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 19
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    for K0, H, O, final, group in ((17, 300, 200, False, 0), (200, 200, 100, False, 0), (200, 200, 200, True, 18)):
        rs = np.random.RandomState(0)
        w1 = (rs.randn(H, K0) / np.sqrt(K0)).astype(np.float32); b1 = rs.randn(H).astype(np.float32)
        w2 = (rs.randn(O, H) / np.sqrt(H)).astype(np.float32); b2 = rs.randn(O).astype(np.float32)
        w3 = rs.randn(O).astype(np.float32); b3 = np.zeros(1, np.float32)
        h = C.c_void_p()
        _capi.check(L.ebc_mlp2_create(0, K0, H, O, w1.ctypes.data, b1.ctypes.data, w2.ctypes.data, b2.ctypes.data,
                                      w3.ctypes.data if final else None, b3.ctypes.data if final else None, C.byref(h)))
        x = torch.randn(M, K0, device="cuda")
        y = torch.empty((M,) if final else (M, O), device="cuda")
        rb = torch.randn((M + group - 1) // group, H, device="cuda") if group else None
        st = torch.cuda.current_stream().cuda_stream
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for it in range(reps + 1):
            if it == 1:
                e0.record()
            _capi.check(L.ebc_mlp2_forward(h, st, x.data_ptr(), M, 0 if final else 1, rb.data_ptr() if group else None, group, y.data_ptr()))
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        fl = 2.0 * M * (K0 * H + H * O)
        print("%4d -> %4d -> %4d%s, M = %d: %7.3f ms  %6.1f TFLOP/s f32-equivalent, %6.1f TFLOP/s of bf16 MFMA work" % (
            K0, H, O, " (+1, group term)" if final else "", M, ms, fl / ms / 1e9, 3 * fl / ms / 1e9))


if __name__ == "__main__":
    main()
