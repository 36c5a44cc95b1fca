"""Dense blocks of the SARL value network on the bf16 matrix cores with split operands
(csrc/ebc_value_net.h) against torch float32."""
import ctypes as C

import numpy as np
import pytest
import torch


def _lib():
    from ebcsim import _capi
    return _capi.lib()


@pytest.mark.gpu
@pytest.mark.parametrize("K0,H,O,M,relu_out", [(17, 300, 200, 1000, 1), (13, 150, 100, 333, 1),
                                               (200, 200, 100, 515, 0), (100, 100, 50, 64, 0),
                                               (200, 200, 1, 2049, 0), (5, 7, 3, 31, 1)])
def test_mlp2_split_bf16_matches_float32(K0, H, O, M, relu_out):
    import torch
    from ebcsim import _capi
    L = _lib()
    rs = np.random.RandomState(K0 * 1000 + H)
    w1 = (rs.randn(H, K0) / np.sqrt(K0)).astype(np.float32)
    b1 = rs.randn(H).astype(np.float32) * 0.1
    w2 = (rs.randn(O, H) / np.sqrt(H)).astype(np.float32)
    b2 = rs.randn(O).astype(np.float32) * 0.1
    x = (rs.randn(M, K0) * 2).astype(np.float32)
    h = C.c_void_p()
    _capi.check(L.ebc_mlp2_create(0, K0, H, O, w1.ctypes.data, b1.ctypes.data, w2.ctypes.data, b2.ctypes.data,
                                  None, None, C.byref(h)))
    xd = torch.from_numpy(x).cuda()
    yd = torch.zeros((M, O), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    _capi.check(L.ebc_mlp2_forward(h, None, xd.data_ptr(), M, relu_out, None, 0, yd.data_ptr()))
    torch.cuda.synchronize()
    ref = np.maximum(x.astype(np.float64) @ w1.T.astype(np.float64) + b1, 0) @ w2.T.astype(np.float64) + b2
    if relu_out:
        ref = np.maximum(ref, 0)
    got = yd.cpu().numpy()
    f32 = torch.nn.functional.linear(torch.relu(torch.nn.functional.linear(torch.from_numpy(x), torch.from_numpy(w1), torch.from_numpy(b1))), torch.from_numpy(w2), torch.from_numpy(b2)).numpy()
    if relu_out:
        f32 = np.maximum(f32, 0)
    err = np.abs(got - ref).max()
    err32 = np.abs(f32 - ref).max()
    scale = np.abs(ref).max()
    # three bf16 products per f32 product: relative error per term ~2^-16; float32 itself is ~1e-6 here
    assert err <= 4e-5 * max(scale, 1.0), (err, err32, scale)
    _capi.check(L.ebc_mlp2_destroy(h))


@pytest.mark.gpu
@pytest.mark.parametrize("K0,H,O,R,B", [(200, 200, 200, 18, 777), (100, 100, 100, 5, 1031), (40, 70, 33, 7, 64)])
def test_mlp2_attention_form(K0, H, O, R, B):
    """The attention stack's form: a per-group term added to the hidden pre-activation (the pair's mean
    state through its half of layer 0) and a third layer with one output taken from the accumulators."""
    import torch
    from ebcsim import _capi
    L = _lib()
    rs = np.random.RandomState(K0 + R)
    M = B * R
    w1 = (rs.randn(H, K0) / np.sqrt(K0)).astype(np.float32); b1 = (rs.randn(H) * 0.1).astype(np.float32)
    w2 = (rs.randn(O, H) / np.sqrt(H)).astype(np.float32); b2 = (rs.randn(O) * 0.1).astype(np.float32)
    w3 = (rs.randn(O) / np.sqrt(O)).astype(np.float32); b3 = np.array([0.3], np.float32)
    x = rs.randn(M, K0).astype(np.float32)
    g = rs.randn(B, H).astype(np.float32)
    h = C.c_void_p()
    _capi.check(L.ebc_mlp2_create(0, K0, H, O, w1.ctypes.data, b1.ctypes.data, w2.ctypes.data, b2.ctypes.data,
                                  w3.ctypes.data, b3.ctypes.data, C.byref(h)))
    xd, gd = torch.from_numpy(x).cuda(), torch.from_numpy(g).cuda()
    yd = torch.zeros(M, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    _capi.check(L.ebc_mlp2_forward(h, None, xd.data_ptr(), M, 0, gd.data_ptr(), R, yd.data_ptr()))
    torch.cuda.synchronize()
    x64, g64 = x.astype(np.float64), np.repeat(g.astype(np.float64), R, axis=0)
    a1 = np.maximum(x64 @ w1.T.astype(np.float64) + b1 + g64, 0)
    a2 = np.maximum(a1 @ w2.T.astype(np.float64) + b2, 0)
    ref = a2 @ w3.astype(np.float64) + b3[0]
    err = np.abs(yd.cpu().numpy() - ref).max()
    assert err <= 4e-5 * max(np.abs(ref).max(), 1.0), err
    _capi.check(L.ebc_mlp2_destroy(h))


@pytest.mark.gpu
@pytest.mark.parametrize("B,R,H,F,ragged", [(257, 18, 200, 100, True), (64, 5, 100, 52, False), (33, 70, 8, 4, True)])
def test_pair_glue_matches_torch(B, R, H, F, ragged):
    """ebc_pair_mean / ebc_pair_attend against the torch float32 form of sarl.py:56-58 and :69-76
    (masked mean over the pair's rows; exp(s) * (s != 0) normalised, weighted feature sum), ragged
    row counts, exact-zero scores, more rows than lanes."""
    import torch
    from ebcsim.sarl import SarlValueNet
    g = torch.Generator().manual_seed(B * 31 + R)
    h = torch.randn(B * R, H, generator=g).cuda()
    feat = torch.randn(B, R, F, generator=g).cuda()
    scores = torch.randn(B, R, generator=g)
    scores[torch.rand(B, R, generator=g) < 0.1] = 0.0   # the reference treats a zero score as "no row"
    scores = scores.cuda()
    nv = None
    if ragged:
        nv = torch.randint(1, R + 1, (B,), generator=g).cuda()
    valid = torch.ones(B, R, device="cuda") if nv is None else (torch.arange(R, device="cuda")[None, :] < nv[:, None]).float()
    denom = float(R) if nv is None else nv.float()[:, None]
    ref_g = (h.view(B, R, H) * valid[:, :, None]).sum(1) / denom
    e = torch.exp(scores) * (scores != 0).float() * valid
    ref_o = ((e / e.sum(1, keepdim=True)).unsqueeze(2) * feat).sum(1)
    got_g = SarlValueNet._pair_mean(h, nv, B, R)
    got_o = SarlValueNet._pair_attend(scores, feat, nv, B, R)
    torch.cuda.synchronize()
    torch.testing.assert_close(got_g, ref_g, atol=2e-6, rtol=1e-5)
    torch.testing.assert_close(got_o, ref_o, atol=2e-6, rtol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("K0,H,O,R,B,ragged,store", [(17, 300, 200, 18, 777, False, True), (200, 200, 100, 18, 1031, True, False),
                                                     (13, 150, 100, 16, 65, True, True), (40, 64, 52, 32, 33, False, False),
                                                     (17, 300, 200, 23, 101, True, True)])
def test_mlp2_reduce_epilogue_gives_pair_sums(K0, H, O, R, B, ragged, store):
    """ebc_mlp2_forward_reduce + ebc_pair_combine: the rows of the block equal ebc_mlp2_forward's bit for bit, and the
    per-pair (weighted, masked) sums equal the sums of those rows in float64 to float32 rounding — the mean of
    sarl.py:56-58 with the mask as weights, the attention-weighted feature sum of sarl.py:73-76 with
    ebc_pair_weights; with store = 0 the rows are not written at all."""
    import torch
    from ebcsim import _capi
    L = _lib()
    rs = np.random.RandomState(K0 + 7 * R + B)
    M = B * R
    w1 = (rs.randn(H, K0) / np.sqrt(K0)).astype(np.float32)
    b1 = (rs.randn(H) * 0.1).astype(np.float32)
    w2 = (rs.randn(O, H) / np.sqrt(H)).astype(np.float32)
    b2 = (rs.randn(O) * 0.1).astype(np.float32)
    x = (rs.randn(M, K0) * 2).astype(np.float32)
    nv = rs.randint(1, R + 1, size=B).astype(np.int64) if ragged else None
    h = C.c_void_p()
    _capi.check(L.ebc_mlp2_create(0, K0, H, O, w1.ctypes.data, b1.ctypes.data, w2.ctypes.data, b2.ctypes.data, None, None, C.byref(h)))
    xd = torch.from_numpy(x).cuda()
    plain = torch.zeros((M, O), dtype=torch.float32, device="cuda")
    _capi.check(L.ebc_mlp2_forward(h, None, xd.data_ptr(), M, 1, None, 0, plain.data_ptr()))
    nvd = None if nv is None else torch.from_numpy(nv).cuda()
    nvp = None if nvd is None else nvd.data_ptr()
    # (a) mean: weights = the mask (or none)
    wd = None
    if nvd is not None:
        wd = torch.zeros(M, dtype=torch.float32, device="cuda")
        _capi.check(L.ebc_pair_mask(None, nvp, B, R, wd.data_ptr()))
    yd = torch.full((M, O), -7.0, dtype=torch.float32, device="cuda")
    part = torch.full(((M + 31) // 32, 3, O), float("nan"), dtype=torch.float64, device="cuda")
    _capi.check(L.ebc_mlp2_forward_reduce(h, None, xd.data_ptr(), M, 1, None, 0, yd.data_ptr() if store else None, R,
                                          None if wd is None else wd.data_ptr(), part.data_ptr()))
    mean = torch.zeros((B, O), dtype=torch.float32, device="cuda")
    _capi.check(L.ebc_pair_combine(None, part.data_ptr(), nvp, B, R, O, 1, mean.data_ptr()))
    torch.cuda.synchronize()
    rows = plain.cpu().numpy().astype(np.float64).reshape(B, R, O)
    if store:
        np.testing.assert_array_equal(yd.cpu().numpy(), plain.cpu().numpy())
    else:
        assert (yd.cpu().numpy() == -7.0).all()
    valid = np.ones((B, R)) if nv is None else (np.arange(R)[None, :] < nv[:, None]).astype(np.float64)
    want = (rows * valid[:, :, None]).sum(1) / valid.sum(1)[:, None]
    np.testing.assert_allclose(mean.cpu().numpy(), want, rtol=2e-6, atol=2e-6 * np.abs(rows).max())
    # (b) attention-weighted sum: weights = softmax' of random scores (some exactly 0: masked by the reference's rule)
    sc = rs.randn(B, R).astype(np.float32)
    sc[rs.rand(B, R) < 0.1] = 0.0
    sc[:, 0] = np.where(sc[:, 0] == 0, 0.5, sc[:, 0])  # at least one live row per pair
    scd = torch.from_numpy(sc).cuda()
    w = torch.zeros(M, dtype=torch.float32, device="cuda")
    _capi.check(L.ebc_pair_weights(None, scd.data_ptr(), nvp, B, R, w.data_ptr()))
    _capi.check(L.ebc_mlp2_forward_reduce(h, None, xd.data_ptr(), M, 1, None, 0, None, R, w.data_ptr(), part.data_ptr()))
    att = torch.zeros((B, O), dtype=torch.float32, device="cuda")
    _capi.check(L.ebc_pair_combine(None, part.data_ptr(), None, B, R, O, 0, att.data_ptr()))
    old = torch.zeros((B, O), dtype=torch.float32, device="cuda")
    _capi.check(L.ebc_pair_attend(None, scd.data_ptr(), plain.data_ptr(), nvp, B, R, O, old.data_ptr()))
    torch.cuda.synchronize()
    e = np.exp(sc.astype(np.float64)) * (sc != 0) * valid
    wt = e / e.sum(1, keepdims=True)
    np.testing.assert_allclose(w.cpu().numpy().reshape(B, R), wt, rtol=3e-6, atol=1e-7)
    want = (rows * wt[:, :, None]).sum(1)
    np.testing.assert_allclose(att.cpu().numpy(), want, rtol=3e-6, atol=3e-6 * np.abs(rows).max())
    np.testing.assert_allclose(att.cpu().numpy(), old.cpu().numpy(), rtol=3e-6, atol=3e-6 * np.abs(rows).max())
    _capi.check(L.ebc_mlp2_destroy(h))


@pytest.mark.gpu
def test_mlp2_reduce_refuses_what_it_cannot_do():
    import torch
    from ebcsim import _capi
    L = _lib()
    rs = np.random.RandomState(1)
    w1, b1 = rs.randn(8, 5).astype(np.float32), rs.randn(8).astype(np.float32)
    w2, b2 = rs.randn(4, 8).astype(np.float32), rs.randn(4).astype(np.float32)
    h = C.c_void_p()
    _capi.check(L.ebc_mlp2_create(0, 5, 8, 4, w1.ctypes.data, b1.ctypes.data, w2.ctypes.data, b2.ctypes.data, None, None, C.byref(h)))
    x = torch.zeros((64, 5), device="cuda")
    part = torch.zeros((2, 3, 4), dtype=torch.float64, device="cuda")
    assert L.ebc_mlp2_forward_reduce(h, None, x.data_ptr(), 64, 1, None, 0, None, 8, None, part.data_ptr()) == -2  # EBC_ERR_UNSUPPORTED: groups of 8 rows
    assert L.ebc_mlp2_forward_reduce(h, None, x.data_ptr(), 64, 1, None, 0, None, 16, None, part.data_ptr()) == -2  # a block of 1 + 1 tiles has no tile epilogue
    _capi.check(L.ebc_mlp2_destroy(h))


@pytest.mark.gpu
def test_block_repacked_on_the_device_equals_the_host_pack():
    """ebc_mlp2_update: a block created from OTHER weights and refreshed from device tensors gives, bit for bit, the
    outputs of a block created from those weights on the host — the split-bf16 forward, its one-output tail and the
    float32 form (the fragments, biases and float32 copies are the host packer's)."""
    from ebcsim.sarl import _NativeMlp2
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(5)
    for K0, H, O, tail in ((17, 300, 200, False), (200, 200, 100, False), (200, 200, 200, True), (206, 300, 200, False), (200, 200, 1, False)):
        mk = lambda: [(torch.randn(H, K0, generator=g) / K0 ** 0.5, torch.randn(H, generator=g)),  # noqa: E731
                      (torch.randn(O, H, generator=g) / H ** 0.5, torch.randn(O, generator=g))]
        mkf = lambda: (torch.randn(1, O, generator=g), torch.randn(1, generator=g))  # noqa: E731
        want_l, want_f = mk(), (mkf() if tail else None)
        a = _NativeMlp2(want_l, 0, final=want_f)
        b = _NativeMlp2(mk(), 0, final=(mkf() if tail else None))
        x = torch.randn(1000, K0, generator=g).to(dev)
        assert not torch.equal(a(x, True), b(x, True))
        b.update([(w.to(dev), bb.to(dev)) for w, bb in want_l], None if want_f is None else tuple(t.to(dev) for t in want_f))
        torch.cuda.synchronize()
        for relu in (True, False):
            assert torch.equal(a(x, relu), b(x, relu)), (K0, H, O, relu)
            assert torch.equal(a.f32(x, relu), b.f32(x, relu)), (K0, H, O, relu)


@pytest.mark.gpu
def test_float32_block_is_float32_gemm_grade():
    """ebc_mlp2_forward_f32 against torch float32 Linear layers (and float64 ones as the yard-stick of both): the
    refinement's values are as good as a float32 GEMM's, with the per-group term and the one-output tail."""
    from ebcsim.sarl import _NativeMlp2
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(6)
    for K0, H, O, tail, group in ((17, 300, 200, False, 0), (200, 200, 100, False, 0), (200, 200, 200, True, 18), (206, 300, 200, False, 0),
                                  (200, 200, 1, False, 0)):
        w1, b1 = torch.randn(H, K0, generator=g) / K0 ** 0.5, torch.randn(H, generator=g)
        w2, b2 = torch.randn(O, H, generator=g) / H ** 0.5, torch.randn(O, generator=g)
        fin = (torch.randn(1, O, generator=g) / O ** 0.5, torch.randn(1, generator=g)) if tail else None
        blk = _NativeMlp2([(w1, b1), (w2, b2)], 0, final=fin)
        for M in (18 * 53 + 7, (1 << 16) + 18 * 2 + 5) if K0 == 200 else (18 * 53 + 7,):  # few rows: a workgroup per tile; many: four tiles
            _check_f32_block(blk, M, K0, H, O, w1, b1, w2, b2, fin, group, g, dev)


def _check_f32_block(blk, M, K0, H, O, w1, b1, w2, b2, fin, group, g, dev):
    if True:
        x = torch.randn(M, K0, generator=g)
        rb = torch.randn((M + 17) // 18, H, generator=g) if group else None

        def ref(dt):
            h = torch.nn.functional.linear(x.to(dt), w1.to(dt), b1.to(dt))
            if rb is not None:
                h = h + rb.to(dt).repeat_interleave(18, 0)[:M]
            y = torch.nn.functional.linear(torch.relu(h), w2.to(dt), b2.to(dt))
            if fin is not None:  # the one-output tail acts on relu(out): the attention stack's third layer (sarl.py:25-27)
                y = torch.nn.functional.linear(torch.relu(y), fin[0].to(dt), fin[1].to(dt)).squeeze(1)
            return y
        got = blk.f32(x.to(dev), False, row_bias=None if rb is None else rb.to(dev), group_rows=18 if group else 0).cpu()
        exact = ref(torch.float64)
        err_native = float((got.double() - exact).abs().max())
        err_torch = float((ref(torch.float32).double() - exact).abs().max())
        assert err_native <= max(3.0 * err_torch, 2e-6), (K0, H, O, err_native, err_torch)


@pytest.mark.gpu
def test_fragment_handoff_equals_row_handoff():
    """h1 handed from mlp1 to the attention stack and to mlp2 as split fragments (ebc_mlp2_forward_ex: frag_out ->
    frag_in) against the same network with h1 as float32 rows: the same float32 values split the same way, multiplied
    in another k order — values agree to float32 rounding, ragged pairs included."""
    import os
    from ebcsim.sarl import SarlValueNet
    from helpers import GOLDEN
    dev = torch.device("cuda", 0)
    net = SarlValueNet.load(os.path.join(GOLDEN, "weights", "sarl_n10_ebcadrl.pth"), device=str(dev))
    g = torch.Generator().manual_seed(9)
    B, R, T = 777, 18, 17
    rows = torch.randn(B, R, T, generator=g).to(dev)
    rows[:, :, 13:] = 0
    rows[:, :, 13] = 1
    nv = torch.randint(16, R + 1, (B,), generator=g).to(dev)
    for n_valid in (None, nv):
        net.frag_handoff = True
        a = net.forward(rows, n_valid)
        assert net._native_frag and net.folded_forwards > 0
        net.frag_handoff = False
        b = net.forward(rows, n_valid)
        ref = net.forward(rows, n_valid, exact=True)
        torch.cuda.synchronize()
        ea, eb = float((a - ref).abs().max()), float((b - ref).abs().max())
        assert ea <= max(2.0 * eb, 2e-5 * max(1.0, float(ref.abs().max()))), (ea, eb)
        print("fragment hand-off error %.3g, row hand-off error %.3g" % (ea, eb))


@pytest.mark.gpu
def test_streamed_attention_block_equals_the_general_block_bit_for_bit():
    """The attention block at 7 x 7 x 7 tiles runs as the streamed kernel (ebc_vn_stream.h: periods over input tiles, then
    over output tiles; nothing wide resident).  Same products, same sums, same order as the general block
    (EBC_MLP_GENERAL_KERNEL): the scores must be the same BITS — whole pairs, ragged pairs, a last tile that is not
    full, a workgroup whose last waves lie past the rows."""
    from ebcsim.sarl import _NativeMlp2
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(11)
    for K0, H, O in ((200, 200, 200), (224, 200, 200), (200, 224, 196), (220, 212, 224)):
        w1, b1 = torch.randn(H, K0, generator=g) / K0 ** 0.5, torch.randn(H, generator=g)
        w2, b2 = torch.randn(O, H, generator=g) / H ** 0.5, torch.randn(O, generator=g)
        fin = (torch.randn(1, O, generator=g) / O ** 0.5, torch.randn(1, generator=g))
        src = _NativeMlp2([(torch.randn(64, 17, generator=g), torch.randn(64, generator=g)),
                           (torch.randn(K0, 64, generator=g) / 8, torch.randn(K0, generator=g))], 0)
        blk = _NativeMlp2([(w1, b1), (w2, b2)], 0, final=fin, in_fragments=True)
        for M in (18 * 600, 18 * 57 + 5, 31, 32 * 8 * 3 + 1):
            x = torch.randn(M, 17, generator=g).to(dev)
            frag = _NativeMlp2.frag_buffer(M, K0, dev)
            src.forward_ex(M, True, x=x, want_y=False, seg_rows=18, want_partial=True, frag_out=frag)
            rb = torch.randn((M + 17) // 18, H, generator=g).to(dev)
            a, _ = blk.forward_ex(M, False, frag_in=frag, row_bias=rb, group_rows=18)
            b, _ = blk.forward_ex(M, False, frag_in=frag, row_bias=rb, group_rows=18, general=True)
            torch.cuda.synchronize()
            assert torch.equal(a, b), (K0, H, O, M, float((a - b).abs().max()))
            assert bool(torch.isfinite(a).all()) and float(a.abs().max()) > 0


@pytest.mark.gpu
def test_streamed_feature_block_equals_the_general_block_to_rounding():
    """`mlp2` at 7 x 7 x 4 tiles (fragment input, its rows only as attention-weighted pair sums) runs as the streamed
    kernel too.  Its hidden tiles add up in ONE chain where the general block alternates two, so the sums differ in the
    last bits: both are held against a float64 evaluation of the same rows, the streamed one must be as close as the
    general one; and whole copies of a pair anywhere in the batch still get bit-identical sums."""
    from ebcsim.sarl import SarlValueNet, _NativeMlp2
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(12)
    R = 18
    for K0, H, O in ((200, 200, 100), (224, 212, 128)):
        def lin(o, i):
            return torch.randn(o, i, generator=g) / i ** 0.5, torch.randn(o, generator=g)
        l0, l1 = lin(64, 17), lin(K0, 64)
        src = _NativeMlp2([l0, l1], 0)
        w1, w2 = lin(H, K0), lin(O, H)
        blk = _NativeMlp2([w1, w2], 0, in_fragments=True)
        for B in (600, 57, 1):
            M = B * R
            pair = torch.randn(R, 17, generator=g)
            x = torch.randn(M, 17, generator=g)
            if B > 40:
                x[7 * R:8 * R] = pair
                x[33 * R:34 * R] = pair  # the same pair at another place of the batch (another tile alignment)
            x = x.to(dev)
            frag = _NativeMlp2.frag_buffer(M, K0, dev)
            rows, _ = src.forward_ex(M, True, x=x, want_y=True, seg_rows=R, want_partial=True, frag_out=frag)
            w = torch.rand(M, generator=g).to(dev)
            if B > 40:
                w[33 * R:34 * R] = w[7 * R:8 * R]
            out = {}
            for general in (False, True):
                _, part = blk.forward_ex(M, False, frag_in=frag, want_y=False, seg_rows=R, row_weight=w, want_partial=True, general=general)
                out[general] = SarlValueNet._pair_combine(part, None, B, R, False)
            torch.cuda.synchronize()
            h = torch.relu(torch.nn.functional.linear(rows.double().cpu(), w1[0].double(), w1[1].double()))
            yref = torch.nn.functional.linear(h, w2[0].double(), w2[1].double())
            ref = (yref * w.double().cpu()[:, None]).view(B, R, O).sum(1)
            es, eg = float((out[False].double().cpu() - ref).abs().max()), float((out[True].double().cpu() - ref).abs().max())
            assert es <= max(2.0 * eg, 1e-5 * max(1.0, float(ref.abs().max()))), (K0, H, O, B, es, eg)
            if B > 40:
                assert torch.equal(out[False][7], out[False][33])


@pytest.mark.gpu
def test_streamed_first_block_equals_the_general_block_to_rounding():
    """`mlp1` (observation rows -> 300 -> 200: one input tile, ten hidden, seven output tiles) as the streamed kernel:
    its output leaves as the consumers' fragments and as masked pair sums.  Against the general block (another order of
    the hidden sums: last bits differ) and a float64 evaluation of the same rows; ragged pairs, a last tile that is not
    full; whole copies of a pair get bit-identical fragments' sums."""
    from ebcsim.sarl import SarlValueNet, _NativeMlp2
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(13)
    R, K0, H, O = 18, 17, 300, 200

    def lin(o, i):
        return torch.randn(o, i, generator=g) / i ** 0.5, torch.randn(o, generator=g)
    w1, w2 = lin(H, K0), lin(O, H)
    blk = _NativeMlp2([w1, w2], 0)
    for B in (700, 29, 1):
        M = B * R
        x = torch.randn(M, K0, generator=g)
        if B > 40:
            x[35 * R:36 * R] = x[9 * R:10 * R]
        nv = torch.randint(12, R + 1, (B,), generator=g)
        mask = (torch.arange(R)[None, :] < nv[:, None]).float().reshape(M)
        if B > 40:
            mask[35 * R:36 * R] = mask[9 * R:10 * R]
        xd, md = x.to(dev), mask.to(dev)
        out = {}
        for general in (False, True):
            frag = _NativeMlp2.frag_buffer(M, O, dev)
            frag.zero_()
            _, part = blk.forward_ex(M, True, x=xd, want_y=False, seg_rows=R, row_weight=md, want_partial=True, frag_out=frag, general=general)
            f = frag.view(torch.bfloat16).float()                      # [tile][column tile][k-step][hi, lo][lane][8]
            out[general] = (SarlValueNet._pair_combine(part, None, B, R, False), f[:, :, :, 0] + f[:, :, :, 1])
        torch.cuda.synchronize()
        h = torch.relu(torch.nn.functional.linear(x.double(), w1[0].double(), w1[1].double()))
        yref = torch.relu(torch.nn.functional.linear(h, w2[0].double(), w2[1].double()))
        ref = (yref * mask.double()[:, None]).view(B, R, O).sum(1)
        es, eg = float((out[False][0].double().cpu() - ref).abs().max()), float((out[True][0].double().cpu() - ref).abs().max())
        assert es <= max(2.0 * eg, 1e-5 * max(1.0, float(ref.abs().max()))), (B, es, eg)
        d = float((out[False][1] - out[True][1]).abs().max())
        assert d <= 2e-5 * max(1.0, float(out[True][1].abs().max())), (B, d)
        assert float(out[False][1].abs().max()) > 0
        if B > 40:
            assert torch.equal(out[False][0][9], out[False][0][35])
