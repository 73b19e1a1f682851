"""GPU parity (through the C ABI): Conv1d / ConvTranspose1d MFMA kernels vs plain torch fp32 CPU ops."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL = 2e-5  # fp32 fmaf-chain vs torch's CPU summation order; inputs O(1), K up to ~3k


def _mk(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g) * scale


@pytest.mark.parametrize('cin,cout,k,d,L,B', [
    (32, 32, 3, 1, 700, 2), (32, 32, 11, 5, 1030, 2), (64, 64, 7, 3, 513, 1), (128, 128, 11, 1, 300, 2),
    (256, 256, 3, 5, 97, 1), (80, 512, 7, 1, 50, 2), (32, 1, 7, 1, 999, 2), (1, 20, 7, 1, 240, 3),
    (20, 20, 7, 1, 65, 1), (64, 256, 3, 1, 7, 2), (512, 80, 5, 1, 33, 1), (16, 8, 3, 1, 1, 1), (8, 4, 11, 5, 9, 2),
])
def test_conv1d_matches_torch(cin, cout, k, d, L, B):
    from ttscube_amd.hip_layers import Conv1dHip
    pad = d * (k - 1) // 2
    w = _mk((cout, cin, k), 1, 1.0 / (cin * k) ** 0.5)
    b = _mk((cout,), 2, 0.1)
    x = _mk((B, cin, L), 3)
    conv = Conv1dHip(cin, cout, k, padding=pad, dilation=d)
    conv.set_weight(w, b)
    y = conv(x.cuda()).cpu()
    ref = F.conv1d(x, w, b, padding=pad, dilation=d)
    assert y.shape == ref.shape
    assert float((y - ref).abs().max()) < TOL


def test_conv1d_fused_prologue_epilogue():
    from ttscube_amd.hip_layers import Conv1dHip
    cin = cout = 64
    k, d, L, B = 7, 3, 400, 2
    pad = d * (k - 1) // 2
    w = _mk((cout, cin, k), 1, 1.0 / (cin * k) ** 0.5)
    b = _mk((cout,), 2, 0.1)
    x = _mk((B, cin, L), 3)
    r = _mk((B, cout, L), 4)
    acc0 = _mk((B, cout, L), 5)
    conv = Conv1dHip(cin, cout, k, padding=pad, dilation=d)
    conv.set_weight(w, b)
    out = acc0.clone().cuda()
    conv(x.cuda(), resid=r.cuda(), out=out, in_scale=1.0 / 3.0, in_slope=0.1, out_scale=0.5, act='tanh', accumulate=True)
    ref = acc0 + torch.tanh((F.conv1d(F.leaky_relu(x / 3.0, 0.1), w, b, padding=pad, dilation=d) + r) * 0.5)
    assert float((out.cpu() - ref).abs().max()) < TOL
    # in-place residual update (resid is out), as the resblock chain uses it
    xr = x.clone().cuda()
    conv(r.cuda(), resid=xr, out=xr, in_slope=0.1)
    ref2 = x + F.conv1d(F.leaky_relu(r, 0.1), w, b, padding=pad, dilation=d)
    assert float((xr.cpu() - ref2).abs().max()) < TOL


@pytest.mark.parametrize('cin,cout,k,s,L,B', [
    (512, 256, 16, 5, 50, 2), (256, 128, 16, 3, 101, 1), (128, 64, 4, 4, 304, 2), (64, 32, 4, 4, 1216, 1),
    (32, 16, 16, 8, 10, 2), (16, 8, 3, 2, 5, 1), (64, 32, 16, 5, 1, 2), (8, 4, 11, 3, 2, 1),
])
def test_conv_transpose1d_matches_torch(cin, cout, k, s, L, B):
    from ttscube_amd.hip_layers import Conv1dHip
    pad = (k - s) // 2
    w = _mk((cin, cout, k), 1, 1.0 / (cin * k / s) ** 0.5)
    b = _mk((cout,), 2, 0.1)
    x = _mk((B, cin, L), 3)
    conv = Conv1dHip(cin, cout, k, stride=s, padding=pad, transposed=True)
    conv.set_weight(w, b)
    y = conv(x.cuda(), in_slope=0.1).cpu()
    ref = F.conv_transpose1d(F.leaky_relu(x, 0.1), w, b, stride=s, padding=pad)
    assert y.shape == ref.shape
    assert float((y - ref).abs().max()) < TOL


def test_conv1d_errors():
    from ttscube_amd.hip_layers import Conv1dHip
    from ttscube_amd._lib import TTSCError
    conv = Conv1dHip(8, 8, 3, padding=1)
    with pytest.raises(TTSCError):
        conv(torch.zeros(1, 8, 10).cuda())  # weights not set
    with pytest.raises(TTSCError):
        conv.set_weight(torch.zeros(8, 7, 3))
    conv.set_weight(torch.zeros(8, 8, 3))
    with pytest.raises(TTSCError):
        conv(torch.zeros(1, 8, 10))  # CPU tensor: no CPU path
    with pytest.raises(TTSCError):
        Conv1dHip(8, 8, 3, stride=2)


# ---- split-precision path (fp16 hi/lo halves, three fp16 MFMAs per product, fp32 accumulation) ----------------------
F16X3_TOL = 1e-5  # ~2^-21 relative per product on O(1) outputs


@pytest.mark.parametrize('cin,cout,k,d,L,B', [
    (32, 32, 3, 1, 700, 2), (32, 32, 11, 5, 1030, 2), (64, 64, 7, 3, 513, 1), (128, 128, 11, 1, 300, 2),
    (256, 256, 3, 5, 97, 1), (80, 512, 7, 1, 50, 2), (32, 1, 7, 1, 999, 2), (20, 20, 7, 1, 65, 1), (8, 4, 11, 5, 9, 2),
])
def test_conv1d_f16x3_matches_torch(cin, cout, k, d, L, B):
    from ttscube_amd.hip_layers import Conv1dHip
    pad = d * (k - 1) // 2
    w = _mk((cout, cin, k), 1, 1.0 / (cin * k) ** 0.5)
    b = _mk((cout,), 2, 0.1)
    x = _mk((B, cin, L), 3)
    conv = Conv1dHip(cin, cout, k, padding=pad, dilation=d).set_precision('f16x3')
    conv.set_weight(w, b)
    y = conv(x.cuda(), in_slope=0.1).cpu()
    ref = F.conv1d(F.leaky_relu(x, 0.1), w, b, padding=pad, dilation=d)
    assert y.shape == ref.shape
    assert float((y - ref).abs().max()) < F16X3_TOL
    # switching back re-packs from the host copy and restores the exact path
    conv.set_precision('fp32')
    y2 = conv(x.cuda(), in_slope=0.1).cpu()
    assert float((y2 - ref).abs().max()) < TOL


@pytest.mark.parametrize('cin,cout,k,s,L,B', [(512, 256, 16, 5, 50, 2), (128, 64, 4, 4, 304, 2), (64, 32, 4, 4, 1216, 1), (16, 8, 3, 2, 5, 1)])
def test_conv_transpose1d_f16x3_matches_torch(cin, cout, k, s, L, B):
    from ttscube_amd.hip_layers import Conv1dHip
    pad = (k - s) // 2
    w = _mk((cin, cout, k), 1, 1.0 / (cin * k / s) ** 0.5)
    b = _mk((cout,), 2, 0.1)
    x = _mk((B, cin, L), 3)
    conv = Conv1dHip(cin, cout, k, stride=s, padding=pad, transposed=True).set_precision('f16x3')
    conv.set_weight(w, b)
    y = conv(x.cuda(), in_slope=0.1).cpu()
    ref = F.conv_transpose1d(F.leaky_relu(x, 0.1), w, b, stride=s, padding=pad)
    assert float((y - ref).abs().max()) < F16X3_TOL


def test_conv1d_f16x3_small_and_large_magnitudes():
    """weights of very different scale (power-of-two pre-scaling) and activations from 1e-4 to 1e3 keep ~fp32 accuracy"""
    from ttscube_amd.hip_layers import Conv1dHip
    cin = cout = 64
    for wscale, xscale in [(1e-3, 1.0), (30.0, 1e-2), (0.02, 500.0)]:
        w = _mk((cout, cin, 3), 1, wscale)
        x = _mk((1, cin, 300), 3, xscale)
        conv = Conv1dHip(cin, cout, 3, padding=1).set_precision('f16x3')
        conv.set_weight(w, None)
        y = conv(x.cuda()).cpu()
        ref = F.conv1d(x, w, None, padding=1)
        rel = float((y - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
        assert rel < 2e-6, (wscale, xscale, rel)


@pytest.mark.parametrize('k,d,L,B', [(3, 1, 1500, 2), (3, 5, 479, 1), (7, 3, 481, 2), (11, 5, 2000, 1), (11, 1, 7, 2), (7, 1, 960, 1)])
def test_fused_residual_pair_matches_torch(k, d, L, B):
    """respair32 kernel (through ttsc_respair_forward): y = x + conv2(lrelu(conv1_d(lrelu(x)))) [+ running sum]"""
    import ctypes as C
    from ttscube_amd import _lib
    from ttscube_amd.hip_layers import Conv1dHip
    c1 = Conv1dHip(32, 32, k, padding=d * (k - 1) // 2, dilation=d).set_precision('f16x3')
    c2 = Conv1dHip(32, 32, k, padding=(k - 1) // 2).set_precision('f16x3')
    w1, b1 = _mk((32, 32, k), 1, 1.0 / (32 * k) ** 0.5), _mk((32,), 2, 0.1)
    w2, b2 = _mk((32, 32, k), 3, 1.0 / (32 * k) ** 0.5), _mk((32,), 4, 0.1)
    c1.set_weight(w1, b1)
    c2.set_weight(w2, b2)
    L_ = _lib.lib()
    assert L_.ttsc_respair_supported(c1._h, c2._h) == 1
    x = _mk((B, 32, L), 5)
    s0 = _mk((B, 32, L), 6)
    ref = x + F.conv1d(F.leaky_relu(F.conv1d(F.leaky_relu(x, 0.1), w1, b1, padding=d * (k - 1) // 2, dilation=d), 0.1), w2, b2,
                       padding=(k - 1) // 2)
    xd = x.cuda()
    for accumulate in (0, 1):
        y = s0.clone().cuda()
        _lib.check(L_.ttsc_respair_forward(c1._h, c2._h, _lib.dev_ptr(xd), B, L, _lib.dev_ptr(y), accumulate, None,
                                           _lib.current_stream()), 'ttsc_respair_forward')
        want = ref + s0 if accumulate else ref
        assert float((y.cpu() - want).abs().max()) < 2e-5, (accumulate,)
    # ragged: utterance 0 is shorter; its valid part equals the utterance run alone
    if B > 1 and L > 300:
        lens = torch.tensor([L - 257, L], dtype=torch.int32).cuda()
        y = torch.zeros_like(xd)
        _lib.check(L_.ttsc_respair_forward(c1._h, c2._h, _lib.dev_ptr(xd), B, L, _lib.dev_ptr(y), 0, _lib.dev_ptr(lens),
                                           _lib.current_stream()), 'ttsc_respair_forward')
        n = L - 257
        xs = x[:1, :, :n]
        solo = xs + F.conv1d(F.leaky_relu(F.conv1d(F.leaky_relu(xs, 0.1), w1, b1, padding=d * (k - 1) // 2, dilation=d), 0.1), w2,
                             b2, padding=(k - 1) // 2)
        assert float((y[:1, :, :n].cpu() - solo).abs().max()) < 2e-5
    # not eligible: fp32 precision / other channel counts
    c1.set_precision('fp32')
    assert L_.ttsc_respair_supported(c1._h, c2._h) == 0


def _chain_layers(C, k, dils, seed=0):
    from ttscube_amd.hip_layers import Conv1dHip
    c1s, c2s, ws = [], [], []
    for m, d in enumerate(dils):
        c1 = Conv1dHip(C, C, k, padding=d * (k - 1) // 2, dilation=d).set_precision('f16x3')
        c2 = Conv1dHip(C, C, k, padding=(k - 1) // 2).set_precision('f16x3')
        w1, b1 = _mk((C, C, k), seed + 10 * m + 1, 1.0 / (C * k) ** 0.5), _mk((C,), seed + 10 * m + 2, 0.1)
        w2, b2 = _mk((C, C, k), seed + 10 * m + 3, 1.0 / (C * k) ** 0.5), _mk((C,), seed + 10 * m + 4, 0.1)
        c1.set_weight(w1, b1)
        c2.set_weight(w2, b2)
        c1s.append(c1)
        c2s.append(c2)
        ws.append((w1, b1, w2, b2, d))
    return c1s, c2s, ws


def _chain_ref(x, ws, k):
    for w1, b1, w2, b2, d in ws:
        xt = F.conv1d(F.leaky_relu(x, 0.1), w1, b1, padding=d * (k - 1) // 2, dilation=d)
        x = x + F.conv1d(F.leaky_relu(xt, 0.1), w2, b2, padding=(k - 1) // 2)
    return x


def _chain_call(c1s, c2s, xd, y, accumulate, lens, shape):
    import ctypes as C
    from ttscube_amd import _lib
    n = len(c1s)
    a1 = (C.c_void_p * n)(*[c._h for c in c1s])
    a2 = (C.c_void_p * n)(*[c._h for c in c2s])
    L_ = _lib.lib()
    assert L_.ttsc_rbchain_supported(a1, a2, n) == 1
    B, _, L = xd.shape
    _lib.check(L_.ttsc_rbchain_forward(a1, a2, n, _lib.dev_ptr(xd), B, L, _lib.dev_ptr(y), accumulate,
                                       _lib.dev_ptr(lens) if lens is not None else None, shape, _lib.current_stream()),
               'ttsc_rbchain_forward')


@pytest.mark.parametrize('C,k,dils,L,B,shape', [
    (32, 3, (1, 3, 5), 1500, 2, 0), (32, 3, (1, 3, 5), 2311, 1, 1), (32, 7, (1, 3, 5), 1111, 2, 0), (32, 7, (1, 3, 5), 2048, 1, 1),
    (32, 11, (1, 3, 5), 1999, 2, 0), (32, 11, (1, 3, 5), 3000, 1, 1), (32, 11, (5,), 7, 2, -1), (32, 3, (1, 3), 392, 1, -1),
    (64, 3, (1, 3, 5), 900, 2, 0), (64, 7, (1, 3, 5), 777, 1, 0), (64, 7, (1, 3, 5), 1300, 2, 1), (64, 11, (1, 3, 5), 1100, 1, 1),
    (64, 11, (3,), 600, 2, 0), (64, 3, (5,), 257, 1, 1),
    (128, 3, (1, 3, 5), 1000, 2, -1), (128, 3, (1, 3, 5), 233, 1, -1), (128, 3, (5,), 470, 2, -1),   # 2 x 4 wave grid, three 1-step weight slots
    # interleaved columns (round 4): vector path (L, tile start and lengths multiples of the vector width) and its column-wise fallback
    (32, 3, (1, 3, 5), 2048, 2, 10), (32, 7, (1, 3, 5), 4000, 2, 11), (32, 11, (1, 3, 5), 4000, 2, 12), (32, 11, (1, 3, 5), 2999, 2, 11),
    (64, 3, (1, 3, 5), 1200, 2, 10), (64, 11, (1, 3, 5), 1600, 2, 11), (64, 7, (1, 3, 5), 1501, 2, 11), (128, 3, (1, 3, 5), 1000, 2, 11),
    (32, 11, (2, 4), 1200, 2, 11),   # dilations outside {1, 3, 5}: falls back to the plain layout
    (32, 7, (1, 3, 5), 2048, 2, 2), (32, 11, (1, 3, 5), 2048, 2, 3), (32, 7, (1, 3, 5), 2048, 2, 4),   # longer weight groups, 768-column tiles
])
def test_fused_resblock_chain_matches_torch(C, k, dils, L, B, shape):
    """rbchain_f16x3_kernel (resblock.hip) through ttsc_rbchain_forward: the whole ResBlock1
    x <- x + conv2(lrelu(conv1_d(lrelu(x)))) for every dilation, residual stream in registers, activations in LDS."""
    c1s, c2s, ws = _chain_layers(C, k, dils, seed=100 * k + C)
    x = _mk((B, C, L), 5)
    s0 = _mk((B, C, L), 6)
    ref = _chain_ref(x, ws, k)
    xd = x.cuda()
    for accumulate in (0, 1):
        y = s0.clone().cuda()
        _chain_call(c1s, c2s, xd, y, accumulate, None, shape)
        want = ref + s0 if accumulate else ref
        err = float((y.cpu() - want).abs().max())
        assert err < 3e-5, (accumulate, err)
    if B > 1 and L > 600:
        # ragged: utterance 0 is shorter; its valid part equals the utterance run alone, the tail is left untouched
        n = L - 333 if L % 4 else L - 332   # (a multiple of 4 keeps the interleaved kernels on their vector path)
        lens = torch.tensor([n] + [L] * (B - 1), dtype=torch.int32).cuda()
        y = torch.full_like(xd, 7.0)
        _chain_call(c1s, c2s, xd, y, 0, lens, shape)
        solo = _chain_ref(x[:1, :, :n], ws, k)
        assert float((y[:1, :, :n].cpu() - solo).abs().max()) < 3e-5
        assert float((y[1:].cpu() - ref[1:]).abs().max()) < 3e-5


def test_interleaved_chain_is_bit_identical_to_the_plain_layout(monkeypatch):
    """the interleaved-column kernels only permute which lane owns which column: same k-order, same sums"""
    for C, k, L in ((32, 11, 4000), (32, 3, 2311), (64, 7, 1600), (128, 3, 1000)):
        c1s, c2s, ws = _chain_layers(C, k, (1, 3, 5), seed=7 * k + C)
        xd = _mk((2, C, L), 9).cuda()
        outs = []
        for il in ('0', '1'):
            monkeypatch.setenv('TTSC_CHAIN_IL', il)
            y = torch.zeros_like(xd)
            _chain_call(c1s, c2s, xd, y, 0, None, 1)
            outs.append(y)
        assert torch.equal(outs[0], outs[1]), (C, k, L, float((outs[0] - outs[1]).abs().max()))


def test_fused_resblock_chain_not_eligible():
    import ctypes as C
    from ttscube_amd import _lib
    c1s, c2s, _ = _chain_layers(32, 7, (1, 3))
    c1s[1].set_precision('fp32')
    a1 = (C.c_void_p * 2)(*[c._h for c in c1s])
    a2 = (C.c_void_p * 2)(*[c._h for c in c2s])
    assert _lib.lib().ttsc_rbchain_supported(a1, a2, 2) == 0
    c1s, c2s, _ = _chain_layers(128, 7, (1,))   # 128 channels: only K = 3 fits the LDS image
    a1 = (C.c_void_p * 1)(c1s[0]._h)
    a2 = (C.c_void_p * 1)(c2s[0]._h)
    assert _lib.lib().ttsc_rbchain_supported(a1, a2, 1) == 0


@pytest.mark.parametrize('C,k,d,L,B', [(256, 3, 1, 300, 2), (256, 3, 5, 129, 1), (256, 7, 3, 257, 2), (256, 7, 5, 128, 1), (256, 11, 1, 400, 1),
                                       (256, 11, 5, 131, 2), (128, 3, 3, 700, 2), (128, 3, 5, 256, 1), (128, 7, 1, 513, 1), (128, 7, 5, 300, 2),
                                       (128, 11, 3, 255, 2), (128, 11, 5, 1000, 1), (128, 7, 3, 5, 2)])
def test_conv1d_f16x3_wide_tile_kernel(C, k, d, L, B, monkeypatch):
    """conv_f16x3_wide_kernel (square 128/256-channel layers; chosen by machine fill in production, forced here):
    prologue leaky-relu, residual, running sum, ragged lengths — against torch, and bit-identical batch independence."""
    from ttscube_amd.hip_layers import Conv1dHip
    monkeypatch.setenv('TTSC_CONV_WIDE', '2')
    pad = d * (k - 1) // 2
    w = _mk((C, C, k), 1, 1.0 / (C * k) ** 0.5)
    b = _mk((C,), 2, 0.1)
    x = _mk((B, C, L), 3)
    r = _mk((B, C, L), 4)
    s0 = _mk((B, C, L), 5)
    conv = Conv1dHip(C, C, k, padding=pad, dilation=d).set_precision('f16x3')
    conv.set_weight(w, b)
    out = s0.clone().cuda()
    conv(x.cuda(), resid=r.cuda(), out=out, in_scale=1.0 / 3.0, in_slope=0.1, accumulate=True)
    ref = s0 + F.conv1d(F.leaky_relu(x / 3.0, 0.1), w, b, padding=pad, dilation=d) + r
    # residual + running sum + bias are the INITIAL value of the accumulators (round 5: their loads leave in the prologue, all at once, instead
    # of as 32 dependent round trips in the epilogue), so the K * C / 16 accumulation steps round at the magnitude of the whole sum (here up to
    # ~8: N(0,1) residual + N(0,1) running sum + the convolution) instead of the convolution's alone: a few ulp of the RESULT — the bound is relative
    assert float((out.cpu() - ref).abs().max()) < max(F16X3_TOL, 3e-6 * float(ref.abs().max()))
    y = conv(x.cuda(), in_slope=0.1)
    ref = F.conv1d(F.leaky_relu(x, 0.1), w, b, padding=pad, dilation=d)
    assert float((y.cpu() - ref).abs().max()) < F16X3_TOL
    # the old 64-row kernel computes the same chains in another order: results agree to rounding
    monkeypatch.setenv('TTSC_CONV_WIDE', '0')
    y0 = conv(x.cuda(), in_slope=0.1)
    assert float((y0 - y).abs().max()) < 1e-5
    monkeypatch.setenv('TTSC_CONV_WIDE', '2')
    if B > 1 and L > 100:
        import ctypes as C_
        from ttscube_amd import _lib
        n = L - 77
        lens = torch.tensor([n] + [L] * (B - 1), dtype=torch.int32).cuda()
        yr = torch.zeros(B, C, L, device='cuda')
        ep = _lib.Conv1dEpilogue(1.0, 0.1, 1.0, _lib.ACT_NONE, 0, None, 1.0)
        xd = x.cuda()
        _lib.check(_lib.lib().ttsc_conv1d_forward_ragged(conv._h, _lib.dev_ptr(xd), B, L, _lib.dev_ptr(yr), None, C_.byref(ep),
                                                         _lib.dev_ptr(lens), _lib.dev_ptr(lens), _lib.current_stream()), 'ragged')
        solo = conv(x[:1, :, :n].contiguous().cuda(), in_slope=0.1)
        assert torch.equal(yr[:1, :, :n], solo)           # ragged == the utterance run alone, bit for bit
        assert torch.equal(yr[1:], y[1:])


@pytest.mark.parametrize('cin,k,L,B,prec', [(32, 7, 5003, 2, 'f16x3'), (32, 7, 1024, 1, 'fp32'), (64, 11, 2050, 1, 'f16x3'), (20, 3, 7, 2, 'fp32')])
def test_single_output_channel_conv(cin, k, L, B, prec):
    """conv_cout1_kernel (HiFi-GAN conv_post, 32 -> 1, tanh): prologue scale + leaky-relu, residual, running sum, ragged."""
    import ctypes as C_
    from ttscube_amd import _lib
    from ttscube_amd.hip_layers import Conv1dHip
    pad = (k - 1) // 2
    w, b = _mk((1, cin, k), 1, 1.0 / (cin * k) ** 0.5), _mk((1,), 2, 0.1)
    x, r, s0 = _mk((B, cin, L), 3), _mk((B, 1, L), 4), _mk((B, 1, L), 5)
    conv = Conv1dHip(cin, 1, k, padding=pad).set_precision(prec)
    conv.set_weight(w, b)
    out = s0.clone().cuda()
    conv(x.cuda(), resid=r.cuda(), out=out, in_scale=1.0 / 3.0, in_slope=0.01, out_scale=0.5, act='tanh', accumulate=True)
    ref = s0 + torch.tanh((F.conv1d(F.leaky_relu(x / 3.0, 0.01), w, b, padding=pad) + r) * 0.5)
    assert float((out.cpu() - ref).abs().max()) < TOL
    if B > 1:
        n = L - 3
        lens = torch.tensor([n] + [L] * (B - 1), dtype=torch.int32).cuda()
        yr = torch.zeros(B, 1, L, device='cuda')
        ep = _lib.Conv1dEpilogue(1.0, 0.1, 1.0, _lib.ACT_NONE, 0, None, 1.0)
        xd = x.cuda()
        _lib.check(_lib.lib().ttsc_conv1d_forward_ragged(conv._h, _lib.dev_ptr(xd), B, L, _lib.dev_ptr(yr), None, C_.byref(ep),
                                                         _lib.dev_ptr(lens), _lib.dev_ptr(lens), _lib.current_stream()), 'ragged')
        solo = conv(x[:1, :, :n].contiguous().cuda(), in_slope=0.1)
        assert torch.equal(yr[:1, :, :n], solo)


@pytest.mark.parametrize('cin,cout,L,B', [(64, 32, 8192, 8), (128, 64, 300, 2), (16, 8, 33, 1)])
def test_conv_transpose_k4s4_interleaved_rows(cin, cout, L, B):
    """ConvTranspose1d(kernel_size = stride = 4) in f16x3: GEMM rows interleaved (channel, phase) so that a lane stores four
    consecutive samples (epilogue_tile_v4) — with residual, running sum and activation, at a size that takes the big tile."""
    from ttscube_amd.hip_layers import Conv1dHip
    w, b = _mk((cin, cout, 4), 1, 1.0 / cin ** 0.5), _mk((cout,), 2, 0.1)
    x = _mk((B, cin, L), 3)
    conv = Conv1dHip(cin, cout, 4, stride=4, padding=0, transposed=True).set_precision('f16x3')
    conv.set_weight(w, b)
    ref = F.conv_transpose1d(F.leaky_relu(x / 3.0, 0.1), w, b, stride=4)
    xd = x.cuda()
    y = conv(xd, in_scale=1.0 / 3.0, in_slope=0.1)
    assert y.shape == ref.shape and float((y.cpu() - ref).abs().max()) < F16X3_TOL
    if L <= 300:
        r, s0 = _mk(tuple(ref.shape), 4), _mk(tuple(ref.shape), 5)
        out = s0.clone().cuda()
        conv(xd, resid=r.cuda(), out=out, in_scale=1.0 / 3.0, in_slope=0.1, out_scale=0.5, act='tanh', accumulate=True)
        assert float((out.cpu() - (s0 + torch.tanh((ref + r) * 0.5))).abs().max()) < F16X3_TOL


@pytest.mark.parametrize('xscale,act_scale', [(1e-4, 2.0 ** 22), (1e-7, 2.0 ** 32), (3e4, 2.0 ** -6), (1.0, 2.0 ** 9)])
def test_conv1d_f16x3_activation_prescale(xscale, act_scale):
    """activations far outside fp16's comfortable range keep ~fp32 accuracy once the layer's power-of-two input pre-scale is
    set (ttsc_conv1d_set_activation_scale): 1e-4 and 1e-7 (hi/lo halves subnormal without it), 3e4 (a 64-channel sum
    overflows fp16 without it)."""
    from ttscube_amd.hip_layers import Conv1dHip
    cin = cout = 64
    w = _mk((cout, cin, 7), 1, 1.0 / (cin * 7) ** 0.5)
    b = _mk((cout,), 2, 0.1 * xscale)
    x = _mk((2, cin, 300), 3, xscale)
    conv = Conv1dHip(cin, cout, 7, padding=3).set_precision('f16x3')
    conv.set_weight(w, b)
    ref = F.conv1d(F.leaky_relu(x, 0.1), w, b, padding=3)
    conv.set_activation_scale(act_scale)
    y = conv(x.cuda(), in_slope=0.1).cpu()
    rel = float((y - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
    assert rel < 2e-6, (xscale, rel)
    with pytest.raises(Exception):
        conv.set_activation_scale(3.0)       # not a power of two


@pytest.mark.parametrize('cin,cout,k,s,L,B', [(512, 256, 16, 5, 300, 2), (512, 256, 16, 5, 37, 1), (256, 128, 16, 3, 600, 2), (256, 128, 16, 3, 259, 1)])
def test_conv_transpose1d_tall_tile_kernel(cin, cout, k, s, L, B, monkeypatch):
    """conv_f16x3_tall_kernel (256 / 128 virtual rows per workgroup; the first two upsamplers of HiFi-GAN V1), forced on for problems too
    small to pick it by itself (TTSC_CONV_WIDE=2), against torch and against the general kernel (TTSC_CONV_TALL=0 is read once per process, so
    the comparison kernel is the fp32 path)"""
    from ttscube_amd.hip_layers import Conv1dHip
    monkeypatch.setenv('TTSC_CONV_WIDE', '2')
    pad = (k - s) // 2
    w = _mk((cin, cout, k), 11, 1.0 / (cin * k / s) ** 0.5)
    b = _mk((cout,), 12, 0.1)
    x = _mk((B, cin, L), 13)
    conv = Conv1dHip(cin, cout, k, stride=s, padding=pad, transposed=True).set_precision('f16x3')
    conv.set_weight(w, b)
    y = conv(x.cuda(), in_slope=0.1).cpu()
    ref = F.conv_transpose1d(F.leaky_relu(x, 0.1), w, b, stride=s, padding=pad)
    assert y.shape == ref.shape and float((y - ref).abs().max()) < F16X3_TOL
    conv.set_precision('fp32')
    y32 = conv(x.cuda(), in_slope=0.1).cpu()
    assert float((y - y32).abs().max()) < F16X3_TOL


@pytest.mark.parametrize('kind', ['conv256', 'conv128_resid', 'ups_tall', 'ups_k4s4'])
def test_pitched_forward_equals_the_dense_call_on_the_valid_region(kind, monkeypatch):
    """ttsc_conv1d_forward_pitched (round 5: the generator's intermediate tensors have 128-byte row pitches): the same layer on tensors whose rows are
    padded to a multiple of 32 floats, real lengths in the per-utterance tables, must give the dense call's bits on every valid position — stride-1
    layers on the wide kernel (input and output pitch = the padded length) and transposed layers (explicit output pitch) alike."""
    import ctypes as C_
    from ttscube_amd import _lib
    from ttscube_amd.hip_layers import Conv1dHip
    monkeypatch.setenv('TTSC_CONV_WIDE', '2')
    B = 2
    if kind == 'conv256':
        Cin, Cout, L, conv = 256, 256, 301, Conv1dHip(256, 256, 7, padding=9, dilation=3)
        w, Lo = _mk((256, 256, 7), 1, 1.0 / (256 * 7) ** 0.5), 301
    elif kind == 'conv128_resid':
        Cin, Cout, L, conv = 128, 128, 517, Conv1dHip(128, 128, 3, padding=1)
        w, Lo = _mk((128, 128, 3), 2, 1.0 / (128 * 3) ** 0.5), 517
    elif kind == 'ups_tall':
        Cin, Cout, L, conv = 512, 256, 101, Conv1dHip(512, 256, 16, stride=5, padding=5, transposed=True)
        w, Lo = _mk((512, 256, 16), 3, 1.0 / (512 * 16 / 5) ** 0.5), 101 * 5 + 1
    else:
        Cin, Cout, L, conv = 128, 64, 203, Conv1dHip(128, 64, 4, stride=4, padding=0, transposed=True)
        w, Lo = _mk((128, 64, 4), 4, 1.0 / (128 * 4 / 4) ** 0.5), 203 * 4
    conv.set_precision('f16x3')
    conv.set_weight(w, _mk((Cout,), 5, 0.1))
    assert conv.out_len(L) == Lo
    x = _mk((B, Cin, L), 6).cuda()
    r = _mk((B, Cout, Lo), 7).cuda() if kind == 'conv128_resid' else None
    dense = conv(x, resid=r, in_slope=0.1)
    Pi, Po = (L + 31) // 32 * 32, (Lo + 31) // 32 * 32
    xp = torch.full((B, Cin, Pi), float('nan'), device='cuda')      # whatever sits in the padding must never be read as data
    xp[:, :, :L] = x
    rp = None
    if r is not None:
        rp = torch.zeros((B, Cout, Po), device='cuda')
        rp[:, :, :Lo] = r
    yp = torch.full((B, Cout, Po), 7.0, device='cuda')
    il = torch.full((B,), L, dtype=torch.int32, device='cuda')
    ol = torch.full((B,), Lo, dtype=torch.int32, device='cuda')
    ep = _lib.Conv1dEpilogue(1.0, 0.1, 1.0, _lib.ACT_NONE, 0, None, 1.0)
    _lib.check(_lib.lib().ttsc_conv1d_forward_pitched(conv._h, _lib.dev_ptr(xp), B, Pi, _lib.dev_ptr(yp), _lib.dev_ptr(rp) if rp is not None else None,
                                                      C_.byref(ep), _lib.dev_ptr(il), _lib.dev_ptr(ol), Po, _lib.current_stream()), 'pitched')
    assert torch.equal(yp[:, :, :Lo], dense)
    assert bool(torch.isfinite(yp[:, :, :Lo]).all())
    # an output pitch without length tables is refused (the pitch alone cannot bound the rows)
    with pytest.raises(_lib.TTSCError):
        _lib.check(_lib.lib().ttsc_conv1d_forward_pitched(conv._h, _lib.dev_ptr(xp), B, Pi, _lib.dev_ptr(yp), None, C_.byref(ep), None, None, Po,
                                                          _lib.current_stream()), 'pitched')


def test_split_status_of_one_stream_is_clean_and_cheap():
    """ttsc_split_status_stream: the verdict of the split recurrences launched on ONE stream (what the Cubegan step asks before each side's update);
    an idle / unknown stream reports nothing"""
    from ttscube_amd import _lib
    s = torch.cuda.Stream()
    assert int(_lib.lib().ttsc_split_status_stream(s.cuda_stream)) == 0
    _lib.check_split_status('test', stream=s.cuda_stream)
