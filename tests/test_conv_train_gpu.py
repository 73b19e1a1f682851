"""Split-precision training convolution (csrc/conv_train.hip, C ABI `ttsc_conv_train`) against torch's float64 conv1d on the same inputs:
forward with leaky-relu prologue / bias / residual, data gradient with the gate epilogue, sequences folded into the tile columns
(lengths that do not divide a tile, batches that end inside one), and inputs 1e-6 .. 1e4 in magnitude — the range words are measured on
the device per launch, nothing is calibrated.  Tolerance: 2e-6 of the output's largest magnitude (22-bit operands, fp32 accumulation)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _run(x, w, b, resid, gate, padding, dilation, flip, groups=1, **kw):
    from ttscube_amd.hifigan.autograd import _conv_split
    if flip:
        Cin, Cout, K = w.shape[0], w.shape[1] * groups, w.shape[2]
    else:
        Cout, Cin, K = w.shape[0], w.shape[1] * groups, w.shape[2]
    return _conv_split(x, w, b, resid, gate, Cin, Cout, K, padding, dilation, flip, groups=groups, **kw)


def _ref(x, w, b, resid, gate, padding, dilation, flip, in_scale=1.0, in_slope=1.0, out_scale=1.0, gate_slope=1.0):
    x, w = x.double(), w.double()
    if flip:
        w = w.permute(1, 0, 2).flip(2)
    a = F.leaky_relu(x * in_scale, in_slope) if in_slope != 1.0 else x * in_scale
    y = F.conv1d(a, w, b.double() if b is not None else None, padding=padding, dilation=dilation)
    if gate is not None:
        y = y * torch.where(gate.double() > 0, 1.0, gate_slope)
    if resid is not None:
        y = y + resid.double()
    return y * out_scale


CASES = [  # B, Cin, Cout, K, L, padding, dilation
    (3, 48, 80, 5, 77, 2, 1),
    (4, 64, 96, 5, 210, 14, 7),      # MPD period 7: (5, 1) kernel over the flat signal
    (2, 96, 64, 2, 333, 0, 3),       # de-interleaved strided layer: 2 taps, no padding
    (5, 32, 32, 3, 50, 1, 1),
    (2, 128, 128, 11, 300, 25, 5),   # generator ResBlock, widest receptive field
    (1, 16, 33, 7, 1000, 3, 1),      # ragged channel counts
    (7, 1024, 64, 5, 22, 4, 2),      # deep discriminator layer: many channels, a sliver of positions
    (3, 1, 32, 5, 500, 2, 1),        # discriminator input layer: one input channel
    (2, 3, 32, 2, 400, 0, 7),        # ... after the stride de-interleave
    (4, 1024, 1, 3, 150, 2, 2),      # conv_post: one output channel
    (2, 1, 128, 15, 3000, 7, 1),     # MSD input layer
]


@pytest.mark.parametrize('B,Cin,Cout,K,L,pad,d', CASES)
@pytest.mark.parametrize('mag', [1.0, 1e-6, 1e4])
def test_forward_matches_float64(B, Cin, Cout, K, L, pad, d, mag):
    g = torch.Generator().manual_seed(B * 1000 + Cin + K)
    x = (torch.randn(B, Cin, L, generator=g) * mag).cuda()
    w = (torch.randn(Cout, Cin, K, generator=g) / (Cin * K) ** 0.5).cuda()
    b = (torch.randn(Cout, generator=g) * mag).cuda()
    Lout = L + 2 * pad - d * (K - 1)
    r = (torch.randn(B, Cout, Lout, generator=g) * mag).cuda()
    y = _run(x, w, b, r, None, pad, d, 0, in_scale=0.5, in_slope=0.1)
    ref = _ref(x, w, b, r, None, pad, d, 0, in_scale=0.5, in_slope=0.1)
    assert y.shape == ref.shape
    assert float((y.double() - ref).abs().max()) <= 2e-6 * float(ref.abs().max())


@pytest.mark.parametrize('B,Cin,Cout,K,L,pad,d', CASES)
def test_data_gradient_matches_autograd(B, Cin, Cout, K, L, pad, d):
    """flip = 1 on the forward weight + gate epilogue == d/dx of conv(leaky_relu(sc * x))"""
    g = torch.Generator().manual_seed(B + Cin * 7 + K)
    x = torch.randn(B, Cin, L, generator=g, dtype=torch.float64).cuda().requires_grad_(True)
    w = (torch.randn(Cout, Cin, K, generator=g, dtype=torch.float64) / (Cin * K) ** 0.5).cuda()
    sc, sl = 0.7, 0.1
    y = F.conv1d(F.leaky_relu(x * sc, sl), w, padding=pad, dilation=d)
    dy = torch.randn(y.shape, generator=g, dtype=torch.float64).cuda() * 1e-5     # gradients are small numbers
    (dx_ref,) = torch.autograd.grad(y, x, dy)
    pd = d * (K - 1) - pad
    dx = _run(dy.float(), w.float().contiguous(), None, None, x.detach().float(), pd, d, 1, out_scale=sc, gate_slope=sl)
    assert dx.shape == dx_ref.shape
    assert float((dx.double() - dx_ref).abs().max()) <= 2e-6 * float(dx_ref.abs().max())


def test_zero_and_non_finite_inputs_are_visible():
    x = torch.zeros(2, 32, 40).cuda()
    w = torch.randn(32, 32, 3).cuda()
    assert float(_run(x, w, None, None, None, 1, 1, 0).abs().max()) == 0.0
    x[1, 3, 7] = float('inf')
    y = _run(x, w, None, None, None, 1, 1, 0)
    assert not bool(torch.isfinite(y[1]).all()) and bool(torch.isfinite(y[0]).all())
    x[1, 3, 7] = float('nan')
    assert bool(torch.isnan(_run(x, w, None, None, None, 1, 1, 0)[1]).any())


def test_unsupported_shapes_are_refused():
    from ttscube_amd import _lib
    assert _lib.lib().ttsc_conv_train_supported(1, 32, 5, 1, 1)           # discriminator input layer: taken (one padded channel chunk)
    assert _lib.lib().ttsc_conv_train_supported(1024, 1, 3, 1, 1)         # conv_post: taken (one padded row tile)
    assert not _lib.lib().ttsc_conv_train_supported(64, 64, 41, 2, 1)     # receptive field beyond the staged window
    assert not _lib.lib().ttsc_conv_train_supported(64, 96, 7, 1, 2)      # 48 output channels per group do not tile into 32-row blocks
    with pytest.raises(_lib.TTSCError):
        _run(torch.zeros(1, 64, 64).cuda(), torch.zeros(64, 64, 41).cuda(), None, None, None, 40, 2, 0)


WG_CASES = [  # N, A, Bc, LP, LQ, J, base, step
    (3, 128, 64, 100, 100, 5, -2, 1),
    (4, 192, 96, 210, 210, 5, -14, 7),      # MPD period 7
    (2, 64, 96, 331, 334, 2, 0, 3),         # de-interleaved strided layer (no padding: LP = LQ - step)
    (2, 128, 128, 300, 300, 11, -25, 5),    # generator ResBlock k = 11, dilation 5: three launches of <= 5 taps
    (2, 160, 64, 80, 77, 4, 0, -1),         # ConvTranspose1d on the de-interleaved output gradient: taps run backwards
    (5, 100, 40, 65, 65, 3, -1, 1),         # ragged tile edges, one position into the second chunk
    (6, 1024, 64, 22, 22, 5, -4, 2),        # deep discriminator layer
]


@pytest.mark.parametrize('N,A,Bc,LP,LQ,J,base,step', WG_CASES)
@pytest.mark.parametrize('mag', [1.0, 1e-6])
def test_split_weight_gradient_matches_float64(N, A, Bc, LP, LQ, J, base, step, mag):
    from ttscube_amd import _lib
    from ttscube_amd.hifigan import autograd as AG
    g = torch.Generator().manual_seed(N * 100 + A + J)
    P = (torch.randn(N, A, LP, generator=g) * mag).cuda()
    Q = torch.randn(N, Bc, LQ, generator=g).cuda()
    sc, sl = 0.6, 0.1
    assert _lib.lib().ttsc_conv_wgrad_split_supported(A, Bc, J, step)
    G = AG._wgrad(P, Q, A, Bc, J, base, step, sc, sl)
    Qa = F.leaky_relu(Q.double() * sc, sl)
    ref = torch.zeros(A, Bc, J, dtype=torch.float64, device='cuda')
    for j in range(J):
        off = base + j * step
        Qs = torch.zeros(N, Bc, LP, dtype=torch.float64, device='cuda')
        lo, hi = max(0, -off), min(LP, LQ - off)
        if hi > lo:
            Qs[:, :, lo:hi] = Qa[:, :, lo + off:hi + off]
        ref[:, :, j] = torch.einsum('nat,nbt->ab', P.double(), Qs)
    assert G.shape == ref.shape
    assert float((G.double() - ref).abs().max()) <= 2e-6 * float(ref.abs().max())
    try:   # and the exact-fp32 kernel on the same inputs, for the record of what the split path replaces
        AG.SPLIT_TRAIN = False
        G32 = AG._wgrad(P, Q, A, Bc, J, base, step, sc, sl)
    finally:
        AG.SPLIT_TRAIN = True
    assert float((G32.double() - ref).abs().max()) <= 2e-6 * float(ref.abs().max())


GROUPED = [  # B, Cin, Cout, K, L, padding, groups   (MSD after the stride de-interleave: k = 41 -> 21 / 11 taps; and its stride-1 layer)
    (3, 256, 128, 21, 150, 0, 4),      # 64 -> 32 channels per group: one group per 32-row tile
    (2, 256, 256, 21, 97, 0, 16),      # 16 -> 16: two groups per tile, block-diagonal weights
    (2, 1024, 512, 11, 60, 0, 16),     # 64 -> 32
    (2, 2048, 1024, 11, 31, 0, 16),    # 128 -> 64: 64-row tiles
    (3, 1024, 1024, 41, 47, 20, 16),   # k = 41 at stride 1: 41 taps
    (2, 128, 256, 21, 140, 20, 4),     # the data-gradient shape of the first case (32 -> 64)
]


@pytest.mark.parametrize('B,Cin,Cout,K,L,pad,G', GROUPED)
def test_grouped_forward_and_data_gradient_match_float64(B, Cin, Cout, K, L, pad, G):
    g = torch.Generator().manual_seed(Cin + Cout + K)
    x = torch.randn(B, Cin, L, generator=g, dtype=torch.float64).cuda().requires_grad_(True)
    w = (torch.randn(Cout, Cin // G, K, generator=g, dtype=torch.float64) / (Cin // G * K) ** 0.5).cuda()
    b = torch.randn(Cout, generator=g, dtype=torch.float64).cuda()
    sc, sl = 0.8, 0.1
    ref = F.conv1d(F.leaky_relu(x * sc, sl), w, b, padding=pad, groups=G)
    y = _run(x.detach().float(), w.float().contiguous(), b.float(), None, None, pad, 1, 0, groups=G, in_scale=sc, in_slope=sl)
    assert y.shape == ref.shape
    assert float((y.double() - ref.detach()).abs().max()) <= 2e-6 * float(ref.detach().abs().max())
    dy = torch.randn(ref.shape, generator=g, dtype=torch.float64).cuda() * 1e-4
    (dx_ref,) = torch.autograd.grad(ref, x, dy)
    pd = (K - 1) - pad
    dx = _run(dy.float(), w.float().contiguous(), None, None, x.detach().float(), pd, 1, 1, groups=G, out_scale=sc, gate_slope=sl)
    assert dx.shape == dx_ref.shape
    assert float((dx.double() - dx_ref).abs().max()) <= 2e-6 * float(dx_ref.abs().max())


# ---- round 6: weight bank, range words left by the producing launch, bias gradient riding in the weight-gradient launches --------------------
def _banked_chain(seed=5, C=(16, 64, 96, 64), K=(5, 3, 7), d=(1, 3, 1)):
    """a small chain of weight-normed Conv1d layers + its WeightBank + the per-layer TrainConv handles"""
    from ttscube_amd.hifigan.autograd import TrainConv
    from ttscube_amd.hifigan.wbank import WeightBank
    torch.manual_seed(seed)
    layers, tcs = [], []
    for i, (k, dl) in enumerate(zip(K, d)):
        l = torch.nn.utils.weight_norm(torch.nn.Conv1d(C[i], C[i + 1], k, padding=dl * (k - 1) // 2, dilation=dl)).cuda()
        layers.append(l)
        tcs.append(TrainConv(C[i], C[i + 1], k, padding=dl * (k - 1) // 2, dilation=dl))
    bank = WeightBank([(l, tc.Cin, tc.Cout, tc.K, 1, 1) for l, tc in zip(layers, tcs)])
    return layers, tcs, bank


def _chain_forward(layers, tcs, bank, x):
    from ttscube_amd.hifigan.autograd import hip_conv
    ys = []
    for i, (l, tc) in enumerate(zip(layers, tcs)):
        x = hip_conv(tc, x, bank.weight(i), l.bias, in_slope=0.1 if i else 1.0)
        ys.append(x)
    return ys


def test_producing_launch_leaves_the_exact_maximum_and_the_next_launch_takes_it():
    from ttscube_amd.hifigan import wbank as WB
    layers, tcs, bank = _banked_chain()
    WB.AmaxPool.of(torch.device('cuda', 0)).reset()
    bank.prepare()
    x = torch.randn(3, 16, 333, device='cuda', requires_grad=True)
    ys = _chain_forward(layers, tcs, bank, x)
    for y in ys:
        word = WB.range_of(y)
        assert word is not None and float(word) == float(y.detach().abs().max()), 'the epilogue word must be max |y| exactly'
    ys[-1].square().mean().backward()
    # a tensor written since its producing launch (version counter moved) must not hand its word on
    with torch.no_grad():
        ys[0].mul_(1.0)
    assert WB.range_of(ys[0]) is None
    # ... and neither after the pool was reset (the word belongs to a later step)
    WB.AmaxPool.of(torch.device('cuda', 0)).reset()
    assert WB.range_of(ys[1]) is None


def test_range_word_propagation_does_not_change_a_bit():
    """the propagated word IS the reduction's result (max is exact), so forward values and every gradient are bit-identical with the switch off"""
    from ttscube_amd.hifigan import wbank as WB
    layers, tcs, bank = _banked_chain(seed=9)
    outs = []
    for prop in (True, False):
        WB.PROPAGATE = prop
        try:
            for l in layers:
                l.zero_grad()
            WB.AmaxPool.of(torch.device('cuda', 0)).reset()
            bank.prepare()
            x = torch.randn(2, 16, 500, device='cuda', generator=torch.Generator(device='cuda').manual_seed(3), requires_grad=True)
            ys = _chain_forward(layers, tcs, bank, x)
            (ys[-1].abs().mean() + ys[0].square().mean()).backward()   # (the first output gets a second gradient: accumulated by autograd)
            outs.append([ys[-1].detach().clone(), x.grad.clone()] + [p.grad.clone() for l in layers for p in (l.weight_g, l.weight_v, l.bias)])
        finally:
            WB.PROPAGATE = True
    for a, b in zip(*outs):
        assert torch.equal(a, b)


@pytest.mark.parametrize('N,A,Bc,LP,J', [(4, 64, 64, 300, 3), (2, 256, 128, 1500, 5), (3, 128, 32, 77, 7)])
def test_bias_gradient_riding_in_the_weight_gradient_launches_has_the_bits_of_bias_grad(N, A, Bc, LP, J):
    from ttscube_amd.hifigan.autograd import TrainConv, _bias_grad, _wgrad
    g = torch.Generator().manual_seed(N + A + J)
    P = torch.randn(N, A, LP, generator=g).cuda()
    Q = torch.randn(N, Bc, LP + J - 1, generator=g).cuda()
    dbl = []
    G1 = _wgrad(P, Q, A, Bc, J, 0, 1, 1.0, 1.0, db=dbl)
    G0 = _wgrad(P, Q, A, Bc, J, 0, 1, 1.0, 1.0)
    assert len(dbl) == 1 and torch.equal(G0, G1)
    ref = _bias_grad(TrainConv(Bc, A, J), P)
    assert torch.equal(dbl[0], ref)
    assert float((dbl[0].double() - P.double().sum((0, 2))).abs().max()) <= 1e-4 * float(P.abs().sum((0, 2)).max())


WG_GROUPED = [  # N, A (rows), Bg (columns per group), groups, LP, LQ, J, base, step   (MSD's grouped layers after the stride de-interleave)
    (3, 128, 64, 4, 150, 170, 21, 0, 1),      # 128 -> 128, k41 s2 g4: 32 rows x 64 columns per group
    (2, 256, 16, 16, 97, 117, 21, 0, 1),      # 128 -> 256, g16: 16 x 16 per group (half a row tile)
    (2, 512, 64, 16, 60, 70, 11, 0, 1),       # 256 -> 512, k41 s4 g16: 32 x 64
    (2, 1024, 128, 16, 31, 41, 11, 0, 1),     # 512 -> 1024: 64 x 128 per group (two row tiles, two column tiles)
    (3, 1024, 64, 16, 47, 47, 41, -20, 1),    # 1024 -> 1024, k41 s1 g16: 41 taps, padding 20
    (2, 96, 24, 4, 333, 333, 5, -4, 2),       # ragged group sizes (24 rows x 24 columns), dilation 2
]


@pytest.mark.parametrize('N,A,Bg,G,LP,LQ,J,base,step', WG_GROUPED)
@pytest.mark.parametrize('mag', [1.0, 1e-5])
def test_grouped_split_weight_gradient_matches_float64(N, A, Bg, G, LP, LQ, J, base, step, mag):
    """round 6: grouped weight gradients on the split-precision tile kernel (wgrad_f16x3_grouped_kernel) against float64, and against the exact fp32
    kernel they replace"""
    from ttscube_amd import _lib
    from ttscube_amd.hifigan import autograd as AG
    g = torch.Generator().manual_seed(N * 100 + A + J + G)
    P = (torch.randn(N, A, LP, generator=g) * mag).cuda()
    Q = torch.randn(N, G * Bg, LQ, generator=g).cuda()
    sc, sl = 0.7, 0.1
    assert _lib.lib().ttsc_conv_wgrad_split_grouped_supported(A, Bg, G, J, step)
    AG.GROUPED_SPLIT_ALL = True     # (the product only sends groups of more than 32 rows here; the entry point takes them all)
    try:
        Gw = AG._wgrad(P, Q, A, G * Bg, J, base, step, sc, sl, groups=G)
    finally:
        AG.GROUPED_SPLIT_ALL = False
    Qa = F.leaky_relu(Q.double() * sc, sl)
    Ag = A // G
    ref = torch.zeros(A, Bg, J, dtype=torch.float64, device='cuda')
    for j in range(J):
        off = base + j * step
        Qs = torch.zeros(N, G * Bg, LP, dtype=torch.float64, device='cuda')
        lo, hi = max(0, -off), min(LP, LQ - off)
        if hi > lo:
            Qs[:, :, lo:hi] = Qa[:, :, lo + off:hi + off]
        for gi in range(G):
            ref[gi * Ag:(gi + 1) * Ag, :, j] = torch.einsum('nat,nbt->ab', P[:, gi * Ag:(gi + 1) * Ag].double(), Qs[:, gi * Bg:(gi + 1) * Bg])
    assert Gw.shape == ref.shape
    assert float((Gw.double() - ref).abs().max()) <= 2e-6 * float(ref.abs().max())
    try:
        AG.GROUPED_SPLIT = False
        G32 = AG._wgrad(P, Q, A, G * Bg, J, base, step, sc, sl, groups=G)
    finally:
        AG.GROUPED_SPLIT = True
    assert float((G32.double() - ref).abs().max()) <= 2e-6 * float(ref.abs().max())
