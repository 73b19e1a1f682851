"""GPU: the HiFi-GAN discriminators on the HIP convolution kernels (hifigan/disc_hip.py) against their
torch-op formulation (hifigan/discriminators.py): strided / grouped Conv1d values and gradients, MPD and MSD outputs, feature maps
and parameter gradients."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.detach().double(), b.detach().double()
    return float((a - b).norm() / (b.norm() + 1e-300))


@pytest.mark.parametrize('Cin,Cout,K,s,p,G,N,L', [
    (1, 32, 5, 3, 2, 1, 6, 601), (32, 128, 5, 3, 2, 1, 4, 200), (512, 1024, 5, 3, 2, 1, 3, 70), (1024, 1024, 5, 1, 2, 1, 2, 30),
    (128, 128, 41, 2, 20, 4, 2, 1500), (128, 256, 41, 2, 20, 16, 2, 700), (256, 512, 41, 4, 20, 16, 2, 401), (512, 1024, 41, 4, 20, 16, 1, 200),
    (256, 256, 41, 1, 20, 16, 2, 150), (64, 128, 7, 1, 3, 2, 2, 333), (48, 48, 3, 2, 1, 3, 3, 100), (1024, 1, 3, 1, 1, 1, 2, 40), (1, 128, 15, 1, 7, 1, 2, 900),
])
def test_strided_grouped_conv_matches_torch(Cin, Cout, K, s, p, G, N, L):
    from ttscube_amd.hifigan.disc_hip import HipStridedConv
    g = torch.Generator().manual_seed(Cin + Cout + K + s)
    x = torch.randn(N, Cin, L, generator=g).cuda().requires_grad_(True)
    w = (torch.randn(Cout, Cin // G, K, generator=g) / (Cin // G * K) ** 0.5).cuda().requires_grad_(True)
    b = (torch.randn(Cout, generator=g) * 0.1).cuda().requires_grad_(True)
    for slope in (1.0, 0.1):
        ref = F.conv1d(F.leaky_relu(x, slope) if slope != 1.0 else x, w, b, stride=s, padding=p, groups=G)
        h = HipStridedConv(Cin, Cout, K, s, p, G)
        y = h(x, w, b, in_slope=slope)
        assert y.shape == ref.shape and _rel(y, ref) < 3e-6, (slope, tuple(y.shape), tuple(ref.shape))
        gy = torch.randn(ref.shape, generator=g).cuda()
        g0 = torch.autograd.grad(ref, (x, w, b), gy)
        g1 = torch.autograd.grad(y, (x, w, b), gy)
        for u, v, name in zip(g1, g0, 'xwb'):
            assert u.shape == v.shape and _rel(u, v) < 1e-5, (name, slope, _rel(u, v))


@pytest.mark.parametrize('Cin,Cout,K,s,pad,P,B,H', [(1, 32, 5, 3, 2, 2, 3, 500), (32, 128, 5, 3, 2, 3, 2, 201), (128, 512, 5, 3, 2, 11, 2, 61),
                                                     (512, 1024, 5, 3, 2, 7, 2, 23), (1024, 1024, 5, 1, 2, 5, 2, 9), (1024, 1, 3, 1, 1, 11, 2, 6)])
def test_period_folded_conv_matches_conv2d(Cin, Cout, K, s, pad, P, B, H):
    """MPD's Conv2d((K, 1), (s, 1)) on [B, C, H, P] == the dilation-P Conv1d on the flat [B, C, H * P] signal"""
    from ttscube_amd.hifigan.disc_hip import HipStridedConv
    g = torch.Generator().manual_seed(Cin + P)
    x = torch.randn(B, Cin, H, P, generator=g).cuda().requires_grad_(True)
    w = (torch.randn(Cout, Cin, K, 1, generator=g) / (Cin * K) ** 0.5).cuda().requires_grad_(True)
    b = (torch.randn(Cout, generator=g) * 0.1).cuda().requires_grad_(True)
    ref = F.conv2d(F.leaky_relu(x, 0.1), w, b, stride=(s, 1), padding=(pad, 0))
    y = HipStridedConv(Cin, Cout, K, s, pad, period=P)(x.flatten(2), w.squeeze(-1), b, in_slope=0.1).view(B, Cout, -1, P)
    assert y.shape == ref.shape and _rel(y, ref) < 3e-6
    gy = torch.randn(ref.shape, generator=g).cuda()
    for u, v, name in zip(torch.autograd.grad(y, (x, w, b), gy), torch.autograd.grad(ref, (x, w, b), gy), 'xwb'):
        assert u.shape == v.shape and _rel(u, v) < 1e-5, (name, _rel(u, v))


def test_group_shapes_that_do_not_tile_are_rejected():
    from ttscube_amd._lib import TTSCError
    from ttscube_amd.hifigan.disc_hip import HipStridedConv
    with pytest.raises(TTSCError, match='per group'):
        HipStridedConv(64, 96, 7, 1, 3, 2)        # 48 output channels per group: neither a multiple nor a divisor of a 32-row tile


@pytest.mark.parametrize('which', ['mpd', 'msd'])
def test_native_discriminators_match_torch_modules(which):
    from ttscube_amd.hifigan import discriminators as D
    from tests import torch_reference as TR          # the torch-op formulation of the same modules (test infrastructure)
    torch.manual_seed(7)
    m = (D.MultiPeriodDiscriminator() if which == 'mpd' else D.MultiScaleDiscriminator()).cuda()
    B, T = 2, 6000
    g = torch.Generator().manual_seed(8)
    y = (torch.rand(B, 1, T, generator=g) - 0.5).cuda()
    y_hat = (torch.rand(B, 1, T, generator=g) - 0.5).cuda().requires_grad_(True)
    ref_fwd = TR.mpd_forward if which == 'mpd' else TR.msd_forward
    with torch.no_grad():
        for _ in range(4):                          # spectral norm: let the power iteration settle (a fresh module's u, v are random),
            ref_fwd(m, y, y_hat)
    m.eval()                                        # ... then freeze it: no iteration between the two evaluations compared below
    fwd = lambda mod, a_, b_, **kw: mod(a_, b_, **kw)   # the drop-in classes dispatch to the HIP kernels themselves (cubegan.py:144,160 call shape)
    params = [p for p in m.parameters() if p.requires_grad]

    def losses(outs):
        rs, gs, fr, fg = outs
        ld = TR.discriminator_loss(rs, gs)[0]
        lg = TR.generator_loss(gs)[0] + TR.feature_loss(fr, fg)
        return ld, lg

    ref = ref_fwd(m, y, y_hat)
    nat = fwd(m, y, y_hat)
    sub = m.discriminators[1]                       # a sub-discriminator called on its own takes the HIP path too
    xin = y if which == 'mpd' else m.meanpools[0](y)
    o_sub, f_sub = sub(xin)
    o_ref, f_ref = (TR.disc_p_forward if which == 'mpd' else TR.disc_s_forward)(sub, xin)
    assert _rel(o_sub, o_ref) < 2e-5 and all(a_.shape == b_.shape and _rel(a_, b_) < 2e-5 for a_, b_ in zip(f_sub, f_ref))
    with pytest.raises(Exception, match='no CPU path'):
        m(y.cpu(), y_hat.detach().cpu())
    for i, d in enumerate(m.discriminators):
        for a, b_ in ((nat[0][i], ref[0][i]), (nat[1][i], ref[1][i])):
            assert a.shape == b_.shape and _rel(a, b_) < 2e-5, (which, i)
        for fa, fb in zip(nat[2][i] + nat[3][i], ref[2][i] + ref[3][i]):   # same layout as the reference modules, MPD's [B, C, H, p] included
            assert fa.shape == fb.shape and _rel(fa, fb) < 2e-5, (which, i, tuple(fa.shape))
    ld0, lg0 = losses(ref)
    ld1, lg1 = losses(nat)
    assert abs(float(ld0) - float(ld1)) < 1e-4 * abs(float(ld0)) and abs(float(lg0) - float(lg1)) < 1e-4 * abs(float(lg0))
    gp0 = torch.autograd.grad(ld0, params, retain_graph=True)
    gp1 = torch.autograd.grad(ld1, params, retain_graph=True)
    names = [n for n, p_ in m.named_parameters() if p_.requires_grad]
    # sums of ~1e5 signed terms per element in two different orders, and a loss that is not smooth: a leaky-relu gate that flips on a 1e-6 difference of
    # the forward pass changes one term of a first-layer weight gradient tenfold (a few 1e-3 of that tensor's norm; the split-precision forward is
    # 1e-6 from torch's fp32, the exact-fp32 kernels 1e-7).  Measured: <= 2e-3 for the deep layers, 2.5e-3 .. 4.1e-3 for the first layers; the
    # bounds leave a factor of 2-5 for run-to-run differences on the torch side (a wrong kernel is off by O(1)).
    bad = [(n, _rel(a, b_), bool(torch.isfinite(a).all()), bool(torch.isfinite(b_).all())) for n, a, b_ in zip(names, gp1, gp0)
           if not (_rel(a, b_) < (2e-2 if '.convs.0.' in n else 5e-3))]
    assert not bad, ('parameter gradients of the discriminator loss (name, rel, native finite, torch finite)', bad[:6])
    # what the generator receives through the discriminators.  The loss is not smooth (leaky-relu gates, the L1 feature loss): a forward pass
    # that differs in the 6th digit flips a few gates / signs, and torch's own fp32 gradient sits ~1e-3 (max norm) from the float64 one for
    # that reason alone.  So (a) against float64, the native gradient may be at most ~10x as far as torch-fp32 is; (b) on the SAME forward
    # graph the split-precision data gradients must agree with the exact-fp32 kernels' to 2e-6.
    from ttscube_amd.hifigan import autograd as A
    gx0 = torch.autograd.grad(lg0, y_hat, retain_graph=True)[0]
    gx1 = torch.autograd.grad(lg1, y_hat, retain_graph=True)[0]
    m64 = (D.MultiPeriodDiscriminator() if which == 'mpd' else D.MultiScaleDiscriminator()).cuda()
    m64.load_state_dict(m.state_dict())
    m64 = m64.double().eval()
    yh64 = y_hat.detach().double().requires_grad_(True)
    gx64 = torch.autograd.grad(losses(ref_fwd(m64, y.double(), yh64))[1], yh64)[0]
    assert _rel(gx1, gx64) < 10 * _rel(gx0, gx64) + 1e-3, (_rel(gx1, gx64), _rel(gx0, gx64))   # (a): a plausibility bound — which gates flip differs run to run on the torch side
    if A.SPLIT_TRAIN:
        try:
            A.SPLIT_TRAIN = False
            gx2 = torch.autograd.grad(lg1, y_hat, retain_graph=True)[0]    # same graph, data gradients on the fp32 kernel
        finally:
            A.SPLIT_TRAIN = True
        assert _rel(gx1, gx2) < 2e-6
    # discriminator step: generated audio carries no graph -> real + generated run as ONE batch; same values
    nat_d = fwd(m, y, y_hat.detach(), want_fmap=False)
    ld2 = D.discriminator_loss(nat_d[0], nat_d[1])[0]   # (the product's loss: gan_loss_kernel)
    assert abs(float(ld2) - float(ld0)) < 1e-4 * abs(float(ld0)) and nat_d[2][0] == []


@pytest.mark.parametrize('N,C,L,G,s,P,pad,K', [(3, 8, 100, 1, 3, 1, 2, 5), (2, 12, 61, 3, 2, 1, 20, 41), (2, 4, 37, 1, 3, 7, 2, 5), (1, 16, 50, 4, 4, 1, 20, 41),
                                              (2, 1, 200, 1, 3, 11, 2, 5), (2, 6, 3000, 2, 2, 1, 20, 41), (2, 4, 1500, 1, 3, 5, 2, 5), (1, 3, 5000, 1, 4, 1, 20, 41)])
def test_fused_deinterleave_equals_the_torch_views(N, C, L, G, s, P, pad, K):
    """`ttsc_deinterleave_x` / `_w` forward and backward == pad + view + permute + reshape and their autograd, exactly (pure data movement)"""
    from ttscube_amd.hifigan.disc_hip import _DeintW, _DeintX
    g = torch.Generator().manual_seed(N * 100 + C + s)
    J = -(-K // s)
    Lout = (L + 2 * pad - K) // s + 1
    M = Lout + J - 1
    x = torch.randn(N, C, L * P, generator=g).cuda().requires_grad_(True)
    xp = F.pad(x, (pad * P, (s * M - L - pad) * P))
    ref = xp.view(N, G, C // G, M, s, P).permute(0, 1, 4, 2, 3, 5).reshape(N, s * C, M * P)
    out = _DeintX.apply(x, G, s, P, pad, M)
    assert torch.equal(out, ref)
    gy = torch.randn(ref.shape, generator=g).cuda()
    assert torch.equal(torch.autograd.grad(out, x, gy)[0], torch.autograd.grad(ref, x, gy)[0])
    Cout = 6
    w = torch.randn(Cout, C // G, K, generator=g).cuda().requires_grad_(True)
    wref = F.pad(w, (0, s * J - K)).view(Cout, C // G, J, s).permute(0, 3, 1, 2).reshape(Cout, s * (C // G), J)
    wout = _DeintW.apply(w, s)
    assert torch.equal(wout, wref)
    gw = torch.randn(wref.shape, generator=g).cuda()
    assert torch.equal(torch.autograd.grad(wout, w, gw)[0], torch.autograd.grad(wref, w, gw)[0])


@pytest.mark.parametrize('Cout,Cin,K,groups', [(128, 1, 15, 1), (256, 128, 41, 16), (1024, 1024, 5, 1), (1, 1024, 3, 1)])
def test_spectral_norm_function_matches_torch_hook(Cout, Cin, K, groups):
    """HipSpectralNormFn against torch.nn.utils.spectral_norm's own pre-forward hook on a twin module: the normalised weight, the in-place
    power iteration on weight_u / weight_v (training mode, two consecutive calls), eval mode (no iteration), and the gradient w.r.t.
    weight_orig — the layers of DiscriminatorS(use_spectral_norm=True)."""
    import copy
    import torch.nn as nn
    from torch.nn.utils import spectral_norm
    from ttscube_amd.hifigan.disc_hip import _weight
    torch.manual_seed(Cout + K)
    a = spectral_norm(nn.Conv1d(Cin, Cout, K, groups=groups)).cuda()
    b = copy.deepcopy(a)
    for mode in ('train', 'train', 'eval'):
        a.train(mode == 'train')
        b.train(mode == 'train')
        for hook in a._forward_pre_hooks.values():
            hook(a, (None,))
        ref = a.weight
        got = _weight(b)
        assert _rel(got, ref) < 2e-6
        assert _rel(b.weight_u, a.weight_u) < 2e-6 and _rel(b.weight_v, a.weight_v) < 2e-6
        g = torch.randn_like(ref)
        gr, = torch.autograd.grad(ref, a.weight_orig, g)
        gg, = torch.autograd.grad(got, b.weight_orig, g)
        assert _rel(gg, gr) < 5e-6, mode
