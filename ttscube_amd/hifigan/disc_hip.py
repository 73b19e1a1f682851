"""HiFi-GAN discriminators on the HIP convolution kernels: `MultiPeriodDiscriminator` and
`MultiScaleDiscriminator` of hifigan.models [EXTERNAL; call sites cube/networks/cubegan.py:144-149,160-167] with forward, data
gradient and weight gradient of every convolution on `conv_mfma_kernel` / `conv_wgrad_kernel` (hifigan/autograd.py::HipConvFn),
leaky-relu fused into the next layer's staging pass, weight-norm on the fused kernels.  Parameters stay in the torch modules of
`discriminators.py` (same state_dict keys), which remain the torch-op formulation these functions are tested against.

How the discriminators' convolutions map onto the stride-1, 1-D kernels:
  * strided Conv1d(stride s) = stride-1 convolution over the phase-de-interleaved input  xr[n, (g, r, ci), m] = x[n, (g, ci), s m + r - p]
    with ceil(K / s) taps  W'[co, (r, ci), j] = w[co, ci, s j + r]  (zero beyond K): s * Cin_g input channels per group;
  * MPD's Conv2d((k, 1), stride (s, 1)) over the period-folded signal [B, C, T / p, p] never mixes the p columns: on the FLAT signal
    (index h p + w, i.e. the audio's own time order) it is a Conv1d with DILATION p — the de-interleave above runs over h with the
    p columns kept inside each block — so the batch stays B sequences of T / s^i samples (B p sequences of a few dozen frames would
    leave the kernel's 64..128-column tiles mostly empty), and the feature maps are the reference's [B, C, H, p] tensors as views;
  * MSD's grouped k = 41 layers use the kernels' group support (block-diagonal M tiles, `ttsc_conv_wgrad_grouped`)."""

import os

import torch
import torch.nn.functional as F

from .. import _lib
from .autograd import HipWeightNormFn, TrainConv, hip_conv
from .losses_hip import RawFmap
from .wbank import pass_range
from .streams import fan_out

LRELU_SLOPE = 0.1
FUSED_DEINTERLEAVE = os.environ.get('TTSC_FUSED_DEINTERLEAVE', '1') != '0'


class _DeintX(torch.autograd.Function):
    """xr[n, (g, r, ci), m P + w] = x[n, (g, ci), ((m s + r) - pad) P + w] (zero outside) — one gather forward, one backward (`ttsc_deinterleave_x`)
    instead of pad + view + permute + reshape"""

    @staticmethod
    def forward(ctx, x, G, s, P, pad, M):
        x = x.contiguous()
        N, C, LP = x.shape
        out = torch.empty((N, s * C, M * P), dtype=torch.float32, device=x.device)
        with _lib.on_device(x.device):
            _lib.check(_lib.lib().ttsc_deinterleave_x(_lib.dev_ptr(x), _lib.dev_ptr(out), N, C, LP // P, G, s, P, pad, M, 0, _lib.current_stream()),
                       'ttsc_deinterleave_x')
        ctx.meta = (N, C, LP, G, s, P, pad, M)
        return out      # (HipStridedConv passes x's range word on: a permutation with zero padding / a crop keeps the bound)

    @staticmethod
    def backward(ctx, g):
        N, C, LP, G, s, P, pad, M = ctx.meta
        g = g.contiguous()
        dx = torch.empty((N, C, LP), dtype=torch.float32, device=g.device)
        with _lib.on_device(g.device):
            _lib.check(_lib.lib().ttsc_deinterleave_x(_lib.dev_ptr(g), _lib.dev_ptr(dx), N, C, LP // P, G, s, P, pad, M, 1, _lib.current_stream()),
                       'ttsc_deinterleave_x')
        return pass_range(g, dx), None, None, None, None, None


class _DeintW(torch.autograd.Function):
    """wp[co, (r, ci), j] = w[co, ci, s j + r] (zero beyond K) and its adjoint (`ttsc_deinterleave_w`)"""

    @staticmethod
    def forward(ctx, w, s):
        w = w.contiguous()
        Cout, Cg, K = w.shape
        J = -(-K // s)
        out = torch.empty((Cout, s * Cg, J), dtype=torch.float32, device=w.device)
        with _lib.on_device(w.device):
            _lib.check(_lib.lib().ttsc_deinterleave_w(_lib.dev_ptr(w), _lib.dev_ptr(out), Cout, Cg, K, s, 0, _lib.current_stream()), 'ttsc_deinterleave_w')
        ctx.meta = (Cout, Cg, K, s)
        return out

    @staticmethod
    def backward(ctx, g):
        Cout, Cg, K, s = ctx.meta
        g = g.contiguous()
        dw = torch.empty((Cout, Cg, K), dtype=torch.float32, device=g.device)
        with _lib.on_device(g.device):
            _lib.check(_lib.lib().ttsc_deinterleave_w(_lib.dev_ptr(g), _lib.dev_ptr(dw), Cout, Cg, K, s, 1, _lib.current_stream()), 'ttsc_deinterleave_w')
        return dw, None


class HipStridedConv:
    """differentiable Conv1d(Cin -> Cout, K, stride, padding, groups) on the HIP kernels"""

    def __init__(self, Cin, Cout, K, stride=1, padding=0, groups=1, period=1):
        """period > 1: the signal is a period-folded image flattened as h * period + w and the convolution runs over h only
        (MPD's (K, 1) kernels): taps `period` samples apart, stride / padding counted in rows of `period` samples."""
        self.Cin, self.Cout, self.K, self.s, self.p, self.G, self.P = Cin, Cout, K, stride, padding, groups, period
        self.J = -(-K // stride)
        if stride == 1:
            self.tc = TrainConv(Cin, Cout, K, padding=padding * period, dilation=period, groups=groups)
        else:
            self.tc = TrainConv(stride * Cin, Cout, self.J, padding=0, dilation=period, groups=groups)

    def __call__(self, x, w, b, in_slope=1.0):
        banked = getattr(w, '_ttsc_pack', None) is not None      # weight out of a WeightBank: already in the layout the convolution sees
        if self.s == 1:
            return hip_conv(self.tc, x, w if banked else w.contiguous(), b, in_slope=in_slope)
        s, K, J, G, P = self.s, self.K, self.J, self.G, self.P
        N, Cin, LP = x.shape
        L = LP // P                                                     # rows of P samples
        Lout = (L + 2 * self.p - K) // s + 1
        M = Lout + J - 1
        if banked:               # (the tap de-interleave of the weight is folded into the bank's fragment packing)
            xr = pass_range(x, _DeintX.apply(x, G, s, P, self.p, M)) if x.is_contiguous() else _DeintX.apply(x, G, s, P, self.p, M)
            wp = w
        elif FUSED_DEINTERLEAVE:   # one gather per operand and direction (csrc/train_ops.hip)
            xr = _DeintX.apply(x, G, s, P, self.p, M)
            wp = _DeintW.apply(w, s)
        else:                    # the same maps as torch views + copies (kept as the formulation the kernels are tested against)
            xp = F.pad(x, (self.p * P, (s * M - L - self.p) * P))           # (a negative right pad crops)
            xr = xp.view(N, G, Cin // G, M, s, P).permute(0, 1, 4, 2, 3, 5).reshape(N, s * Cin, M * P)
            wp = F.pad(w, (0, s * J - K)).view(self.Cout, Cin // G, J, s).permute(0, 3, 1, 2).reshape(self.Cout, s * (Cin // G), J).contiguous()
        # leaky-relu commutes with the de-interleave (and lrelu(0) = 0 keeps the zero padding): it rides in the kernel's staging pass
        return hip_conv(self.tc, xr, wp, b, in_slope=in_slope)


class HipSpectralNormFn(torch.autograd.Function):
    """torch.nn.utils.spectral_norm's weight (SpectralNorm.compute_weight, dim = 0, one power iteration, eps 1e-12) on HIP kernels:
    training mode updates the module's `weight_u` / `weight_v` buffers in place exactly where torch's pre-forward hook does —
    v <- normalize(W^T u), u <- normalize(W v) — then wn = W / sigma with sigma = u^T W v; u and v are constants of the backward pass:
    dW = dWn / sigma - (sum(dWn . W) / sigma^2) u v^T.  Mat-vecs, normalisations, the dot product and the two element-wise passes are the
    fixed-order kernels of csrc/train_ops.hip (no library call: torch's hook goes through rocBLAS gemv)."""

    @staticmethod
    def forward(ctx, w, u, v, training, eps):
        L = _lib.lib()
        R = w.shape[0]
        W2 = w.detach().contiguous().reshape(R, -1)
        Cc = W2.shape[1]
        P, S = _lib.dev_ptr, _lib.current_stream
        sigma = torch.empty(1, dtype=torch.float32, device=w.device)
        t2 = torch.empty(R, dtype=torch.float32, device=w.device)
        with _lib.on_device(w.device):
            if training:
                t1 = torch.empty(Cc, dtype=torch.float32, device=w.device)
                ws = torch.empty(max(int(L.ttsc_matvec_workspace_bytes(R, Cc)) // 4, 1), dtype=torch.float32, device=w.device)
                _lib.check(L.ttsc_matvec(P(W2), R, Cc, P(u), 1, P(t1), P(ws), ws.numel() * 4, S()), 'ttsc_matvec')       # W^T u
                _lib.check(L.ttsc_l2_normalize(P(t1), Cc, eps, P(v), None, S()), 'ttsc_l2_normalize')
                _lib.check(L.ttsc_matvec(P(W2), R, Cc, P(v), 0, P(t2), None, 0, S()), 'ttsc_matvec')                      # W v
                _lib.check(L.ttsc_l2_normalize(P(t2), R, eps, P(u), None, S()), 'ttsc_l2_normalize')
            else:
                _lib.check(L.ttsc_matvec(P(W2), R, Cc, P(v), 0, P(t2), None, 0, S()), 'ttsc_matvec')
            wd = torch.empty(max(int(L.ttsc_dot_workspace_bytes(R)) // 4, 1), dtype=torch.float32, device=w.device)
            _lib.check(L.ttsc_dot(P(u), P(t2), R, P(sigma), P(wd), wd.numel() * 4, S()), 'ttsc_dot')                  # sigma = u . (W v)
            wn = torch.empty_like(W2)
            _lib.check(L.ttsc_div_scalar(P(W2), P(sigma), P(wn), W2.numel(), S()), 'ttsc_div_scalar')
        ctx.save_for_backward(W2, u.clone(), v.clone(), sigma)      # (the buffers move on with the next power iteration)
        ctx.shape = w.shape
        return wn.view(w.shape)

    @staticmethod
    def backward(ctx, dwn):
        W2, u, v, sigma = ctx.saved_tensors
        L = _lib.lib()
        R, Cc = W2.shape
        P, S = _lib.dev_ptr, _lib.current_stream
        d2 = dwn.contiguous().reshape(R, Cc)
        dot = torch.empty(1, dtype=torch.float32, device=W2.device)
        dw = torch.empty_like(W2)
        with _lib.on_device(W2.device):
            ws = torch.empty(max(int(L.ttsc_dot_workspace_bytes(W2.numel())) // 4, 1), dtype=torch.float32, device=W2.device)
            _lib.check(L.ttsc_dot(P(d2), P(W2), W2.numel(), P(dot), P(ws), ws.numel() * 4, S()), 'ttsc_dot')
            _lib.check(L.ttsc_spectral_norm_backward(P(d2), P(u), P(v), P(sigma), P(dot), P(dw), R, Cc, S()), 'ttsc_spectral_norm_backward')
        return dw.view(ctx.shape), None, None, None, None


def _spectral_eps(l):
    for hook in l._forward_pre_hooks.values():
        if hasattr(hook, 'eps') and hasattr(hook, 'n_power_iterations'):
            if hook.n_power_iterations != 1 or hook.dim != 0:
                raise _lib.TTSCError('spectral norm: only dim = 0 with one power iteration is built (torch defaults; got dim %d, %d iterations)'
                                     % (hook.dim, hook.n_power_iterations))
            return float(hook.eps)
    return 1e-12


def _weight(l):
    """the layer's effective weight, differentiable w.r.t. its parameters: weight norm and spectral norm (power iteration included, in place on
    the module's buffers, as a module call would do) on the fused kernels"""
    if hasattr(l, 'weight_orig'):
        return HipSpectralNormFn.apply(l.weight_orig, l.weight_u, l.weight_v, l.training, _spectral_eps(l))
    if hasattr(l, 'weight_g'):   # torch.nn.utils.weight_norm keeps a stale plain `weight` attribute beside (weight_g, weight_v): never read it
        return HipWeightNormFn.apply(l.weight_v, l.weight_g)
    return l.weight


def _layers(d, kind):
    cache = getattr(d, '_hip_layers', None)
    if cache is None:
        cache = []
        for l in list(d.convs) + [d.conv_post]:
            if kind == 'p':
                cache.append(HipStridedConv(l.in_channels, l.out_channels, l.kernel_size[0], l.stride[0], l.padding[0], period=d.period))
            else:
                cache.append(HipStridedConv(l.in_channels, l.out_channels, l.kernel_size[0], l.stride[0], l.padding[0], l.groups))
        object.__setattr__(d, '_hip_layers', cache)
    return cache


def _bank_of(module, kind):
    """the WeightBank over every weight-normed layer of all sub-discriminators of `module` (MultiPeriodDiscriminator / MultiScaleDiscriminator) that
    the split-precision path takes in both directions, and {id(layer): entry index}; (False, {}) when there is nothing to bank"""
    from .autograd import USE_BANK, SPLIT_TRAIN, _split_ok
    cached = getattr(module, '_wbank', None)
    if cached is not None:
        return cached
    specs, index = [], {}
    if USE_BANK and SPLIT_TRAIN:
        from .wbank import WeightBank
        for d in module.discriminators:
            for l, h in zip(list(d.convs) + [d.conv_post], _layers(d, kind)):
                tc = h.tc
                if hasattr(l, 'weight_g') and not hasattr(l, 'weight_orig') and _split_ok(tc.Cin, tc.Cout, tc.K, tc.dilation, tc.groups) and \
                        _split_ok(tc.Cout, tc.Cin, tc.K, tc.dilation, tc.groups) and tc.dilation * (tc.K - 1) - tc.padding >= 0:
                    index[id(l)] = len(specs)
                    specs.append((l, h.Cin, h.Cout, h.K, h.G, h.s))
    cached = (WeightBank(specs), index) if specs else (False, {})
    object.__setattr__(module, '_wbank', cached)
    return cached


def _run(d, kind, x, want_fmap, bank=None, index=None):
    """x [N, 1, L] -> (scores [N, L'], fmap list; MPD: feature maps as [N, C, H, period] views, the reference's layout).  Layer i's leaky-relu is applied by layer i+1's staging pass; the activated
    feature maps are only materialised when the caller needs them (the generator step's feature-matching loss)."""
    hl = _layers(d, kind)
    mods = list(d.convs) + [d.conv_post]
    fmap = []
    slope = 1.0
    for i, (l, h) in enumerate(zip(mods, hl)):
        bi = index.get(id(l)) if bank else None
        if bi is not None:
            w = bank.weight(bi)
        else:
            w = _weight(l)
            if w.dim() == 4:
                w = w.squeeze(-1)
        x = h(x, w, l.bias, in_slope=slope)
        slope = LRELU_SLOPE
        if want_fmap == 'raw':       # the training step's feature-matching loss applies the activation itself (losses_hip.RawFmap)
            fmap.append(RawFmap(x, LRELU_SLOPE if i < len(mods) - 1 else 1.0))
        elif want_fmap:
            f = F.leaky_relu(x, LRELU_SLOPE) if i < len(mods) - 1 else x
            fmap.append(f.view(f.shape[0], f.shape[1], -1, d.period) if kind == 'p' else f)
    return torch.flatten(x, 1, -1), fmap


def _fold(x, p):
    """[B, 1, T] reflect-padded to a multiple of the period (DiscriminatorP.forward); the fold itself is implicit: index h p + w"""
    t = x.shape[2]
    return F.pad(x, (0, p - (t % p)), 'reflect') if t % p != 0 else x


def _pair(d, kind, y, y_hat, batch_ok, want_fmap, bank=None, index=None):
    if batch_ok and not y_hat.requires_grad and y.shape == y_hat.shape:   # discriminator step: real + generated as one batch
        n = y.shape[0]
        out, fmap = _run(d, kind, torch.cat([y, y_hat], dim=0), want_fmap, bank, index)
        cut = (lambda f, a, b: RawFmap(f.x[a:b], f.slope)) if want_fmap == 'raw' else (lambda f, a, b: f[a:b])
        return out[:n], [cut(f, 0, n) for f in fmap], out[n:], [cut(f, n, None) for f in fmap]
    out_r, fmap_r = _run(d, kind, y, want_fmap, bank, index)
    out_g, fmap_g = _run(d, kind, y_hat, want_fmap, bank, index)
    return out_r, fmap_r, out_g, fmap_g


# The sub-discriminators (5 periods, 3 scales) are independent graphs of small launches: each runs on its own stream (streams.fan_out)


def mpd_forward(mpd, y, y_hat, want_fmap=True):
    """MultiPeriodDiscriminator.forward(y, y_hat) on the HIP kernels: (y_d_rs, y_d_gs, fmap_rs, fmap_gs)"""
    if not y.is_cuda:
        raise _lib.TTSCError('discriminators: inputs must live on a HIP device; no CPU path')
    res = ([], [], [], [])
    bank, index = _bank_of(mpd, 'p')
    if bank:
        bank.prepare()      # one preparation for all five sub-discriminators, on the current stream: the side streams fork behind it
    jobs = [(lambda d=d: _pair(d, 'p', _fold(y, d.period), _fold(y_hat, d.period), True, want_fmap, bank, index)) for d in mpd.discriminators]
    for r in fan_out(jobs, y.device, inputs=[y, y_hat], tag='mpd'):
        for acc, v in zip(res, (r[0], r[2], r[1], r[3])):
            acc.append(v)
    return res


def msd_forward(msd, y, y_hat, want_fmap=True):
    """MultiScaleDiscriminator.forward(y, y_hat) on the HIP kernels"""
    if not y.is_cuda:
        raise _lib.TTSCError('discriminators: inputs must live on a HIP device; no CPU path')
    res = ([], [], [], [])
    ins = []
    for i, d in enumerate(msd.discriminators):
        if i != 0:
            y = msd.meanpools[i - 1](y)
            y_hat = msd.meanpools[i - 1](y_hat)
        ins.append((y, y_hat))
    # discriminator 0 is spectrally normed: two calls, two power iterations
    bank, index = _bank_of(msd, 's')
    if bank:
        bank.prepare()
    jobs = [(lambda i=i, d=d: _pair(d, 's', ins[i][0], ins[i][1], i != 0, want_fmap, bank, index)) for i, d in enumerate(msd.discriminators)]
    for r in fan_out(jobs, y.device, inputs=[t for pair in ins for t in pair], tag='msd'):
        for acc, v in zip(res, (r[0], r[2], r[1], r[3])):
            acc.append(v)
    return res
