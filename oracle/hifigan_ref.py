"""ORACLE (test infrastructure — never imported by the product path).

CPU restatement, in plain fp32 torch functional ops, of the HiFi-GAN V1 generator
that TTS-Cube calls as ``hifigan.models.Generator`` (reference call sites:
cube/networks/cubegan.py:41-43,72,83,131 and cube/io_utils/runtime.py:49-54,78).

PARITY STATUS: **unpinned against the reference's own source** — ``hifigan/`` is an
empty, un-vendored submodule in /root/reference (.gitmodules:4-6, tiberiu44/hifi-gan,
pinned SHA unrecoverable).  The algorithm below is the published one (Kong, Kim, Bae
2020, arXiv:2010.05646, "V1" generator) driven by the reference's own config
(examples/hifigan/config_v1.json:2,11-15).  It is cross-validated in
tests/test_oracle_hifigan.py against golden vectors produced by an *independent*
implementation of the same architecture (transformers' SpeechT5HifiGan), generated
by tools/gen_golden_hifigan.py.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import torch
import torch.nn.functional as F

LRELU_SLOPE = 0.1  # hifigan models.py LRELU_SLOPE (published implementation)


def fold_weight_norm(g, v):
    """w = g * v / ||v||, norm over every dim but 0 (torch.nn.utils.weight_norm, dim=0).

    H9 in SURVEY.md §2.3; runtime.py:53 ``remove_weight_norm`` performs the same fold."""
    norm = v.reshape(v.shape[0], -1).norm(dim=1).reshape([-1] + [1] * (v.dim() - 1))
    return g * v / norm


def fold_state_dict(sd):
    """Accept the reference checkpoint layout (``weight_g``/``weight_v``), the new
    torch parametrization layout (``parametrizations.weight.original0/1``) or already
    folded ``weight`` keys; return {name.weight, name.bias} fp32 tensors."""
    out = {}
    for k, v in sd.items():
        v = torch.as_tensor(v).float()
        if k.endswith('.weight_g'):
            base = k[:-len('.weight_g')]
            out[base + '.weight'] = fold_weight_norm(v, torch.as_tensor(sd[base + '.weight_v']).float())
        elif k.endswith('.parametrizations.weight.original0'):
            base = k[:-len('.parametrizations.weight.original0')]
            out[base + '.weight'] = fold_weight_norm(
                v, torch.as_tensor(sd[base + '.parametrizations.weight.original1']).float())
        elif k.endswith('.weight_v') or k.endswith('.parametrizations.weight.original1'):
            continue
        else:
            out[k] = v
    return out


def get_padding(kernel_size, dilation=1):
    return int((kernel_size * dilation - dilation) / 2)


def resblock1(w, prefix, x, kernel_size, dilations):
    """ResBlock1: for each dilation d: x = x + c2(lrelu(c1_d(lrelu(x)))) (H6)."""
    for m, d in enumerate(dilations):
        xt = F.leaky_relu(x, LRELU_SLOPE)
        xt = F.conv1d(xt, w['%s.convs1.%d.weight' % (prefix, m)], w['%s.convs1.%d.bias' % (prefix, m)],
                      dilation=d, padding=get_padding(kernel_size, d))
        xt = F.leaky_relu(xt, LRELU_SLOPE)
        xt = F.conv1d(xt, w['%s.convs2.%d.weight' % (prefix, m)], w['%s.convs2.%d.bias' % (prefix, m)],
                      dilation=1, padding=get_padding(kernel_size, 1))
        x = xt + x
    return x


def resblock2(w, prefix, x, kernel_size, dilations):
    """ResBlock2 (config "resblock": "2"): x = x + c_d(lrelu(x))."""
    for m, d in enumerate(dilations):
        xt = F.leaky_relu(x, LRELU_SLOPE)
        xt = F.conv1d(xt, w['%s.convs.%d.weight' % (prefix, m)], w['%s.convs.%d.bias' % (prefix, m)],
                      dilation=d, padding=get_padding(kernel_size, d))
        x = xt + x
    return x


def generator_forward(w, h, mel, return_stages=False):
    """mel [B, num_mels, T] fp32 -> wav [B, 1, L].  ``w`` = folded weights, ``h`` = config dict.

    conv_pre(k7,p3) -> for each upsample i: lrelu(0.1) -> ConvTranspose1d(k_i, u_i, pad (k_i-u_i)//2)
    -> mean over the resblocks -> lrelu(0.01) -> conv_post(k7,p3) -> tanh   (H1..H8)."""
    rates = h['upsample_rates']
    ksz = h['upsample_kernel_sizes']
    rks = h['resblock_kernel_sizes']
    rds = h['resblock_dilation_sizes']
    nk = len(rks)
    rb = resblock1 if str(h.get('resblock', '1')) == '1' else resblock2
    stages, ups_out = [], []
    x = F.conv1d(mel, w['conv_pre.weight'], w['conv_pre.bias'], padding=3)
    stages.append(x)
    for i, (u, k) in enumerate(zip(rates, ksz)):
        x = F.leaky_relu(x, LRELU_SLOPE)
        x = F.conv_transpose1d(x, w['ups.%d.weight' % i], w['ups.%d.bias' % i], stride=u, padding=(k - u) // 2)
        ups_out.append(x)
        xs = None
        for j in range(nk):
            r = rb(w, 'resblocks.%d' % (i * nk + j), x, rks[j], rds[j])
            xs = r if xs is None else xs + r
        x = xs / nk
        stages.append(x)
    x = F.leaky_relu(x)  # default slope 0.01 (H8)
    x = F.conv1d(x, w['conv_post.weight'], w['conv_post.bias'], padding=3)
    x = torch.tanh(x)
    if return_stages == 'all':   # + every upsampler's output (tests/test_oracle_hifigan.py: layer-by-layer pin against the surrogate)
        return x, stages, ups_out
    if return_stages:
        return x, stages
    return x


def out_len(h, T):
    L = T
    for u, k in zip(h['upsample_rates'], h['upsample_kernel_sizes']):
        L = (L - 1) * u - 2 * ((k - u) // 2) + k
    return L


def synthetic_state_dict(h, seed=1234, weight_norm=True, gain=1.0):
    """Seeded, variance-preserving synthetic weights in the reference checkpoint key layout
    (conv_pre, ups.N, resblocks.N.convs{1,2}.M, conv_post; each weight_g/weight_v/bias).
    HiFi-GAN's own init N(0, 0.01) gives ~0 output, which would make an RMS check vacuous (§7 hard parts)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}

    def add(name, shape, fan_in, transposed=False):
        std = gain / (fan_in ** 0.5)
        v = torch.randn(shape, generator=g) * std
        b = torch.randn(shape[1] if transposed else shape[0], generator=g) * 0.05
        if weight_norm:
            # perturb g around ||v|| so that the fold is not an identity
            norm = v.reshape(v.shape[0], -1).norm(dim=1).reshape([-1] + [1] * (v.dim() - 1))
            sd[name + '.weight_g'] = norm * (1.0 + 0.1 * torch.randn(norm.shape, generator=g))
            sd[name + '.weight_v'] = v * (0.5 + torch.rand(norm.shape, generator=g))
        else:
            sd[name + '.weight'] = v
        sd[name + '.bias'] = b

    ch = h['upsample_initial_channel']
    nm = h.get('num_mels', 80)
    add('conv_pre', (ch, nm, 7), nm * 7)
    nk = len(h['resblock_kernel_sizes'])
    for i, (u, k) in enumerate(zip(h['upsample_rates'], h['upsample_kernel_sizes'])):
        cin, cout = ch // (2 ** i), ch // (2 ** (i + 1))
        # each output sample sees ~k/u taps of cin channels; lrelu halves the variance
        add('ups.%d' % i, (cin, cout, k), cin * k / u * 0.55, transposed=True)
        for j, (rk, rd) in enumerate(zip(h['resblock_kernel_sizes'], h['resblock_dilation_sizes'])):
            n = i * nk + j
            if str(h.get('resblock', '1')) == '1':
                for m in range(len(rd)):
                    add('resblocks.%d.convs1.%d' % (n, m), (cout, cout, rk), cout * rk * 0.55 * 4)
                    add('resblocks.%d.convs2.%d' % (n, m), (cout, cout, rk), cout * rk * 0.55 * 4)
            else:
                for m in range(len(rd)):
                    add('resblocks.%d.convs.%d' % (n, m), (cout, cout, rk), cout * rk * 0.55 * 4)
    clast = ch // (2 ** len(h['upsample_rates']))
    add('conv_post', (1, clast, 7), clast * 7 * 12.0)  # pre-tanh rms ~0.4: keep tanh unsaturated
    return sd


def synthetic_mel(B, T, num_mels=80, seed=1234):
    """clip(N(-2,1), -5, 1): log10-mel range of the reference (floor -5: cube/io_utils/vocoder.py:96-98)."""
    g = torch.Generator().manual_seed(seed)
    return torch.clamp(torch.randn(B, num_mels, T, generator=g) - 2.0, -5.0, 1.0)


CONFIG_V1 = {
    # examples/hifigan/config_v1.json:2,11-15,17-24
    "resblock": "1",
    "upsample_rates": [5, 3, 4, 4],
    "upsample_kernel_sizes": [16, 16, 4, 4],
    "upsample_initial_channel": 512,
    "resblock_kernel_sizes": [3, 7, 11],
    "resblock_dilation_sizes": [[1, 3, 5], [1, 3, 5], [1, 3, 5]],
    "num_mels": 80,
    "hop_size": 240,
    "sampling_rate": 24000,
}
