"""The all-torch formulation of the training steps — TEST INFRASTRUCTURE, never imported by the product package.

`install(monkeypatch)` swaps every native building block of ttscube_amd.networks.training (HIP autograd functions, GAN-loss kernels,
flat-arena AdamW, native discriminators, MFMA Linears) for its torch-op counterpart on the SAME parameter tensors, so that a test can
run the product's step function twice — native and all-torch — and compare losses / gradients
(reference semantics: cube/networks/cubegan.py:85-189, cube/networks/modules.py:505-563)."""
import torch
import torch.nn.functional as F


def _wn(l):
    """live weight-norm: w = g * v / ||v|| (so that gradients reach weight_g and weight_v)."""
    if hasattr(l, 'weight'):
        return l.weight
    v, g = l.weight_v, l.weight_g
    return g * v / v.reshape(v.shape[0], -1).norm(dim=1).reshape(-1, 1, 1)


def generator_forward_train(gen, x):
    """HiFi-GAN generator forward as torch ops: the autograd reference of hifigan/autograd.py::generator_forward_with_grad."""
    from ttscube_amd.hifigan.models import ResBlock1
    h = gen.h
    x = F.conv1d(x, _wn(gen.conv_pre), gen.conv_pre.bias, padding=3)
    nk = gen.num_kernels
    for i, (u, k) in enumerate(zip(h['upsample_rates'], h['upsample_kernel_sizes'])):
        x = F.leaky_relu(x, 0.1)
        x = F.conv_transpose1d(x, _wn(gen.ups[i]), gen.ups[i].bias, stride=u, padding=(k - u) // 2)
        xs = None
        for j in range(nk):
            rb = gen.resblocks[i * nk + j]
            kr, ds = h['resblock_kernel_sizes'][j], h['resblock_dilation_sizes'][j]
            r = x
            if isinstance(rb, ResBlock1):
                for c1, c2, d in zip(rb.convs1, rb.convs2, ds):
                    xt = F.conv1d(F.leaky_relu(r, 0.1), _wn(c1), c1.bias, dilation=d, padding=d * (kr - 1) // 2)
                    xt = F.conv1d(F.leaky_relu(xt, 0.1), _wn(c2), c2.bias, padding=(kr - 1) // 2)
                    r = xt + r
            else:
                for c, d in zip(rb.convs, ds):
                    r = F.conv1d(F.leaky_relu(r, 0.1), _wn(c), c.bias, dilation=d, padding=d * (kr - 1) // 2) + r
            xs = r if xs is None else xs + r
        x = xs / nk
    x = F.leaky_relu(x)
    x = F.conv1d(x, _wn(gen.conv_post), gen.conv_post.bias, padding=3)
    return torch.tanh(x)


def _text_ops(lang):
    def cnn(name, h):
        for layer in getattr(lang, name):
            if hasattr(layer, 'conv'):
                h = torch.tanh(F.conv1d(h, layer.conv.weight, layer.conv.bias, padding=1))
        return h
    return (lambda emb, idx: emb(idx)), F.linear, cnn


def _make_adamw(params, lr):
    return torch.optim.AdamW(params, lr, betas=[0.8, 0.99], fused=all(p.is_cuda for p in params))


def _gan_loss_fns():
    return discriminator_loss, feature_loss, generator_loss


def _discriminator_fns(model):
    return (lambda a_, b_, fm=True: mpd_forward(model._mpd, a_, b_)), (lambda a_, b_, fm=True: msd_forward(model._msd, a_, b_))


def _lowres_features(net, hidden):
    for conv in net._lowres_conv:
        hidden = torch.tanh(F.conv1d(hidden, conv.conv.weight, conv.conv.bias, padding=3))
    return hidden


def _output_linears(net, hidden):
    pre = torch.tanh(F.linear(hidden, net._preoutput.linear_layer.weight, net._preoutput.linear_layer.bias))
    return F.linear(pre, net._output.linear_layer.weight, net._output.linear_layer.bias)


# ---- HiFi-GAN discriminators, GAN losses and mel-spectrogram as torch ops (Kong et al. 2020, public layout) ------------------------------
# The product classes (ttscube_amd/hifigan/discriminators.py) hold the parameters and dispatch to the HIP kernels; these functions evaluate
# the same modules' layers with torch ops (l(x) runs torch's weight-norm / spectral-norm hooks and F.conv1d / F.conv2d).

LRELU_SLOPE = 0.1


def disc_p_forward(d, x):
    fmap = []
    b, c, t = x.shape
    if t % d.period != 0:
        n_pad = d.period - (t % d.period)
        x = F.pad(x, (0, n_pad), 'reflect')
        t = t + n_pad
    x = x.view(b, c, t // d.period, d.period)
    for l in d.convs:
        x = F.leaky_relu(l(x), LRELU_SLOPE)
        fmap.append(x)
    x = d.conv_post(x)
    fmap.append(x)
    return torch.flatten(x, 1, -1), fmap


def disc_s_forward(d, x):
    fmap = []
    for l in d.convs:
        x = F.leaky_relu(l(x), LRELU_SLOPE)
        fmap.append(x)
    x = d.conv_post(x)
    fmap.append(x)
    return torch.flatten(x, 1, -1), fmap


def _sub_forward(d, x):
    return disc_p_forward(d, x) if hasattr(d, 'period') else disc_s_forward(d, x)


def disc_pair(d, y, y_hat, batch_ok, fwd=None):
    """(out_r, fmap_r, out_g, fmap_g) of sub-discriminator d on real / generated audio; when the generated signal carries no gradient a
    weight-normed sub-discriminator may see both as ONE batch (there are no batch statistics) — the launch diet disc_hip._pair uses"""
    fwd = fwd or (lambda x: _sub_forward(d, x))
    if batch_ok and not y_hat.requires_grad and y.shape == y_hat.shape:
        n = y.shape[0]
        out, fmap = fwd(torch.cat([y, y_hat], dim=0))
        return out[:n], [f[:n] for f in fmap], out[n:], [f[n:] for f in fmap]
    y_d_r, fmap_r = fwd(y)
    y_d_g, fmap_g = fwd(y_hat)
    return y_d_r, fmap_r, y_d_g, fmap_g


def mpd_forward(m, y, y_hat):
    res = ([], [], [], [])
    for d in m.discriminators:
        r = disc_pair(d, y, y_hat, True)
        for acc, v in zip(res, (r[0], r[2], r[1], r[3])):
            acc.append(v)
    return res


def msd_forward(m, y, y_hat):
    res = ([], [], [], [])
    for i, d in enumerate(m.discriminators):
        if i != 0:
            y = m.meanpools[i - 1](y)
            y_hat = m.meanpools[i - 1](y_hat)
        r = disc_pair(d, y, y_hat, i != 0)   # discriminator 0 is spectrally normed: two calls, two power iterations
        for acc, v in zip(res, (r[0], r[2], r[1], r[3])):
            acc.append(v)
    return res


def feature_loss(fmap_r, fmap_g):
    loss = 0
    for dr, dg in zip(fmap_r, fmap_g):
        for rl, gl in zip(dr, dg):
            loss = loss + torch.mean(torch.abs(rl - gl))
    return loss * 2


def discriminator_loss(disc_real_outputs, disc_generated_outputs):
    loss = 0
    r_losses, g_losses = [], []
    for dr, dg in zip(disc_real_outputs, disc_generated_outputs):
        r_loss = torch.mean((1 - dr) ** 2)
        g_loss = torch.mean(dg ** 2)
        loss = loss + (r_loss + g_loss)
        r_losses.append(r_loss.detach())
        g_losses.append(g_loss.detach())
    return loss, r_losses, g_losses


def generator_loss(disc_outputs):
    loss = 0
    gen_losses = []
    for dg in disc_outputs:
        l = torch.mean((1 - dg) ** 2)
        gen_losses.append(l)
        loss = loss + l
    return loss, gen_losses


_mel_basis = {}
_hann = {}


def _mel_filterbank(sr, n_fft, n_mels, fmin, fmax):
    """Slaney-style mel filterbank (librosa.filters.mel defaults: htk=False, norm='slaney'), restated in numpy."""
    import numpy as np

    def hz_to_mel(f):
        f = np.asarray(f, dtype=np.float64)
        f_sp = 200.0 / 3
        mels = f / f_sp
        min_log_hz = 1000.0
        min_log_mel = min_log_hz / f_sp
        logstep = np.log(6.4) / 27.0
        return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-10) / min_log_hz) / logstep, mels)

    def mel_to_hz(m):
        m = np.asarray(m, dtype=np.float64)
        f_sp = 200.0 / 3
        min_log_hz = 1000.0
        min_log_mel = min_log_hz / f_sp
        logstep = np.log(6.4) / 27.0
        return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)

    fftfreqs = np.linspace(0, sr / 2.0, 1 + n_fft // 2)
    mel_f = mel_to_hz(np.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    w = np.zeros((n_mels, 1 + n_fft // 2))
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        w[i] = np.maximum(0, np.minimum(lower, upper))
    w *= (2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels]))[:, None]
    return w.astype(np.float32)


def mel_spectrogram(y, n_fft, num_mels, sampling_rate, hop_size, win_size, fmin, fmax, center=False):
    """hifigan.meldataset.mel_spectrogram (published implementation): reflect-pad (n_fft-hop)/2, STFT (hann), magnitude
    sqrt(re^2+im^2+1e-9), mel projection, log(clamp(x, 1e-5)).  y [B, L] -> [B, num_mels, frames]."""
    key = '%s_%s_%s_%s_%s' % (n_fft, num_mels, sampling_rate, fmin, fmax)
    dk = key + '_' + str(y.device)
    if dk not in _mel_basis:
        _mel_basis[dk] = torch.from_numpy(_mel_filterbank(sampling_rate, n_fft, num_mels, fmin, fmax)).to(y.device)
        _hann[str(win_size) + '_' + str(y.device)] = torch.hann_window(win_size).to(y.device)
    pad = int((n_fft - hop_size) / 2)
    y = F.pad(y.unsqueeze(1), (pad, pad), mode='reflect').squeeze(1)
    spec = torch.stft(y, n_fft, hop_length=hop_size, win_length=win_size, window=_hann[str(win_size) + '_' + str(y.device)],
                      center=center, pad_mode='reflect', normalized=False, onesided=True, return_complex=True)
    spec = torch.sqrt(spec.real.pow(2) + spec.imag.pow(2) + 1e-9)
    spec = torch.matmul(_mel_basis[dk], spec)
    return torch.log(torch.clamp(spec, min=1e-5))


def install(monkeypatch, recurrences=True, mel=True):
    """every native piece of ttscube_amd.networks.training -> torch ops (undone by monkeypatch at the end of the test)"""
    from ttscube_amd.io_utils import melspec as MS
    from ttscube_amd.networks import training as T
    for name in ('_text_ops', '_make_adamw', '_gan_loss_fns', '_discriminator_fns', '_lowres_features', '_output_linears'):
        monkeypatch.setattr(T, name, globals()[name])
    monkeypatch.setattr(T, 'generator_forward_with_grad', generator_forward_train)
    if recurrences:
        monkeypatch.setattr(T, 'lstm_forward_train', lambda rnn, x: rnn(x)[0])
        monkeypatch.setattr(T, 'gru_forward_train', lambda rnn, x: rnn(x)[0])
    if mel:
        monkeypatch.setattr(MS, 'mel_spectrogram', mel_spectrogram)
