"""HiFi-GAN discriminators and GAN losses — stand-ins for the reference's un-vendored ``hifigan.models``
(`MultiPeriodDiscriminator`, `MultiScaleDiscriminator`, `feature_loss`, `generator_loss`, `discriminator_loss`;
imported at cube/networks/cubegan.py:18-19, used at cubegan.py:144-167) and ``hifigan.meldataset.mel_spectrogram``
(cubegan.py:21,137-138).  The reference source is absent: restated from Kong, Kim, Bae 2020
(arXiv:2010.05646, §2.2-2.3, App. A) with the public module/key layout (`discriminators.N.convs.M`, `conv_post`).
TRAINING-ONLY: these run as plain torch-ROCm modules; they are not on the inference hot path.

Launch diet (the b = 16 training step is host-bound: ~7 700 launches for ~105 ms of GPU work): when the generated signal carries
no gradient (the discriminator step) a weight-normed sub-discriminator sees real and generated audio as ONE batch — there are
no batch statistics, so outputs and gradients are those of two separate calls with half the launches (and one weight
normalisation instead of two).  The spectrally-normed discriminator keeps its two calls (each call advances its power
iteration, as in the reference).  In the generator step the real branch needs no graph and the generated branch must not pay
data gradients for the real half, so the two stay separate."""
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn.utils import spectral_norm, weight_norm

LRELU_SLOPE = 0.1


def _pair(d, y, y_hat, batch_ok):
    """(out_r, fmap_r, out_g, fmap_g) of sub-discriminator d on real / generated audio."""
    if batch_ok and not y_hat.requires_grad and y.shape == y_hat.shape:
        n = y.shape[0]
        out, fmap = d(torch.cat([y, y_hat], dim=0))
        return out[:n], [f[:n] for f in fmap], out[n:], [f[n:] for f in fmap]
    y_d_r, fmap_r = d(y)
    y_d_g, fmap_g = d(y_hat)
    return y_d_r, fmap_r, y_d_g, fmap_g


def get_padding(kernel_size, dilation=1):
    return int((kernel_size * dilation - dilation) / 2)


class DiscriminatorP(nn.Module):
    def __init__(self, period, kernel_size=5, stride=3, use_spectral_norm=False):
        super().__init__()
        self.period = period
        norm_f = spectral_norm if use_spectral_norm else weight_norm
        chans = [1, 32, 128, 512, 1024]
        self.convs = nn.ModuleList([norm_f(nn.Conv2d(chans[i], chans[i + 1], (kernel_size, 1), (stride, 1),
                                                     padding=(get_padding(5, 1), 0))) for i in range(4)]
                                   + [norm_f(nn.Conv2d(1024, 1024, (kernel_size, 1), 1, padding=(2, 0)))])
        self.conv_post = norm_f(nn.Conv2d(1024, 1, (3, 1), 1, padding=(1, 0)))

    def forward(self, x):
        fmap = []
        b, c, t = x.shape
        if t % self.period != 0:
            n_pad = self.period - (t % self.period)
            x = F.pad(x, (0, n_pad), 'reflect')
            t = t + n_pad
        x = x.view(b, c, t // self.period, self.period)
        for l in self.convs:
            x = F.leaky_relu(l(x), LRELU_SLOPE)
            fmap.append(x)
        x = self.conv_post(x)
        fmap.append(x)
        return torch.flatten(x, 1, -1), fmap


class MultiPeriodDiscriminator(nn.Module):
    def __init__(self):
        super().__init__()
        self.discriminators = nn.ModuleList([DiscriminatorP(p) for p in (2, 3, 5, 7, 11)])

    def forward(self, y, y_hat):
        y_d_rs, y_d_gs, fmap_rs, fmap_gs = [], [], [], []
        for d in self.discriminators:
            y_d_r, fmap_r, y_d_g, fmap_g = _pair(d, y, y_hat, True)
            y_d_rs.append(y_d_r)
            fmap_rs.append(fmap_r)
            y_d_gs.append(y_d_g)
            fmap_gs.append(fmap_g)
        return y_d_rs, y_d_gs, fmap_rs, fmap_gs


class DiscriminatorS(nn.Module):
    def __init__(self, use_spectral_norm=False):
        super().__init__()
        norm_f = spectral_norm if use_spectral_norm else weight_norm
        self.convs = nn.ModuleList([
            norm_f(nn.Conv1d(1, 128, 15, 1, padding=7)),
            norm_f(nn.Conv1d(128, 128, 41, 2, groups=4, padding=20)),
            norm_f(nn.Conv1d(128, 256, 41, 2, groups=16, padding=20)),
            norm_f(nn.Conv1d(256, 512, 41, 4, groups=16, padding=20)),
            norm_f(nn.Conv1d(512, 1024, 41, 4, groups=16, padding=20)),
            norm_f(nn.Conv1d(1024, 1024, 41, 1, groups=16, padding=20)),
            norm_f(nn.Conv1d(1024, 1024, 5, 1, padding=2)),
        ])
        self.conv_post = norm_f(nn.Conv1d(1024, 1, 3, 1, padding=1))

    def forward(self, x):
        fmap = []
        for l in self.convs:
            x = F.leaky_relu(l(x), LRELU_SLOPE)
            fmap.append(x)
        x = self.conv_post(x)
        fmap.append(x)
        return torch.flatten(x, 1, -1), fmap


class MultiScaleDiscriminator(nn.Module):
    def __init__(self):
        super().__init__()
        self.discriminators = nn.ModuleList([DiscriminatorS(use_spectral_norm=True), DiscriminatorS(), DiscriminatorS()])
        self.meanpools = nn.ModuleList([nn.AvgPool1d(4, 2, padding=2), nn.AvgPool1d(4, 2, padding=2)])

    def forward(self, y, y_hat):
        y_d_rs, y_d_gs, fmap_rs, fmap_gs = [], [], [], []
        for i, d in enumerate(self.discriminators):
            if i != 0:
                y = self.meanpools[i - 1](y)
                y_hat = self.meanpools[i - 1](y_hat)
            y_d_r, fmap_r, y_d_g, fmap_g = _pair(d, y, y_hat, i != 0)   # discriminator 0 is spectrally normed
            y_d_rs.append(y_d_r)
            fmap_rs.append(fmap_r)
            y_d_gs.append(y_d_g)
            fmap_gs.append(fmap_g)
        return y_d_rs, y_d_gs, fmap_rs, fmap_gs


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
        r_losses.append(r_loss.detach())   # tensors, not .item(): 16 host syncs per step would drain the launch queue
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
