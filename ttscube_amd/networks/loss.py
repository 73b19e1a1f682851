"""Output distributions of the vocoder — the host-side helper API of cube/networks/loss.py (encode / decode / loss /
sample_size / stats) for all five outputs.  On the GPU every sampler lives inside the persistent WaveRNN kernel
(csrc/wavernn.hip; definitions shared with the oracle in include/ttscube_math.h); the classes here carry what the
training step needs (the losses over the kernel's teacher-forced outputs) and the noise layout the kernel expects in
mode='noise' (`noise_width`, WR kind id `kind`).

Only the reference's public method names and numeric definitions are kept; the formulations are this package's own and
are pinned to the reference's values by tests/golden/losses_kat.npz (tools/gen_golden_losses.py imports the reference)."""
import math

import numpy as np
import torch
import torch.nn.functional as F

from .. import _lib

def _class_nll(logits, target):
    """mean negative log-likelihood of integer targets under softmax(logits); logits [B, L, S], target [B, L]."""
    return F.cross_entropy(logits.flatten(0, 1), target.flatten())


class _Waveform:
    """continuous outputs model the waveform value itself: the codec is the identity"""
    stats = (6e-6, 0.15)

    def encode(self, x):
        return x

    def decode(self, x):
        return x


class GaussianOutput(_Waveform):
    """cube/networks/loss.py:35-66: y_hat = (mean, log std); Gaussian negative log-likelihood
    0.5 * (log 2 pi + 2 log_std + (y - mean)^2 / std^2), log_std floored at log_std_min."""
    kind, noise_width, sample_size = _lib.WR_OUT_GM, 1, 2

    def loss(self, y_hat, y, log_std_min=-14.0):
        mean, log_std = y_hat[..., 0], y_hat[..., 1].clamp_min(log_std_min)
        z2 = (y - mean).square() * torch.exp(-2.0 * log_std)
        return (0.5 * (z2 + 2.0 * log_std + math.log(2.0 * math.pi))).mean()


class BetaOutput(_Waveform):
    """cube/networks/loss.py:69-106: y_hat = (log alpha, log beta) of a Beta law over (y + 1) / 2."""
    kind, noise_width, sample_size = _lib.WR_OUT_BETA, 18, 2

    def loss(self, y_hat, y):
        la, lb = y_hat[..., 0], y_hat[..., 1]
        a, b = la.exp(), lb.exp()
        u = ((y + 1.0) * 0.5).clamp(1e-5, 0.99999)
        log_norm = torch.lgamma(a) + torch.lgamma(b) - torch.lgamma(a + b)
        log_pdf = torch.xlogy(a - 1.0, u) + torch.xlogy(b - 1.0, 1.0 - u) - log_norm
        return -log_pdf.mean()


class MOLOutput(_Waveform):
    """cube/networks/loss.py:109-215: discretized mixture of logistics (the reference's default output), y_hat =
    (mixture logits | means | log scales).  A bin of half-width h = 1/(num_classes-1) around y has, per component,
        log P = log sigmoid(s (y - m + h))                   at the left edge   (y < -0.999)
                log sigmoid(-s (y - m - h))                  at the right edge  (y >  0.999)
                log(sigmoid(s (y-m+h)) - sigmoid(s (y-m-h))) inside, or — when that difference underflows (< 1e-5) — the
                                                             logistic density at the bin centre times the bin width,
    s = exp(-log_scale).  The mixture is summed with logsumexp over log_softmax(mixture logits)."""
    kind, noise_width, sample_size = _lib.WR_OUT_MOL, 11, 30

    def loss(self, y_hat, y, num_classes=65536, log_scale_min=None):
        floor = math.log(1e-14) if log_scale_min is None else log_scale_min
        logit_pi, mu, log_s = torch.chunk(y_hat, 3, dim=-1)
        log_s = log_s.clamp_min(floor)
        h = 1.0 / (num_classes - 1)
        t = y.unsqueeze(-1)
        rate = torch.exp(-log_s)
        up, dn, mid = rate * (t - mu + h), rate * (t - mu - h), rate * (t - mu)
        mass = torch.sigmoid(up) - torch.sigmoid(dn)
        # logistic log-density at the centre (log sigmoid(m) + log sigmoid(-m) - log scale) times the bin width 2 / (num_classes - 1)
        centre = F.logsigmoid(mid) + F.logsigmoid(-mid) - log_s - math.log((num_classes - 1) / 2)
        inside = torch.where(mass > 1e-5, torch.log(mass.clamp_min(1e-12)), centre)
        log_p = torch.where(t < -0.999, F.logsigmoid(up), torch.where(t > 0.999, F.logsigmoid(-dn), inside))
        return -torch.logsumexp(log_p + F.log_softmax(logit_pi, dim=-1), dim=-1).mean()


class MULAWOutput:
    """cube/networks/loss.py:218-277: 8-bit µ-law classes.  Companding q(x) = sign(x) log(1 + 255 |x|) / log 256 in [-1, 1];
    class = floor((q + 1) / 2 * 255 + 0.5) clipped to 0..255; decode inverts q on the class centres (the kernel uses the 256-entry
    table captured from the reference, include/ttscube_mulaw_lut.h)."""
    kind, noise_width, sample_size, stats = _lib.WR_OUT_MULAW, 256, 256, (-0.019, 0.51)

    def loss(self, y_hat, y):
        return _class_nll(y_hat, self.encode(y))

    def encode(self, x):
        if isinstance(x, np.ndarray):
            q = np.sign(x) * np.log1p(255 * np.abs(x)) / np.log1p(255)
            return np.clip(((q + 1) / 2 * 255 + 0.5).astype(int), 0, 255)
        x = x.float()
        q = torch.sign(x) * torch.log1p(x.abs() * 255.0) / torch.log1p(x.new_full((1,), 255.0))
        return ((q + 1) / 2 * 255.0 + 0.5).long().clamp(0, 255)

    def decode(self, x_mu):
        if isinstance(x_mu, np.ndarray):
            q = x_mu / 255. * 2 - 1.
            return np.sign(q) * (np.exp(np.abs(q) * np.log1p(255.)) - 1.) / 255.
        q = x_mu.float() / 255. * 2 - 1.
        return torch.sign(q) * (torch.exp(q.abs() * float(np.log1p(255.))) - 1.) / 255.


class RAWOutput:
    """cube/networks/loss.py:280-307: 256 uniform classes over [-1, 1]."""
    kind, noise_width, sample_size, stats = _lib.WR_OUT_RAW, 256, 256, (-0.019, 0.15)

    def loss(self, y_hat, y):
        return _class_nll(y_hat, self.encode(y))

    def encode(self, x):
        return ((x + 1.0) / 2 * 255).clamp(0, 255).long()

    def decode(self, x):
        return (x / 255 - 0.5) * 2
