"""Output distributions of the vocoder — the host-side helper API of cube/networks/loss.py (encode / decode / loss /
sample_size / stats) for all five outputs.  On the GPU every sampler lives inside the persistent WaveRNN kernel
(csrc/wavernn.hip; definitions shared with the oracle in include/ttscube_math.h); the classes here carry what the
training step needs (the losses, as torch expressions over the kernel's teacher-forced outputs) and the noise layout
the kernel expects in mode='noise' (`noise_width`, WR kind id `kind`)."""
import math

import numpy as np
import torch
import torch.nn.functional as F
from torch.nn import CrossEntropyLoss

from .. import _lib


def log_sum_exp(x):
    m, _ = torch.max(x, dim=-1)
    m2, _ = torch.max(x, dim=-1, keepdim=True)
    return m + torch.log(torch.sum(torch.exp(x - m2), dim=-1))


class _Continuous:
    def encode(self, x):
        return x

    def decode(self, x):
        return x

    @property
    def stats(self):
        return 6e-6, 0.15


class GaussianOutput(_Continuous):
    """cube/networks/loss.py:35-66"""
    kind, noise_width = _lib.WR_OUT_GM, 1

    def loss(self, y_hat, y, log_std_min=-14.0):
        y = y.unsqueeze(2)
        mean = y_hat[:, :, :1]
        log_std = torch.clamp(y_hat[:, :, 1:], min=log_std_min)
        log_probs = -0.5 * (-math.log(2.0 * math.pi) - 2. * log_std - torch.pow(y - mean, 2) * torch.exp(-2.0 * log_std))
        return log_probs.squeeze().mean()

    @property
    def sample_size(self):
        return 2


class BetaOutput(_Continuous):
    """cube/networks/loss.py:69-106"""
    kind, noise_width = _lib.WR_OUT_BETA, 18

    def loss(self, y_hat, y):
        loc_y = y_hat.exp()
        dist = torch.distributions.Beta(loc_y[:, :, 0].unsqueeze(-1), loc_y[:, :, 1].unsqueeze(-1))
        y = torch.clamp((y + 1.0) / 2.0, 1e-5, 0.99999).unsqueeze(-1)
        return (-dist.log_prob(y).squeeze(-1)).mean()

    @property
    def sample_size(self):
        return 2


class MOLOutput(_Continuous):
    """cube/networks/loss.py:109-215 (discretized mixture of 10 logistics; the reference's default output)"""
    kind, noise_width = _lib.WR_OUT_MOL, 11

    def loss(self, y_hat, y, num_classes=65536, log_scale_min=None):
        if log_scale_min is None:
            log_scale_min = float(np.log(1e-14))
        nr_mix = y_hat.shape[2] // 3
        y = y.unsqueeze(2)
        logit_probs = y_hat[:, :, :nr_mix]
        means = y_hat[:, :, nr_mix:2 * nr_mix]
        log_scales = torch.clamp(y_hat[:, :, 2 * nr_mix:3 * nr_mix], min=log_scale_min)
        y = y.expand_as(means)
        centered_y = y - means
        inv_stdv = torch.exp(-log_scales)
        plus_in = inv_stdv * (centered_y + 1. / (num_classes - 1))
        cdf_plus = torch.sigmoid(plus_in)
        min_in = inv_stdv * (centered_y - 1. / (num_classes - 1))
        cdf_min = torch.sigmoid(min_in)
        log_cdf_plus = plus_in - F.softplus(plus_in)
        log_one_minus_cdf_min = -F.softplus(min_in)
        cdf_delta = cdf_plus - cdf_min
        mid_in = inv_stdv * centered_y
        log_pdf_mid = mid_in - log_scales - 2. * F.softplus(mid_in)
        inner_inner_cond = (cdf_delta > 1e-5).float()
        inner_inner_out = inner_inner_cond * torch.log(torch.clamp(cdf_delta, min=1e-12)) + \
            (1. - inner_inner_cond) * (log_pdf_mid - np.log((num_classes - 1) / 2))
        inner_cond = (y > 0.999).float()
        inner_out = inner_cond * log_one_minus_cdf_min + (1. - inner_cond) * inner_inner_out
        cond = (y < -0.999).float()
        log_probs = cond * log_cdf_plus + (1. - cond) * inner_out
        log_probs = log_probs + F.log_softmax(logit_probs, -1)
        return -torch.mean(log_sum_exp(log_probs))

    @property
    def sample_size(self):
        return 30


class MULAWOutput:
    """cube/networks/loss.py:218-277"""
    kind, noise_width = _lib.WR_OUT_MULAW, 256

    def __init__(self):
        self._loss = CrossEntropyLoss()

    def loss(self, y_hat, y):
        y = self.encode(y)
        return self._loss(y_hat.reshape(y_hat.shape[0] * y_hat.shape[1], -1), y.reshape(y.shape[0] * y.shape[1]))

    def encode(self, x):
        mu = 255
        if isinstance(x, np.ndarray):
            x_mu = np.sign(x) * np.log1p(mu * np.abs(x)) / np.log1p(mu)
            x_mu = ((x_mu + 1) / 2 * mu + 0.5).astype(int)
            return np.clip(x_mu, 0, 255)
        x = x.float()
        mu_t = torch.tensor([255.0], device=x.device)
        x_mu = torch.sign(x) * torch.log1p(mu_t * torch.abs(x)) / torch.log1p(mu_t)
        return torch.clip(((x_mu + 1) / 2 * mu_t + 0.5).long(), 0, 255)

    def decode(self, x_mu):
        mu = 255.
        if isinstance(x_mu, np.ndarray):
            x = (x_mu / mu) * 2 - 1.
            return np.sign(x) * (np.exp(np.abs(x) * np.log1p(mu)) - 1.) / mu
        x = (x_mu.float() / mu) * 2 - 1.
        return torch.sign(x) * (torch.exp(torch.abs(x) * float(np.log1p(mu))) - 1.) / mu

    @property
    def sample_size(self):
        return 256

    @property
    def stats(self):
        return -0.019, 0.51


class RAWOutput:
    """cube/networks/loss.py:280-307"""
    kind, noise_width = _lib.WR_OUT_RAW, 256

    def __init__(self):
        self._loss = CrossEntropyLoss()

    def loss(self, y_hat, y):
        y = self.encode(y)
        return self._loss(y_hat.reshape(y_hat.shape[0] * y_hat.shape[1], -1), y.reshape(y.shape[0] * y.shape[1]))

    def encode(self, x):
        return torch.clip(((x + 1.0) / 2) * 255, 0, 255).long()

    def decode(self, x):
        return ((x / 255) - 0.5) * 2

    @property
    def sample_size(self):
        return 256

    @property
    def stats(self):
        return -0.019, 0.15
