"""Output distributions of the vocoder — mirror of cube/networks/loss.py for the discrete (µ-law / raw) outputs.

On the GPU the sampler lives inside the persistent WaveRNN kernel (Gumbel-max == Categorical(logits).sample(),
loss.py:227-230); this module keeps the reference's host-side helper API (encode / decode / loss / sample_size)."""
import numpy as np
import torch
from torch.nn import CrossEntropyLoss


class MULAWOutput:
    """cube/networks/loss.py:218-277"""

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
