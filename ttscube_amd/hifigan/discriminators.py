"""HiFi-GAN discriminators and GAN losses — drop-in for the reference's un-vendored ``hifigan.models``
(`MultiPeriodDiscriminator`, `MultiScaleDiscriminator`, `feature_loss`, `generator_loss`, `discriminator_loss`;
imported at cube/networks/cubegan.py:18-19, called at cubegan.py:144-167 and :243-260) and ``hifigan.meldataset.mel_spectrogram``
(cubegan.py:21,137-138).  The reference source is absent: the layer tables follow Kong, Kim, Bae 2020
(arXiv:2010.05646, §2.2-2.3, App. A) with the public module / key layout (`discriminators.N.convs.M`, `conv_post`), so reference
checkpoints load key for key.

The classes are parameter containers (torch's weight-norm / spectral-norm parametrisation, so `state_dict()` carries `weight_g` /
`weight_v` / `weight_orig` / `weight_u` as the reference's does) whose ``forward`` runs on the HIP kernels by itself, the way
``Generator.forward`` does: every convolution — forward, data gradient, weight gradient — on `conv_mfma_kernel` / the split-precision
training kernels (hifigan/disc_hip.py), the three losses on `gan_loss_kernel` (hifigan/losses_hip.py), the mel-spectrogram on the DFT /
mel GEMMs (io_utils/melspec.py).  A training step written in the reference's own shape (module calls and these loss functions, cubegan.py:131-176)
therefore runs the same launches as `networks/training.py::cubegan_training_step`.  There is no CPU path: a CPU tensor raises TTSCError.  The torch-op
formulation these are tested against lives in tests/torch_reference.py."""
import torch
import torch.nn as nn
from torch.nn.utils import spectral_norm, weight_norm

from .. import _lib
from ..io_utils.melspec import mel_spectrogram          # noqa: F401  (hifigan.meldataset.mel_spectrogram, cubegan.py:21)
from .losses_hip import discriminator_loss, feature_loss, generator_loss   # noqa: F401  (hifigan.models' three losses)

LRELU_SLOPE = 0.1


def get_padding(kernel_size, dilation=1):
    return int((kernel_size * dilation - dilation) / 2)


def _require_hip(x, who):
    if not x.is_cuda:
        raise _lib.TTSCError('%s: input must live on a HIP device (got %s); the discriminators have no CPU path' % (who, x.device))


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
        """x [B, 1, T] -> (scores [B, H' * period], feature maps [B, C, H, period]); the period fold is implicit (disc_hip: dilation-`period`
        convolutions over the flat signal)"""
        from . import disc_hip
        _require_hip(x, 'DiscriminatorP')
        return disc_hip._run(self, 'p', disc_hip._fold(x, self.period), True)


class MultiPeriodDiscriminator(nn.Module):
    def __init__(self):
        super().__init__()
        self.discriminators = nn.ModuleList([DiscriminatorP(p) for p in (2, 3, 5, 7, 11)])

    def forward(self, y, y_hat, want_fmap=True):
        """(y_d_rs, y_d_gs, fmap_rs, fmap_gs), cubegan.py:144,160.  want_fmap=False (not in the reference) skips materialising the activated
        feature maps when the caller discards them (the discriminator step does)."""
        from .disc_hip import mpd_forward
        _require_hip(y, 'MultiPeriodDiscriminator')
        return mpd_forward(self, y, y_hat, want_fmap=want_fmap)


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
        from . import disc_hip
        _require_hip(x, 'DiscriminatorS')
        return disc_hip._run(self, 's', x, True)


class MultiScaleDiscriminator(nn.Module):
    def __init__(self):
        super().__init__()
        self.discriminators = nn.ModuleList([DiscriminatorS(use_spectral_norm=True), DiscriminatorS(), DiscriminatorS()])
        self.meanpools = nn.ModuleList([nn.AvgPool1d(4, 2, padding=2), nn.AvgPool1d(4, 2, padding=2)])

    def forward(self, y, y_hat, want_fmap=True):
        from .disc_hip import msd_forward
        _require_hip(y, 'MultiScaleDiscriminator')
        return msd_forward(self, y, y_hat, want_fmap=want_fmap)
