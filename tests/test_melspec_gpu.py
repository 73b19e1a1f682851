"""GPU: mel-spectrogram on the HIP kernels (DFT + mel projection as MFMA GEMMs, csrc/stft.hip element-wise kernels) against the
numpy oracle (both definitions on the path) and, for the backward pass, against torch autograd over the torch formulation."""
import numpy as np
import pytest
import torch

from oracle import melspec_ref as M

pytestmark = pytest.mark.gpu


def _audio(B, L, seed):
    rng = np.random.RandomState(seed)
    t = np.arange(L) / 24000.0
    y = np.stack([0.4 * np.sin(2 * np.pi * rng.uniform(80, 4000) * t + rng.uniform(0, 6)) + 0.05 * rng.randn(L) for _ in range(B)])
    y[:, : L // 7] *= 0.0   # leading silence: exercises the log floor
    return y.astype(np.float32)


@pytest.mark.parametrize('B,L', [(1, 24000), (3, 12000), (2, 5003)])
def test_feature_extraction_matches_oracle(B, L):
    """MelVocoder.melspectrogram (cube/io_utils/vocoder.py:54-98): log10 mel, frames = 1 + L // hop"""
    from ttscube_amd.io_utils.vocoder import MelVocoder
    y = _audio(B, L, 1)
    got = MelVocoder().melspectrogram(y, sample_rate=24000, num_mels=80, hop_size=240)
    assert got.shape == (B, 1 + L // 240, 80) and got.dtype == np.float32
    for b in range(B):
        want = M.melspectrogram_log10(y[b])
        assert float(np.abs(got[b] - want).max()) < 2e-4
    one = MelVocoder().melspectrogram(y[0], sample_rate=24000, num_mels=80, hop_size=240)
    assert np.array_equal(one, got[0])


def test_loss_mel_forward_and_backward():
    """hifigan mel_spectrogram (cubegan.py:137-138): forward vs the oracle, gradient of the 45 x L1 loss vs torch autograd over
    the torch.stft formulation (tests/torch_reference.py::mel_spectrogram)."""
    from tests.torch_reference import mel_spectrogram as mel_torch
    from ttscube_amd.io_utils.melspec import mel_spectrogram as mel_hip
    y = torch.from_numpy(_audio(4, 12000, 2)).cuda()
    tgt = torch.from_numpy(_audio(4, 12000, 3)).cuda()
    a = y.clone().requires_grad_(True)
    b = y.clone().requires_grad_(True)
    ma = mel_hip(a, 1024, 80, 24000, 240, 1024, 0, 12000)
    mb = mel_torch(b, 1024, 80, 24000, 240, 1024, 0, 12000)
    assert ma.shape == mb.shape == (4, 80, 50)
    want = np.stack([M.mel_spectrogram_ln(y[i].cpu().numpy(), 1024, 80, 24000, 240, 1024, 0, 12000) for i in range(4)])
    assert float(np.abs(ma.detach().cpu().numpy() - want).max()) < 5e-4
    assert float((ma - mb).abs().max()) < 5e-4
    with torch.no_grad():
        mt = mel_torch(tgt, 1024, 80, 24000, 240, 1024, 0, 12000)
    (torch.nn.functional.l1_loss(ma, mt) * 45).backward()
    (torch.nn.functional.l1_loss(mb, mt) * 45).backward()
    ga, gb = a.grad, b.grad
    rel = float((ga - gb).pow(2).mean().sqrt() / gb.pow(2).mean().sqrt())
    assert rel < 2e-3, rel     # sign(.) of the L1 loss flips on near-ties between the two forward paths: a few frames differ
    cos = float((ga * gb).sum() / (ga.norm() * gb.norm()))
    assert cos > 0.9999
