"""CPU: the mel-spectrogram oracle (numpy restatement of vocoder.py:54-98 and of hifigan's mel_spectrogram) against
torch.stft — an independent STFT — and known properties of the Slaney filterbank."""
import numpy as np
import pytest
import torch

from oracle import melspec_ref as M


def test_oracle_stft_matches_torch_stft():
    rng = np.random.RandomState(0)
    y = rng.uniform(-1, 1, size=5000)
    yp = np.pad(y, 512, mode='reflect')
    mag = M.stft_mag(yp, 1024, 240)
    z = torch.stft(torch.from_numpy(y), 1024, hop_length=240, win_length=1024, window=torch.hann_window(1024, dtype=torch.float64),
                   center=True, pad_mode='reflect', return_complex=True)
    assert mag.shape == tuple(z.shape) == (513, 1 + 5000 // 240)
    assert float(np.abs(mag - z.abs().numpy()).max()) < 1e-9


def test_mel_filterbank_properties_and_log_floor():
    w = M.mel_filterbank(24000, 1024, 80)
    assert w.shape == (80, 513) and (w >= 0).all()
    peaks = w.argmax(axis=1)
    assert (np.diff(peaks) > 0).all()                       # centre frequencies increase
    assert abs(w[0].sum() * (12000 / 512) - 1.0) < 0.35    # slaney norm: roughly unit area per band (Hz)
    m = M.melspectrogram_log10(np.zeros(2400))
    assert m.shape == (11, 80) and np.allclose(m, -5.0)     # silence hits the floor log10(1e-5)
    y = 0.5 * np.sin(2 * np.pi * 1000.0 * np.arange(24000) / 24000.0)
    m = M.melspectrogram_log10(y)
    assert m.shape == (101, 80)
    band = int(np.argmax(m[50]))
    lo, hi = np.flatnonzero(w[band])[[0, -1]] * (24000 / 1024)
    assert lo <= 1000.0 <= hi
    ln = M.mel_spectrogram_ln(y, 1024, 80, 24000, 240, 1024, 0, 12000)
    assert ln.shape == (80, 24000 // 240) and int(np.argmax(ln[:, 50])) == band


def test_oracle_matches_an_independent_librosa_restatement():
    """librosa itself is not in the image (the oracle's header says "parity unpinned" for that part).  `transformers.audio_utils` is an
    independent restatement of librosa's STFT / Slaney mel filterbank pipeline: the oracle's filterbank and both of its
    mel-spectrogram definitions must agree with it (same cross-validation role SpeechT5HifiGan plays for the generator oracle)."""
    au = pytest.importorskip('transformers.audio_utils')
    from oracle import melspec_ref as M
    fb = au.mel_filter_bank(num_frequency_bins=513, num_mel_filters=80, min_frequency=0, max_frequency=12000, sampling_rate=24000,
                            norm='slaney', mel_scale='slaney')
    assert float(np.abs(M.mel_filterbank(24000, 1024, 80, 0, 12000) - fb.T).max()) < 1e-8
    rng = np.random.RandomState(0)
    x = (0.3 * np.sin(np.cumsum(rng.uniform(0.01, 0.3, size=6000))) + 0.01 * rng.randn(6000)).astype(np.float64)
    win = au.window_function(1024, 'hann', periodic=True)
    # MelVocoder.melspectrogram (cube/io_utils/vocoder.py:54-98): centred STFT, log10 with a 1e-5 floor
    ref = au.spectrogram(x, win, frame_length=1024, hop_length=240, fft_length=1024, power=1.0, center=True, pad_mode='reflect',
                         mel_filters=fb, log_mel='log10', mel_floor=1e-5).T
    got = M.melspectrogram_log10(x)
    n = min(len(ref), len(got))
    assert n >= 25 and float(np.abs(got[:n] - ref[:n]).max()) < 1e-5
    # hifigan.meldataset.mel_spectrogram: reflect padding (n_fft - hop) / 2, no centring, natural log with a 1e-5 clamp
    pad = (1024 - 240) // 2
    lin = au.spectrogram(np.pad(x, (pad, pad), mode='reflect'), win, frame_length=1024, hop_length=240, fft_length=1024, power=1.0,
                         center=False, mel_filters=fb, log_mel=None, mel_floor=1e-30)
    got2 = M.mel_spectrogram_ln(x, 1024, 80, 24000, 240, 1024, 0, 12000)
    n = min(lin.shape[1], got2.shape[1])
    assert n >= 25 and float(np.abs(got2[:, :n] - np.log(np.maximum(lin[:, :n], 1e-5))).max()) < 1e-5
