"""CPU: the mel-spectrogram oracle (numpy restatement of vocoder.py:54-98 and of hifigan's mel_spectrogram) against
torch.stft — an independent STFT — and known properties of the Slaney filterbank."""
import numpy as np
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
