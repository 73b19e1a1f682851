"""ORACLE (test infrastructure — never imported by the product path): numpy float64 restatement of the two mel-spectrogram
definitions on the path.

  melspectrogram_log10   cube/io_utils/vocoder.py:54-98 (MelVocoder.melspectrogram): librosa.stft(n_fft=1024, hop, win=1024, hann,
                         center=True, reflect padding) -> |.| -> librosa.filters.mel(sr, 1024, 80) -> log10(max(1e-5, .))
  mel_spectrogram_ln     hifigan.meldataset.mel_spectrogram [EXTERNAL, published implementation]: reflect-pad (n_fft-hop)/2,
                         hann STFT, sqrt(re^2 + im^2 + 1e-9), mel basis [fmin, fmax], ln(clamp(., 1e-5))

PARITY UNPINNED for the librosa-dependent part: librosa is not installed here and the reference holds no fixture for it; the
STFT / Slaney filterbank are restated from librosa's documented definitions (periodic Hann, Slaney mel scale with slaney
normalisation).  Cross-checked in tests/test_melspec_cpu.py against torch.stft and against `transformers.audio_utils` (an independent
restatement of librosa's STFT / Slaney filterbank pipeline): filterbank 1e-9, both mel-spectrogram definitions 1e-7."""
import numpy as np


def _hann(n):
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / n)


def mel_filterbank(sr, n_fft, n_mels, fmin=0.0, fmax=None):
    fmax = sr / 2.0 if fmax is None else fmax
    f_sp, min_log_hz = 200.0 / 3, 1000.0
    min_log_mel, logstep = min_log_hz / f_sp, np.log(6.4) / 27.0
    h2m = lambda f: np.where(np.asarray(f, float) >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-10) / min_log_hz) / logstep,
                             np.asarray(f, float) / f_sp)
    m2h = lambda m: np.where(np.asarray(m, float) >= min_log_mel, min_log_hz * np.exp(logstep * (np.asarray(m, float) - min_log_mel)),
                             f_sp * np.asarray(m, float))
    freqs = np.linspace(0, sr / 2.0, 1 + n_fft // 2)
    mel_f = m2h(np.linspace(h2m(fmin), h2m(fmax), n_mels + 2))
    fd = np.diff(mel_f)
    ramps = mel_f[:, None] - freqs[None, :]
    w = np.zeros((n_mels, 1 + n_fft // 2))
    for i in range(n_mels):
        w[i] = np.maximum(0, np.minimum(-ramps[i] / fd[i], ramps[i + 2] / fd[i + 1]))
    return w * (2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels]))[:, None]


def stft_mag(yp, n_fft, hop, eps=0.0):
    """yp: padded 1-D signal -> |STFT| [n_fft/2+1, frames] (float64)"""
    F = (len(yp) - n_fft) // hop + 1
    fr = np.stack([yp[f * hop:f * hop + n_fft] for f in range(F)]) * _hann(n_fft)[None, :]
    z = np.fft.rfft(fr, axis=1)
    return np.sqrt(z.real ** 2 + z.imag ** 2 + eps).T


def melspectrogram_log10(y, sample_rate=24000, num_mels=80, hop_size=240, n_fft=1024):
    yp = np.pad(np.asarray(y, dtype=np.float64), n_fft // 2, mode='reflect')
    mag = stft_mag(yp, n_fft, hop_size)
    return np.log10(np.maximum(1e-5, mel_filterbank(sample_rate, n_fft, num_mels) @ mag)).T       # [frames, num_mels]


def mel_spectrogram_ln(y, n_fft, num_mels, sampling_rate, hop_size, win_size, fmin, fmax):
    assert win_size == n_fft
    pad = int((n_fft - hop_size) / 2)
    yp = np.pad(np.asarray(y, dtype=np.float64), pad, mode='reflect')
    mag = stft_mag(yp, n_fft, hop_size, eps=1e-9)
    return np.log(np.maximum(mel_filterbank(sampling_rate, n_fft, num_mels, fmin, fmax) @ mag, 1e-5))   # [num_mels, frames]
