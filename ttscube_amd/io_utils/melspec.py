"""Mel-spectrogram on the HIP kernels: the DFT and the mel projection are MFMA GEMMs (ttsc_linear_forward), magnitude / log /
overlap-add are small element-wise kernels (csrc/stft.hip).  Serves

  * `mel_spectrogram(y, n_fft, num_mels, sampling_rate, hop_size, win_size, fmin, fmax)` — hifigan.meldataset.mel_spectrogram
    [EXTERNAL; cube/networks/cubegan.py:21,137-138,247-248]: reflect-pad (n_fft - hop)/2, magnitude sqrt(re^2+im^2+1e-9),
    natural log of clamp(., 1e-5); differentiable (the generated waveform carries the gradient of the 45 x mel-L1 loss);
  * `MelVocoder.melspectrogram` (cube/io_utils/vocoder.py:54-98) through io_utils/vocoder.py: centred frames (reflect-pad
    n_fft/2), |.|, log10(max(1e-5, .)).

No torch.stft / torch.matmul on this path; torch only pads and allocates."""
import ctypes as C
import math

import numpy as np
import torch

from .. import _lib

_cache = {}


def mel_filterbank(sr, n_fft, n_mels, fmin=0.0, fmax=None):
    """librosa.filters.mel defaults (Slaney scale, slaney norm) restated in numpy — librosa is not in this image."""
    fmax = sr / 2.0 if fmax is None else fmax

    def hz_to_mel(f):
        f = np.asarray(f, dtype=np.float64)
        f_sp = 200.0 / 3
        min_log_hz = 1000.0
        min_log_mel = min_log_hz / f_sp
        logstep = np.log(6.4) / 27.0
        return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-10) / min_log_hz) / logstep, f / f_sp)

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
        w[i] = np.maximum(0, np.minimum(-ramps[i] / fdiff[i], ramps[i + 2] / fdiff[i + 1]))
    w *= (2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels]))[:, None]
    return w.astype(np.float32)


def _bases(n_fft, win, sr, n_mels, fmin, fmax, dev):
    key = (n_fft, win, sr, n_mels, fmin, fmax, str(dev))
    if key not in _cache:
        nb = n_fft // 2 + 1
        ldm = (nb + 3) // 4 * 4
        k = np.arange(n_fft, dtype=np.float64)
        hann = np.zeros(n_fft)
        off = (n_fft - win) // 2
        hann[off:off + win] = 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(win) / win)        # periodic Hann, as torch / librosa
        ang = 2.0 * np.pi * np.outer(np.arange(nb, dtype=np.float64), k) / n_fft
        dft = np.concatenate([np.cos(ang) * hann, -np.sin(ang) * hann], axis=0).astype(np.float32)   # [2 nb, n_fft]
        mel = np.zeros((n_mels, ldm), dtype=np.float32)
        mel[:, :nb] = mel_filterbank(sr, n_fft, n_mels, fmin, fmax)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        _cache[key] = (t(dft), t(dft.T), t(mel), t(mel.T), nb, ldm)
    return _cache[key]


def _gemm(x_ptr, w, y, M, N, K, ldx, stream):
    _lib.check(_lib.lib().ttsc_linear_forward(x_ptr, _lib.dev_ptr(w), None, _lib.dev_ptr(y), M, N, K, ldx, N, _lib.ACT_NONE, 0, stream),
               'ttsc_linear_forward')


class _MelFn(torch.autograd.Function):
    """y_pad [B, Lp] (already padded) -> scale * log(max(mel_basis . |STFT|, minv)) as [B, n_mels, F]"""

    @staticmethod
    def forward(ctx, yp, n_fft, hop, win, sr, n_mels, fmin, fmax, eps, minv, scale):
        L = _lib.lib()
        yp = yp.float().contiguous()
        B, Lp = yp.shape
        F_ = (Lp - n_fft) // hop + 1
        dft, dft_t, mel, mel_t, nb, ldm = _bases(n_fft, win, sr, n_mels, fmin, fmax, yp.device)
        # Frames of one utterance sit at a constant stride `hop` inside its padded signal: the DFT reads them in place (no gather).  A BATCH is laid
        # out so that this holds across utterances too — every signal padded with zeros to a whole number Fp of hops, one zeroed n_fft tail behind the
        # last — and all B * Fp row positions go through ONE GEMM launch (round 5 looped over the utterances: 16 launches of 50 rows each per
        # spectrogram at the training step's b = 16, ~85 us apiece on a dozen workgroups, 2.7 ms of the step's main stream).  Rows F_ .. Fp - 1 of an
        # utterance straddle its end: they are computed, carried through the element-wise passes and cut off at the end; their gradient is zero.
        Fp = F_ if B == 1 else -(-Lp // hop)
        if Fp == F_:
            sig = yp
        else:
            sig = torch.zeros(B * Fp * hop + n_fft, dtype=torch.float32, device=yp.device)
            sig[:B * Fp * hop].view(B, Fp * hop)[:, :Lp].copy_(yp)
        reim = torch.empty((B, Fp, 2 * nb), dtype=torch.float32, device=yp.device)
        mag = torch.empty((B, Fp, ldm), dtype=torch.float32, device=yp.device)
        lin = torch.empty((B, Fp, n_mels), dtype=torch.float32, device=yp.device)
        out = torch.empty((B, Fp, n_mels), dtype=torch.float32, device=yp.device)
        with _lib.on_device(yp.device):
            s = _lib.current_stream()
            if Fp == F_:
                for b in range(B):
                    _gemm(C.c_void_p(sig[b].data_ptr()), dft, reim[b], F_, 2 * nb, n_fft, hop, s)
            else:
                _gemm(C.c_void_p(sig.data_ptr()), dft, reim, B * Fp, 2 * nb, n_fft, hop, s)
            _lib.check(L.ttsc_stft_mag(_lib.dev_ptr(reim), B * Fp, nb, ldm, eps, _lib.dev_ptr(mag), s), 'ttsc_stft_mag')
            _gemm(_lib.dev_ptr(mag), mel, lin, B * Fp, n_mels, ldm, ldm, s)
            _lib.check(L.ttsc_log_clamp(_lib.dev_ptr(lin), lin.numel(), minv, scale, _lib.dev_ptr(out), s), 'ttsc_log_clamp')
        ctx.save_for_backward(reim, mag, lin)
        ctx.cfg = (n_fft, hop, win, sr, n_mels, fmin, fmax, minv, scale, Lp, F_)
        return out[:, :F_].permute(0, 2, 1)

    @staticmethod
    def backward(ctx, g):
        L = _lib.lib()
        reim, mag, lin = ctx.saved_tensors
        n_fft, hop, win, sr, n_mels, fmin, fmax, minv, scale, Lp, F_ = ctx.cfg
        B, Fp, _ = reim.shape
        dft, dft_t, mel, mel_t, nb, ldm = _bases(n_fft, win, sr, n_mels, fmin, fmax, reim.device)
        g = g.permute(0, 2, 1).float()                                    # [B, F_, n_mels]
        if Fp != F_:
            g = torch.nn.functional.pad(g, (0, 0, 0, Fp - F_))            # the straddling rows carry no gradient
        g = g.contiguous()
        dlin = torch.empty_like(lin)
        dmag = torch.empty_like(mag)
        dreim = torch.empty_like(reim)
        dfr = torch.empty((B, Fp, n_fft), dtype=torch.float32, device=reim.device)
        Lo = max(Lp, (Fp - 1) * hop + n_fft)
        dy = torch.empty((B, Lo), dtype=torch.float32, device=reim.device)
        with _lib.on_device(reim.device):
            s = _lib.current_stream()
            _lib.check(L.ttsc_log_clamp_backward(_lib.dev_ptr(g), _lib.dev_ptr(lin), lin.numel(), minv, scale, _lib.dev_ptr(dlin), s), 'log_bwd')
            _gemm(_lib.dev_ptr(dlin), mel_t, dmag, B * Fp, ldm, n_mels, n_mels, s)                 # dmag = dlin . mel_basis
            _lib.check(L.ttsc_stft_mag_backward(_lib.dev_ptr(dmag), _lib.dev_ptr(reim), _lib.dev_ptr(mag), B * Fp, nb, ldm, _lib.dev_ptr(dreim), s),
                       'mag_bwd')
            _gemm(_lib.dev_ptr(dreim), dft_t, dfr, B * Fp, n_fft, 2 * nb, 2 * nb, s)               # dframes = d(re|im) . basis
            _lib.check(L.ttsc_overlap_add(_lib.dev_ptr(dfr), B, Fp, n_fft, hop, Lo, _lib.dev_ptr(dy), s), 'overlap_add')
        return (dy[:, :Lp],) + (None,) * 10


def mel_spectrogram(y, n_fft, num_mels, sampling_rate, hop_size, win_size, fmin, fmax, center=False):
    """hifigan.meldataset.mel_spectrogram: y [B, L] in [-1, 1] -> [B, num_mels, frames] natural-log mel (differentiable)."""
    if not y.is_cuda:
        raise _lib.TTSCError('mel_spectrogram: input must live on a HIP device; no CPU path')
    pad = int((n_fft - hop_size) / 2)
    yp = torch.nn.functional.pad(y.unsqueeze(1), (pad, pad), mode='reflect').squeeze(1)
    return _MelFn.apply(yp, n_fft, hop_size, win_size, sampling_rate, num_mels, float(fmin), float(fmax) if fmax is not None else None, 1e-9, 1e-5,
                        1.0)


def melspectrogram_log10(y, sample_rate=24000, num_mels=80, hop_size=240, n_fft=1024):
    """MelVocoder.melspectrogram (vocoder.py:54-63,70-98): librosa.stft(center=True, reflect padding), |.|, mel basis over
    [0, sr/2], log10(max(1e-5, .)).  y [B, L] on a HIP device -> [B, frames, num_mels] (frames = 1 + L // hop)."""
    if not y.is_cuda:
        raise _lib.TTSCError('melspectrogram_log10: input must live on a HIP device; no CPU path')
    yp = torch.nn.functional.pad(y.unsqueeze(1).float(), (n_fft // 2, n_fft // 2), mode='reflect').squeeze(1)
    with torch.no_grad():
        m = _MelFn.apply(yp, n_fft, hop_size, n_fft, sample_rate, num_mels, 0.0, None, 0.0, 1e-5, 1.0 / math.log(10.0))
    return m.permute(0, 2, 1).contiguous()
