"""Waveform file I/O for the dataset readers: what the reference does with `librosa.load(path, sr=...)` (cube/io_utils/
io_cubegan.py:97, io_vocoder.py:52-53) done with scipy (librosa is not in this image): PCM / float WAV -> mono float32 in
[-1, 1], polyphase resampling to the requested rate."""
from math import gcd

import numpy as np
import scipy.io.wavfile
import scipy.signal


def load_wav(path, sr):
    rate, x = scipy.io.wavfile.read(path)
    if x.dtype == np.int16:
        x = x.astype(np.float32) / 32768.0
    elif x.dtype == np.int32:
        x = x.astype(np.float32) / 2147483648.0
    elif x.dtype == np.uint8:
        x = (x.astype(np.float32) - 128.0) / 128.0
    else:
        x = x.astype(np.float32)
    if x.ndim > 1:
        x = x.mean(axis=1)
    if rate != sr:
        g = gcd(int(rate), int(sr))
        x = scipy.signal.resample_poly(x, sr // g, rate // g).astype(np.float32)
    return x, sr


def save_wav(path, audio, sr=24000):
    """int16 PCM, as scipy.io.wavfile.write(..., 24000, int16) in cube/io_utils/runtime.py:80,109"""
    a = np.asarray(audio)
    if a.dtype != np.int16:
        a = np.asarray(np.clip(a, -1.0, 1.0) * 32767, dtype=np.int16)
    scipy.io.wavfile.write(path, sr, a)
