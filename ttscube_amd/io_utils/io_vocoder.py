"""Mirror of cube/io_utils/io_vocoder.py: ``VocoderDataset`` (:20-82 — wav folder -> (audio, low-rate audio, log10-mel) with the
`data/cache` files `<name>.mgc.npy / .audio.npy / .audio_low.npy`, random hop-aligned crops) and ``VocoderCollate`` (:85-112).
Differences: librosa.load -> scipy (io_utils/audio.py); the mel features come from the GPU (io_utils/vocoder.py::MelVocoder)."""
import os
import random

import numpy as np
import torch

from .audio import load_wav


class VocoderDataset:
    def __init__(self, path, target_sample_rate=24000, lowres_sample_rate=2400, max_segment_size=-1, random_start=True, hop_size=240,
                 cache_dir='data/cache', mel_vocoder=None):
        self._examples = []
        self._sample_rate = target_sample_rate
        self._sample_rate_low = lowres_sample_rate
        self._max_segment_size = max_segment_size
        self._mel_vocoder = mel_vocoder
        self._hop_size = hop_size
        self._random_start = random_start
        self._cache_dir = cache_dir
        for f in sorted(os.listdir(path)):
            full = os.path.join(path, f)
            if f.endswith('.wav') and os.path.isfile(full):
                w_size = os.stat(full).st_size
                if w_size > 4096 and w_size > max_segment_size * 2:
                    self._examples.append(full)
        os.makedirs(cache_dir, exist_ok=True)

    def __len__(self):
        return len(self._examples)

    def _features(self, filename):
        cache = os.path.join(self._cache_dir, filename.replace('/', '_').replace('\\', '_'))
        if os.path.exists(cache + '.mgc.npy'):
            return np.load(cache + '.audio.npy'), np.load(cache + '.audio_low.npy'), np.load(cache + '.mgc.npy')
        wav, _ = load_wav(filename, self._sample_rate)
        wav_low, _ = load_wav(filename, self._sample_rate_low)
        wav = (wav / np.max(np.abs(wav))) * 0.98
        wav_low = (wav_low / np.max(np.abs(wav_low))) * 0.98
        if self._mel_vocoder is None:
            from .vocoder import MelVocoder
            self._mel_vocoder = MelVocoder()
        mel = self._mel_vocoder.melspectrogram(wav, sample_rate=self._sample_rate, num_mels=80, hop_size=self._hop_size,
                                               use_preemphasis=False)
        np.save(cache + '.mgc', mel)
        np.save(cache + '.audio', wav)
        np.save(cache + '.audio_low', wav_low)
        return wav, wav_low, mel

    def __getitem__(self, item):
        wav, wav_low, mel = self._features(self._examples[item])
        ms, hs = self._max_segment_size, self._sample_rate // self._sample_rate_low
        if ms == -1 or len(wav) < ms or not self._random_start:
            if not self._random_start and ms != -1 and len(wav) > ms:
                return wav[:ms], wav_low[:ms // hs], mel[:ms // self._hop_size + 1]
            return wav, wav_low, mel
        start = random.randint(0, len(wav) - ms - 1)
        start = start // self._hop_size * self._hop_size   # multiple of the hop size
        stop = start + ms
        start_low = start // hs
        return wav[start:stop], wav_low[start_low:start_low + ms // hs], mel[start // self._hop_size:stop // self._hop_size + 1]


class VocoderCollate:
    def __init__(self, x_zero=0, mel_zero=-5):
        self._x_zero = x_zero
        self._mel_zero = mel_zero

    def collate_fn(self, examples):
        la = max(x[0].shape[0] for x in examples)
        ll = max(x[1].shape[0] for x in examples)
        lm = max(x[2].shape[0] for x in examples)
        mel = np.ones((len(examples), lm, examples[0][2].shape[1]), dtype=np.float64) * self._mel_zero
        x = np.ones((len(examples), la)) * self._x_zero
        x_low = np.ones((len(examples), ll)) * self._x_zero
        for ii, (cx, cxl, cmel) in enumerate(examples):
            mel[ii, :cmel.shape[0], :] = cmel
            x[ii, :cx.shape[0]] = cx
            x_low[ii, :cxl.shape[0]] = cxl
        return {'x': torch.tensor(x, dtype=torch.float), 'x_low': torch.tensor(x_low, dtype=torch.float),
                'mel': torch.tensor(mel, dtype=torch.float)}
