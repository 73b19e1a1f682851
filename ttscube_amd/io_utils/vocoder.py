"""Mirror of cube/io_utils/vocoder.py::MelVocoder for the part the vocoder pipeline uses (`melspectrogram`, vocoder.py:54-63):
STFT 1024 / hop_size, Hann, centred frames -> 80-bin mel (Slaney) -> log10(max(1e-5, .)).  The reference computes it with
librosa on the CPU inside DataLoader workers; here it runs on the GPU (io_utils/melspec.py: DFT and mel projection as MFMA
GEMMs).  Pre-emphasis / Griffin-Lim / ifft of the reference class are not on the path and not provided."""
import numpy as np
import torch

from . import melspec


class MelVocoder:
    def __init__(self, device='cuda:0'):
        self._device = torch.device(device)

    def melspectrogram(self, y, sample_rate, num_mels, hop_size, use_preemphasis=False):
        """y: 1-D numpy array / tensor (or [B, L]) -> numpy [frames, num_mels] (or [B, frames, num_mels]), float32."""
        if use_preemphasis:
            raise NotImplementedError('use_preemphasis=True is never used by the reference pipeline (io_vocoder.py:55-59)')
        t = torch.as_tensor(np.asarray(y) if not torch.is_tensor(y) else y, dtype=torch.float32)
        single = t.dim() == 1
        if single:
            t = t.unsqueeze(0)
        m = melspec.melspectrogram_log10(t.to(self._device), sample_rate=sample_rate, num_mels=num_mels, hop_size=hop_size)
        m = m.cpu().numpy()
        return m[0] if single else m
