"""Mirror of the synthesis glue in cube/io_utils/runtime.py:
  cubegan_synthesize_dataset (:83-109)  Cubegan over a processed dev set -> <out>/<id>.wav (int16 @ 24 kHz), free or forced alignment
  synthesize_devset          (:41-80)   two-stage path: CubenetTextcoder -> log10-mel -> ln-mel -> HiFi-GAN Generator checkpoint
                                        (`ckpt['generator']`, config.json next to it, remove_weight_norm)
The PNG spectrogram rendering of the reference (PIL) is a debugging aid and is not reproduced."""
import json
import os

import numpy as np
import torch

from .audio import save_wav
from .io_cubegan import CubeganCollate, CubeganDataset


def cubegan_synthesize_dataset(model, output_path, devset_path, limit=-1, free=True, conditioning=None, speaker=None):
    collate = CubeganCollate(model._encodings, conditioning_type=conditioning)
    dataset = CubeganDataset(devset_path)
    m_gen = len(dataset) if limit == -1 or limit >= len(dataset) else limit
    os.makedirs(output_path, exist_ok=True)
    with torch.no_grad():
        for ii in range(m_gen):
            ex = dataset[ii]
            if speaker is not None:
                ex['meta']['speaker'] = speaker
            X = collate.collate_fn([ex])
            for key in X:
                if isinstance(X[key], torch.Tensor):
                    X[key] = X[key].to(model.get_device())
            audio = model.inference(X) if free else model(X)
            audio = audio.detach().cpu().numpy().squeeze()
            save_wav('{0}/{1}.wav'.format(output_path, ex['meta']['id']), np.asarray(audio * 32767, dtype=np.int16), 24000)
    return m_gen


def load_generator_checkpoint(vocoder_path):
    """runtime.py:44-54: config.json beside the checkpoint, state under 'generator', weight norm removed, eval mode"""
    from ..hifigan.env import AttrDict
    from ..hifigan.models import Generator
    h = AttrDict(json.loads(open(os.path.join(os.path.split(vocoder_path)[0], 'config.json')).read()))
    vocoder = Generator(h)
    vocoder.load_state_dict(torch.load(vocoder_path, map_location='cpu')['generator'])
    vocoder.remove_weight_norm()
    return vocoder.eval()


def synthesize_devset(textcoder, collate, dataset, vocoder, output_path='generated_files/', forced_synthesis=True, limit=-1, dropout_masks=None):
    """runtime.py:41-80 with the loaded objects passed in (textcoder: CubenetTextcoder on a HIP device; collate / dataset: whatever
    produces its batch dicts — the reference's TextcoderDataset / TextcoderCollate are corpus tooling; vocoder: from
    load_generator_checkpoint, on the same device).  dropout_masks (optional, one tensor per item): PreNet dropout masks to inject instead of
    the in-kernel Philox draw — the PreNet's dropout is always on (modules.py:159-164), so reproducible synthesis (tests, A/B listening)
    needs them."""
    os.makedirs(output_path, exist_ok=True)
    m_gen = len(dataset) if limit == -1 or limit >= len(dataset) else limit
    with torch.no_grad():
        for ii in range(m_gen):
            ex = dataset[ii]
            X = collate.collate_fn([ex])
            mk = dropout_masks[ii] if dropout_masks is not None else None
            mel = textcoder(X, dropout_masks=mk)[3] if forced_synthesis else textcoder.inference(X, dropout_masks=mk)
            mel = torch.log(10 ** mel)                       # log10-mel -> natural-log mel (runtime.py:77)
            audio = vocoder(mel.permute(0, 2, 1).contiguous()).detach().cpu().numpy().squeeze()
            save_wav('{0}/{1}.wav'.format(output_path, ex['meta']['id']), np.array(audio * 32767, dtype=np.int16), 24000)
    return m_gen
