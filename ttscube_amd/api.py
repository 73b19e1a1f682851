"""Mirror of cube/api.py: ``TTSCube.load(model_name)`` / ``TTSCube(model_path, phonemizer_path)`` /
``tts(text, speaker) -> np.int16 @ 24 kHz``.

The text front-end (phonemizer network + tokenizers, cube/io_utils/io_text.py) is outside the hot path (SURVEY.md
§2.1): pass ``text2feat`` = any callable ``text -> {'phones': [...], 'words': [...], 'phon2word': [...]}`` (the
reference's Text2Feat* objects satisfy it); without one, the text is read as whitespace-separated phoneme symbols.
New on top of the reference (B=1 only): ``synthesize_batch`` runs many sentences per call, length-bucketed, and
``shard`` splits a sentence list across ranks (one process per GPU, no collectives)."""
import os
from pathlib import Path

import numpy as np
import torch
import yaml

from . import _lib
from .io_utils.io_cubegan import CubeganCollate, CubeganEncodings
from .networks.cubegan import Cubegan


class PhoneText2Feat:
    """Fallback front-end: 'text' is a whitespace-separated phoneme string; '|' separates words."""

    def __call__(self, text):
        words, phones, p2w = [], [], []
        for wi, w in enumerate(text.split('|')):
            words.append(w.strip())
            for p in w.split():
                phones.append(p)
                p2w.append(wi)
        return {'orig_text': text, 'words': words, 'phones': phones, 'phon2word': p2w}


class TTSCube:
    def __init__(self, model_path: str, phonemizer_path: str = None, text2feat=None, device='cuda:0'):
        encodings = CubeganEncodings('{0}.encodings'.format(model_path))
        conf = yaml.load(open('{0}.yaml'.format(model_path)), yaml.Loader)
        cond_type = conf.get('conditioning')
        self._model = Cubegan(encodings, conditioning=cond_type, train=False)
        self._model.load('{0}.model'.format(model_path))
        self._collate = CubeganCollate(encodings, conditioning_type=cond_type)
        self._text2feat = text2feat if text2feat is not None else PhoneText2Feat()
        self._model.eval()
        self._model.to(device)

    @staticmethod
    def load(model_name: str, **kw):
        base_name = '{0}/.ttscube/models/{1}'.format(str(Path.home()), model_name)
        if not os.path.exists('{0}/cubegan.model'.format(base_name)):
            raise FileNotFoundError('%s/cubegan.{model,yaml,encodings} not found and this build has no network access to '
                                    'download it (cube/io_utils/repository.py:27-61); unpack the exported model there' % base_name)
        return TTSCube('{0}/cubegan'.format(base_name), '{0}/phonemizer'.format(base_name), **kw)

    def _example(self, text, speaker):
        """The dummy-target example cube/api.py:47-57 builds around the front-end output."""
        rez = {'meta': dict(self._text2feat(text))}
        rez['meta']['speaker'] = speaker
        rez['pitch'] = np.zeros((100))
        rez['mgc'] = np.zeros((100, 80))
        rez['meta']['words_left'] = []
        rez['meta']['words_right'] = []
        rez['meta']['frame2phon'] = [0] * 100
        return rez

    def __call__(self, text, speaker='none'):
        """cube/api.py:45-66: one sentence -> int16 audio at 24 kHz."""
        with torch.no_grad():
            X = self._collate.collate_fn([self._example(text, speaker)])
            for key in X:
                if isinstance(X[key], torch.Tensor):
                    X[key] = X[key].to(self._model.get_device())
            # the generator's range guard runs deferred: nothing waits for the GPU until the audio is copied back — where a synchronisation happens
            # anyway — and the guard's verdict is collected right behind that copy; a tripped guard reruns the sentence with the self-repairing
            # synchronous check (re-calibration on this input)
            gen = self._model._generator
            tripped = False
            try:
                audio = self._model.inference(X, check='deferred')
                host = audio.detach().cpu()
            finally:
                # whatever happens between the deferred forward and here (an exception in the copy, a KeyboardInterrupt), the pending verdict is
                # collected: a stale one would be blamed on the next, unrelated call
                if gen.range_check_pending():
                    tripped = gen.finish_range_check(raise_on_trip=False)
            if tripped:
                host = self._model.inference(X, check='sync').detach().cpu()
            return np.asarray(host.numpy().squeeze() * 32767, dtype=np.int16)

    def synthesize_batch(self, texts, speaker='none', max_batch=64):
        """Many sentences -> list of int16 arrays (same order).  Sentences are sorted by phoneme count and run in
        padded batches of up to `max_batch`; ragged lengths are handled inside the kernels (masked BiLSTMs), so each
        result equals the single-sentence call."""
        speakers = speaker if isinstance(speaker, (list, tuple)) else [speaker] * len(texts)
        ex = [self._example(t, s) for t, s in zip(texts, speakers)]
        order = sorted(range(len(ex)), key=lambda i: len(ex[i]['meta']['phones']))
        out = [None] * len(ex)
        groups = [order[s:s + max_batch] for s in range(0, len(order), max_batch)]

        def feed():
            for ids in groups:
                X = self._collate.collate_fn([ex[i] for i in ids])
                for key in X:
                    if isinstance(X[key], torch.Tensor):
                        X[key] = X[key].to(self._model.get_device())
                yield X

        def collect(results):
            for ids, (wav, lens) in zip(groups, results):
                wav = wav.detach().cpu().numpy()
                for j, i in enumerate(ids):
                    out[i] = np.asarray(wav[j, 0, :lens[j]] * 32767, dtype=np.int16)

        with torch.no_grad():
            if len(groups) > 1:
                # several padded batches: the text / frame stacks of batch k + 1 run under the generator of batch k (Cubegan.inference_pipelined);
                # per-batch results are the ones `inference` gives.  The pipeline runs the range guard deferred; a batch that left the calibrated
                # range raises there, and the whole list is then redone batch by batch with the self-repairing synchronous guard.
                try:
                    collect(self._model.inference_pipelined(feed()))
                    return out
                except _lib.TTSCError as e:
                    if 'check="sync"' not in str(e):
                        raise
            collect(self._model.inference(X, return_lengths=True) for X in feed())
        return out

    @staticmethod
    def shard(items, rank, world_size):
        """Contiguous utterance shard for one rank (inference is embarrassingly parallel: no collectives)."""
        n = len(items)
        lo, hi = rank * n // world_size, (rank + 1) * n // world_size
        return items[lo:hi]
