"""Mirror of cube/io_utils/io_cubegan.py: ``CubeganDataset`` (:20-110, the processed-corpus reader: <id>.json / .mgc / .pitch /
.wav), ``CubeganEncodings`` (:112-160, same JSON file format) and ``CubeganCollate.collate_fn`` (:162-231, same batch-dict keys,
dtypes and padding values — SURVEY.md §8b).  External text conditioning (fastText / HF encoders) needs downloads and is not
available here — pass pre-computed ``x_words`` instead."""
import json
import os

import numpy as np
import torch

from .audio import load_wav


class CubeganDataset:
    """cube/io_utils/io_cubegan.py:20-110.  `base_path` holds, per utterance id, `<id>.json` (phones, frame2phon, speaker,
    left/right context ...), `<id>.mgc` and `<id>.pitch` (bare np.save streams) and `<id>.wav`.  Utterances with a phone longer
    than 400 frames are dropped (:41-45); `__getitem__` silences the frames aligned to the first / last phone (:80-90)."""

    def __init__(self, base_path, hf_model=None):
        if hf_model is not None:
            raise NotImplementedError('hf:<model> conditioning needs a downloaded tokenizer; this build supports conditioning=None')
        self._base_path = base_path
        self._examples = []
        for f in sorted(os.listdir(base_path)):
            if not f.endswith('.mgc'):
                continue
            bpath = os.path.join(base_path, f[:-4])
            if not (os.path.exists(bpath + '.json') and os.path.exists(bpath + '.pitch')):
                continue
            example = json.load(open(bpath + '.json'))
            durs = np.zeros((len(example['phones'])))
            for index in example['frame2phon']:
                durs[index] += 1
            if durs.size == 0 or max(durs) > 400:
                continue
            example.setdefault('id', f[:-4])
            example['words_left'] = str(example.get('left_context', '')).split()     # (the reference's SimpleTokenizer is a front-end
            example['words_right'] = str(example.get('right_context', '')).split()   #  component; unused with conditioning=None)
            self._examples.append(example)

    def __len__(self):
        return len(self._examples)

    def meta_items(self):
        """{'meta', 'pitch'} of every item WITHOUT decoding its wav / mgc: all CubeganEncodings.compute needs (io_cubegan.py:120-137)."""
        for description in self._examples:
            base_fn = os.path.join(self._base_path, description['id'])
            yield {'meta': description, 'pitch': np.array(np.load(open(base_fn + '.pitch', 'rb')), dtype=np.float64)}

    @staticmethod
    def _make_absolute_silence(audio, pitch, meta):
        max_phone = max(meta['frame2phon'])
        for i_frame, ph in enumerate(meta['frame2phon']):
            if ph == 0 or ph == max_phone:
                audio[i_frame * 240:i_frame * 240 + 240] = 0
                if i_frame < len(pitch):
                    pitch[i_frame] = 0
        return audio, pitch

    def __getitem__(self, item):
        description = self._examples[item]
        base_fn = os.path.join(self._base_path, description['id'])
        mgc = np.load(open(base_fn + '.mgc', 'rb'))
        pitch = np.array(np.load(open(base_fn + '.pitch', 'rb')), dtype=np.float64)
        audio, _ = load_wav(base_fn + '.wav', 24000)
        audio, pitch = self._make_absolute_silence(np.array(audio), pitch, description)
        return {'meta': description, 'mgc': mgc, 'pitch': pitch, 'audio': audio}


class CubeganEncodings:
    """Symbol tables and value ranges of a training corpus — the `<base>.encodings` JSON next to every Cubegan checkpoint.  Schema
    (fixed by the reference's files, cube/io_utils/io_cubegan.py:113-155): {"speaker2int", "phon2int", "max_duration", "max_pitch"};
    ids are assigned in order of first appearance, durations are frames per phoneme counted from `frame2phon`."""
    _FIELDS = ('speaker2int', 'phon2int', 'max_duration', 'max_pitch')

    def __init__(self, filename: str = None):
        self.speaker2int, self.phon2int = {}, {}
        self.max_duration = self.max_pitch = 0
        if filename is not None:
            self.load(filename)

    @staticmethod
    def _intern(table, symbol):
        return table.setdefault(symbol, len(table))

    def compute(self, dataset):
        """one pass over the corpus: extend both tables, track the largest pitch value and the longest phoneme (in frames)"""
        for ex in dataset:
            meta = ex['meta']
            self._intern(self.speaker2int, meta['speaker'])
            for ph in meta['phones']:
                self._intern(self.phon2int, ph)
            if len(ex['pitch']):
                self.max_pitch = max(self.max_pitch, np.max(ex['pitch']))
            frames_per_phone = np.bincount(np.asarray(meta['frame2phon'], dtype=np.int64), minlength=len(meta['phones']))
            if frames_per_phone.size:
                self.max_duration = max(self.max_duration, int(frames_per_phone.max()))

    def load(self, filename: str):
        with open(filename) as f:
            blob = json.load(f)
        for name in self._FIELDS:
            setattr(self, name, blob[name])

    def save(self, filename: str):
        blob = {'speaker2int': self.speaker2int, 'phon2int': self.phon2int,
                'max_duration': int(self.max_duration), 'max_pitch': int(self.max_pitch)}
        with open(filename, 'w') as f:
            json.dump(blob, f)


class CubeganCollate:
    def __init__(self, encodings: CubeganEncodings, conditioning_type=None, training=True):
        self._encodings = encodings
        self._ignore_index = int(max(encodings.max_pitch, encodings.max_duration) + 1)
        self._training = training
        if conditioning_type not in (None, 'none'):
            raise NotImplementedError("conditioning_type=%r needs a downloaded fastText / HuggingFace model; this build "
                                      "supports conditioning=None (SURVEY.md §2.1)" % (conditioning_type,))
        self._conditioning_type = None

    def collate_fn(self, batch):
        """io_cubegan.py:169-231."""
        max_char = max(len(e['meta']['phones']) for e in batch)
        max_mel = max(e['mgc'].shape[0] for e in batch)
        B = len(batch)
        x_char = np.zeros((B, max_char))
        x_p2w = np.zeros((B, max_char), dtype=np.int64)
        y_mgc = np.ones((B, max_mel, 80)) * -5
        x_speaker = np.zeros((B, 1))
        y_dur = np.zeros((B, max_char))
        y_pitch = np.zeros((B, max_mel))
        y_frame2phone = []
        y_audio = np.zeros((B, max_mel * 240), dtype=np.float64)
        for ii, example in enumerate(batch):
            y_mgc[ii, :example['mgc'].shape[0], :] = example['mgc']
            x_speaker[ii] = self._encodings.speaker2int[example['meta']['speaker']] + 1
            for jj, phoneme in enumerate(example['meta']['phones']):
                if phoneme in self._encodings.phon2int:
                    x_char[ii, jj] = self._encodings.phon2int[phoneme] + 1
            y_frame2phone.append(example['meta']['frame2phon'])
            p2w = example['meta'].get('phon2word', [])
            x_p2w[ii, :len(p2w)] = np.array(p2w, dtype=np.int64)
            for phone_idx in y_frame2phone[-1]:
                y_dur[ii, phone_idx] += 1
            y_dur[ii, len(example['meta']['phones']):] = self._ignore_index
            y_pitch[ii, :example['pitch'].shape[0]] = example['pitch']
            if 'audio' in example:
                m = min(y_audio.shape[1], example['audio'].shape[0])
                y_audio[ii, :m] = example['audio'][:m]
        for ii, example in enumerate(batch):
            m = len(example['meta']['phones'])
            y_dur[ii, :m] = np.clip(y_dur[ii, :m], 0, 100)
        # x_len is an addition to the reference's dict: the true phone count per example.  An out-of-vocabulary phone is encoded
        # as 0 — the padding id — so lengths cannot be recovered from x_char alone (batched inference masks by length).
        x_len = [len(e['meta']['phones']) for e in batch]
        return {'x_char': torch.tensor(x_char, dtype=torch.long), 'x_len': torch.tensor(x_len, dtype=torch.long), 'x_words': None, 'x_tok_ids': None, 'x_word2tok': None,
                'x_phon2word': torch.tensor(x_p2w), 'x_speaker': torch.tensor(x_speaker, dtype=torch.long),
                'y_mgc': torch.tensor(y_mgc, dtype=torch.float), 'y_frame2phone': y_frame2phone,
                'y_pitch': torch.tensor(y_pitch, dtype=torch.long), 'y_dur': torch.tensor(y_dur, dtype=torch.long),
                'y_audio': torch.tensor(y_audio, dtype=torch.float)}
