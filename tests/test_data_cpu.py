"""CPU: the corpus readers behind --train-folder / --dev-folder (mirrors of cube/io_utils/io_cubegan.py:20-110 and
io_vocoder.py:20-112) on a two-utterance corpus written to a temp directory, and the trainers' refusal to fall back to synthetic
data silently."""
import importlib.util
import json
import os
from argparse import Namespace

import numpy as np
import pytest
import torch

from oracle import melspec_ref as M
from tests.conftest import ROOT


def _write_corpus(d, n=2, sr=24000):
    rng = np.random.RandomState(0)
    from ttscube_amd.io_utils.audio import save_wav
    ids = []
    for i in range(n):
        nph = 5 + i
        durs = rng.randint(3, 8, size=nph)
        f2p = [p for p, k in enumerate(durs) for _ in range(k)]
        F_ = len(f2p)
        uid = 'utt%02d' % i
        json.dump({'id': uid, 'phones': ['_'] + ['p%d' % v for v in rng.randint(0, 4, size=nph - 2)] + ['_'], 'frame2phon': f2p,
                   'speaker': 'spk%d' % (i % 2), 'phon2word': [0] * nph, 'left_context': 'a b', 'right_context': 'c', 'words': ['w']},
                  open(os.path.join(d, uid + '.json'), 'w'))
        np.save(open(os.path.join(d, uid + '.mgc'), 'wb'), np.clip(rng.randn(F_, 80) - 2, -5, 1))
        np.save(open(os.path.join(d, uid + '.pitch'), 'wb'), rng.randint(60, 300, size=F_).astype(np.float64))
        save_wav(os.path.join(d, uid + '.wav'), 0.5 * np.sin(np.arange(F_ * 240) * 0.05), sr)
        ids.append((uid, f2p))
    json.dump({'id': 'too_long', 'phones': ['a'], 'frame2phon': [0] * 401, 'speaker': 's', 'left_context': '', 'right_context': ''},
              open(os.path.join(d, 'too_long.json'), 'w'))
    np.save(open(os.path.join(d, 'too_long.mgc'), 'wb'), np.zeros((401, 80)))
    np.save(open(os.path.join(d, 'too_long.pitch'), 'wb'), np.zeros(401))
    return ids


def test_cubegan_dataset_collate_and_encodings(tmp_path):
    from ttscube_amd.io_utils.io_cubegan import CubeganCollate, CubeganDataset, CubeganEncodings
    ids = _write_corpus(str(tmp_path))
    ds = CubeganDataset(str(tmp_path))
    assert len(ds) == 2                                         # the utterance with a 401-frame phone is dropped (io_cubegan.py:41-45)
    ex = ds[0]
    f2p = ids[0][1]
    assert ex['meta']['id'] == 'utt00' and ex['mgc'].shape == (len(f2p), 80) and ex['audio'].dtype == np.float32
    assert abs(len(ex['audio']) - 240 * len(f2p)) <= 1
    first = [i for i, p in enumerate(f2p) if p == 0 or p == max(f2p)]
    inner = [i for i, p in enumerate(f2p) if 0 < p < max(f2p)]
    assert all(not ex['audio'][i * 240:(i + 1) * 240].any() and ex['pitch'][i] == 0 for i in first)   # _make_absolute_silence
    assert any(ex['audio'][i * 240:(i + 1) * 240].any() for i in inner)
    enc = CubeganEncodings()
    enc.compute([ds[i] for i in range(len(ds))])
    assert set(enc.speaker2int) == {'spk0', 'spk1'} and enc.max_duration <= 7 and enc.max_pitch < 300
    enc.save(str(tmp_path / 'e.json'))
    assert CubeganEncodings(str(tmp_path / 'e.json')).phon2int == enc.phon2int
    X = CubeganCollate(enc).collate_fn([ds[0], ds[1]])
    assert X['x_char'].shape == (2, 6) and X['x_len'].tolist() == [5, 6] and X['y_audio'].shape[1] == 240 * X['y_mgc'].shape[1]
    assert X['y_dur'][0, 5] == int(max(enc.max_pitch, enc.max_duration) + 1)       # padding = ignore_index


def test_audio_io_roundtrip_and_resampling(tmp_path):
    from ttscube_amd.io_utils.audio import load_wav, save_wav
    t = np.arange(48000) / 48000.0
    y = 0.5 * np.sin(2 * np.pi * 440 * t)
    save_wav(str(tmp_path / 'a.wav'), y, 48000)
    z, sr = load_wav(str(tmp_path / 'a.wav'), 24000)
    assert sr == 24000 and abs(len(z) - 24000) <= 1 and z.dtype == np.float32
    ref = 0.5 * np.sin(2 * np.pi * 440 * np.arange(len(z)) / 24000.0)
    assert float(np.abs(z[200:-200] - ref[200:-200]).max()) < 2e-3
    z2, _ = load_wav(str(tmp_path / 'a.wav'), 48000)
    assert float(np.abs(z2 - y).max()) < 1e-4             # int16 PCM quantisation


class _CpuMel:
    def melspectrogram(self, y, sample_rate, num_mels, hop_size, use_preemphasis=False):
        return M.melspectrogram_log10(y, sample_rate, num_mels, hop_size).astype(np.float32)


def test_vocoder_dataset_cache_crops_and_collate(tmp_path):
    from ttscube_amd.io_utils.audio import save_wav
    from ttscube_amd.io_utils.io_vocoder import VocoderCollate, VocoderDataset
    d = tmp_path / 'wavs'
    d.mkdir()
    for i, n in enumerate((30000, 40000)):
        save_wav(str(d / ('w%d.wav' % i)), 0.3 * np.sin(np.arange(n) * (0.03 + 0.01 * i)), 24000)
    (d / 'tiny.wav').write_bytes(b'RIFF')                        # below the 4096-byte threshold: skipped
    cache = str(tmp_path / 'cache')
    ds = VocoderDataset(str(d), max_segment_size=12000, random_start=True, cache_dir=cache, mel_vocoder=_CpuMel())
    assert len(ds) == 2
    wav, low, mel = ds[0]
    _ = ds[1]                                                    # (fills the cache for the second file too)
    assert wav.shape == (12000,) and low.shape == (1200,) and mel.shape == (51, 80)
    assert len(os.listdir(cache)) == 6 and abs(np.abs(np.load(os.path.join(cache, os.listdir(cache)[0]))).max()) > 0
    full = VocoderDataset(str(d), max_segment_size=-1, cache_dir=cache, mel_vocoder=None)   # served from the cache: no mel vocoder needed
    w_full, l_full, m_full = full[0]
    assert abs(np.abs(w_full).max() - 0.98) < 1e-6 and m_full.shape == (1 + len(w_full) // 240, 80) and len(l_full) == len(w_full) // 10
    head = VocoderDataset(str(d), max_segment_size=12000, random_start=False, cache_dir=cache)[1]
    assert np.array_equal(head[0], full[1][0][:12000])
    b = VocoderCollate().collate_fn([ds[0], full[1]])
    assert b['x'].shape == (2, 40000) and b['mel'].shape[2] == 80 and float(b['mel'][0, -1, 0]) == -5.0 and b['x'].dtype == torch.float32


def _load_script(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, 'scripts', name + '.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_trainers_do_not_fall_back_to_synthetic_data(tmp_path):
    tv = _load_script('train_vocoder')
    p = Namespace(synthetic=0, train_folder=str(tmp_path / 'nope'), dev_folder=str(tmp_path / 'nope'), sample_rate=24000, sample_rate_low=2400,
                  hop_size=240, maximum_segment_size=2400)
    with pytest.raises(SystemExit):
        tv._datasets(p, 0, 1)
    p.synthetic = 3
    train, dev = tv._datasets(p, 1, 2)
    assert len(train) == 3 and len(dev) == 2 and train[0][2].shape == (11, 80)


def test_rank_shards_have_equal_step_counts_and_loader_prefetches():
    """with N=97 items, world=2, bs=16 the old `items[rank::world]` slices gave 4 vs 3 batches -> one rank would block in
    an extra gradient exchange.  rank_shard wrap-pads to ceil(N / world) so every rank runs the same number of steps."""
    from ttscube_amd.io_utils.loader import BatchLoader, equal_batches, rank_shard
    for n, world, bs in ((97, 2, 16), (5, 8, 2), (64, 8, 16), (1, 4, 3)):
        shards = [rank_shard(n, r, world) for r in range(world)]
        assert len({len(s) for s in shards}) == 1 and len(shards[0]) == -(-n // world)
        assert len({len(equal_batches(s, bs)) for s in shards}) == 1
        assert set(i for s in shards for i in s) == set(range(n))          # nothing dropped
    assert rank_shard(0, 0, 2) == []

    class DS:
        def __getitem__(self, i):
            if i == 13:
                raise ValueError('bad item')
            return i * i

    col = lambda items: sum(items)
    batches = equal_batches(list(range(10)), 3)
    for nw in (0, 3):
        assert list(BatchLoader(DS(), batches, col, num_workers=nw)) == [sum(i * i for i in b) for b in batches]
    import pytest
    with pytest.raises(ValueError):
        list(BatchLoader(DS(), [[1, 2], [13]], col, num_workers=2))
    it = iter(BatchLoader(DS(), equal_batches(list(range(12)), 2), col, num_workers=2))   # abandoning the iterator must not hang
    assert next(it) == 1
    del it
