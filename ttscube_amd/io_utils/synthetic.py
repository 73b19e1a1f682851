"""Seeded synthetic training examples in the reference's per-example dict layout (cube/io_utils/io_cubegan.py:20-110: 'meta'
{phones, speaker, frame2phon, phon2word}, 'mgc' [F,80], 'pitch' [F], 'audio' [240 F]) — what the trainers and benches use when
no corpus is on disk (there is none in this environment: no network)."""
import numpy as np


def synthetic_examples(n, seed, nphones=40, min_ph=20, max_ph=60, speakers=2):
    rng = np.random.RandomState(seed)
    for _ in range(n):
        nph = int(rng.randint(min_ph, max_ph))
        durs = rng.randint(2, 12, size=nph)
        f2p = [p for p, d in enumerate(durs) for _ in range(d)]
        F_ = len(f2p)
        yield {'meta': {'phones': ['p%d' % v for v in rng.randint(0, nphones, size=nph)], 'speaker': 's%d' % rng.randint(0, speakers),
                        'frame2phon': f2p, 'phon2word': [0] * nph},
               'mgc': np.clip(rng.randn(F_, 80) - 2, -5, 1), 'pitch': rng.randint(60, 300, size=F_).astype(np.float64),
               'audio': (0.3 * np.sin(np.cumsum(rng.uniform(0.01, 0.3, size=F_ * 240)))).astype(np.float32)}


def synthetic_encodings(nphones=40, speakers=2):
    """encodings covering everything synthetic_examples can emit (identical on every rank)"""
    from .io_cubegan import CubeganEncodings
    enc = CubeganEncodings()
    enc.phon2int = {'p%d' % i: i for i in range(nphones)}
    enc.speaker2int = {'s%d' % i: i for i in range(speakers)}
    enc.max_pitch, enc.max_duration = 300, 12
    return enc


def synthetic_sentences(n, seed=1234, nphones=50, min_ph=20, max_ph=120):
    """BASELINE configs[4] text side: n random phoneme-id sentences, ids U{1..nphones} (0 = padding, the collate's
    `id + 1` convention of io_cubegan.py:219-231), lengths U{min_ph..max_ph}.  Returns (x_char [n, max_len] int64 zero-padded, lens)."""
    rs = np.random.RandomState(seed)
    lens = rs.randint(min_ph, max_ph + 1, size=n)
    xc = np.zeros((n, int(lens.max())), dtype=np.int64)
    for b, l in enumerate(lens):
        xc[b, :l] = rs.randint(1, nphones + 1, size=l)
    return xc, lens
