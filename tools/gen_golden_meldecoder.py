"""Golden vectors for the mel decoders, produced by the REFERENCE ITSELF (imported from /root/reference).

    python tools/gen_golden_meldecoder.py  ->  tests/golden/languasito2_*.npz, textcoder_*.npz

Weights: oracle.meldecoder_ref.fill_state_dict(named_shapes(reference module), seed) loaded with strict=True — the
fixture stores only the seed and the (name, shape) list, the tests rebuild identical tensors.  The only stochastic
layer, PreNet's always-on dropout (cube/networks/modules.py:163), is made reproducible by replacing `torch.dropout`
with a function that replays pre-drawn masks (torch itself is patched, the reference source is untouched)."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import ref_import  # noqa: E402

ref_import.setup()
from cube.networks.modules import Languasito2  # noqa: E402
from cube.networks.textcoder import CubenetTextcoder  # noqa: E402
from oracle import meldecoder_ref as M  # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden')


def gen_languasito(name, seed, nph, num_phones=50, num_speakers=3, max_pitch=300, max_duration=12):
    torch.manual_seed(0)
    net = Languasito2(num_phones, num_speakers, max_pitch, max_duration, cond_type=None)
    shapes = M.named_shapes(net)
    net.load_state_dict(M.fill_state_dict(shapes, seed), strict=True)
    net.eval()
    rng = np.random.RandomState(seed)
    x_char = torch.from_numpy(rng.randint(1, num_phones + 1, size=(1, nph))).long()
    x_speaker = torch.tensor([[2]]).long()
    X = {'x_char': x_char, 'x_speaker': x_speaker, 'y_frame2phone': [[0]]}
    with torch.no_grad():
        cond = net.inference(X)
    f2p = X['y_frame2phone'][0]
    durs = np.bincount(np.asarray(f2p, dtype=np.int64), minlength=nph) if len(f2p) else np.zeros(nph, dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, name + '.npz'), seed=seed, shapes=json.dumps(shapes), x_char=x_char.numpy(),
                        x_speaker=x_speaker.numpy(), cond=cond.numpy(), durs=durs, pitch=X['y_pitch'].numpy(),
                        cfg=json.dumps(dict(num_phones=num_phones, num_speakers=num_speakers, max_pitch=max_pitch,
                                            max_duration=max_duration)))
    print(name, 'frames', cond.shape[1], 'cond rms', float(cond.pow(2).mean().sqrt()), 'durs', durs[:10])


class Enc:
    def __init__(self, nph, nsp, max_pitch, max_duration):
        self.phon2int = {str(i): i for i in range(nph)}
        self.speaker2int = {str(i): i for i in range(nsp)}
        self.max_pitch = max_pitch
        self.max_duration = max_duration


def gen_textcoder(name, seed, nph, max_duration=9):
    torch.manual_seed(0)
    enc = Enc(40, 2, 200, max_duration)
    net = CubenetTextcoder(enc)
    shapes = M.named_shapes(net)
    net.load_state_dict(M.fill_state_dict(shapes, seed), strict=True)
    net.eval()
    rng = np.random.RandomState(seed)
    x_char = torch.from_numpy(rng.randint(1, 41, size=(1, nph))).long()
    x_speaker = torch.tensor([[1]]).long()
    masks = torch.from_numpy((rng.uniform(size=(400, 2, 1, 1, 256)) > 0.5).astype(np.float32))
    calls = [0]
    orig = torch.dropout

    def replay(x, p, train):
        assert p == 0.5 and train
        m = masks[calls[0] // 2, calls[0] % 2]
        calls[0] += 1
        return x * m * 2.0

    torch.dropout = replay
    try:
        with torch.no_grad():
            mel = net.inference({'x_char': x_char, 'x_speaker': x_speaker})
        steps = calls[0] // 2
        # teacher-forced forward on the same text with a synthetic alignment / target mel
        durs_tf = rng.randint(1, 7, size=nph)
        f2p = [p for p, d in enumerate(durs_tf) for _ in range(d)]
        F_ = len(f2p)
        y_mgc = torch.from_numpy(np.clip(rng.randn(1, F_, 80) - 2, -5, 1).astype(np.float32))
        n_tf = F_ // 3 + 1
        masks_tf = torch.from_numpy((rng.uniform(size=(2, 1, n_tf, 256)) > 0.5).astype(np.float32))
        calls[0] = 0

        def replay_tf(x, p, train):
            m = masks_tf[calls[0]]
            calls[0] += 1
            return x * m * 2.0

        torch.dropout = replay_tf
        with torch.no_grad():
            o_dur, o_pitch, o_mel, o_post = net.forward({'x_char': x_char, 'x_speaker': x_speaker, 'y_frame2phone': [f2p],
                                                         'y_mgc': y_mgc})
    finally:
        torch.dropout = orig
    np.savez_compressed(os.path.join(OUT, name + '.npz'), seed=seed, shapes=json.dumps(shapes), x_char=x_char.numpy(),
                        x_speaker=x_speaker.numpy(), masks=masks[:steps, :, 0].numpy(), mel=mel.numpy(),
                        f2p_tf=np.asarray(f2p), y_mgc=y_mgc.numpy(), masks_tf=masks_tf.numpy(), tf_dur=o_dur.numpy(),
                        tf_mel=o_mel.numpy(), tf_post=o_post.numpy(), max_duration=max_duration)
    print(name, 'AR steps', steps, 'mel', tuple(mel.shape), 'rms', float(mel.pow(2).mean().sqrt()), 'tf mel', tuple(o_mel.shape))


if __name__ == '__main__':
    os.makedirs(OUT, exist_ok=True)
    gen_languasito('languasito2_a', 31, 17)
    gen_languasito('languasito2_b', 32, 5)
    gen_textcoder('textcoder_a', 41, 11)
