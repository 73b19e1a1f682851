"""Oracle pinning (CPU): the mel-decoder restatement vs vectors produced by the reference itself
(tools/gen_golden_meldecoder.py imports cube.networks.modules.Languasito2 / cube.networks.textcoder.CubenetTextcoder)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import meldecoder_ref as M


def _load(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name + '.npz'))
    shapes = [(k, tuple(s)) for k, s in json.loads(str(z['shapes']))]
    return z, M.fill_state_dict(shapes, int(z['seed']))


@pytest.mark.parametrize('name', ['languasito2_a', 'languasito2_b'])
def test_languasito2_inference_matches_reference(golden_dir, name):
    z, sd = _load(golden_dir, name)
    cfg = json.loads(str(z['cfg']))
    with torch.no_grad():
        cond, durs, pitch = M.languasito2_inference(sd, torch.from_numpy(z['x_char']), torch.from_numpy(z['x_speaker']), cfg['max_pitch'])
    assert list(durs) == list(z['durs'])                       # identical durations (SURVEY §8d parity gate a5)
    assert cond.shape == z['cond'].shape
    assert float((cond - torch.from_numpy(z['cond'])).pow(2).mean().sqrt()) < 1e-5
    assert float((pitch - torch.from_numpy(z['pitch'])).abs().max()) < 1e-3


def test_textcoder_inference_and_forward_match_reference(golden_dir):
    z, sd = _load(golden_dir, 'textcoder_a')
    x_char, x_spk = torch.from_numpy(z['x_char']), torch.from_numpy(z['x_speaker'])
    with torch.no_grad():
        mel, _ = M.textcoder_inference(sd, x_char, x_spk, torch.from_numpy(z['masks']).unsqueeze(2))
        assert mel.shape == z['mel'].shape
        assert float((mel - torch.from_numpy(z['mel'])).pow(2).mean().sqrt()) < 1e-5
        o_dur, o_mel, o_post = M.textcoder_forward(sd, x_char, x_spk, [list(z['f2p_tf'])], torch.from_numpy(z['y_mgc']),
                                                   torch.from_numpy(z['masks_tf']))
    assert float((o_dur - torch.from_numpy(z['tf_dur'])).abs().max()) < 1e-4
    assert float((o_mel - torch.from_numpy(z['tf_mel'])).pow(2).mean().sqrt()) < 1e-5
    assert float((o_post - torch.from_numpy(z['tf_post'])).pow(2).mean().sqrt()) < 1e-5


def test_lstm_restatement_matches_torch_lstm():
    torch.manual_seed(0)
    m = torch.nn.LSTM(input_size=20, hidden_size=16, num_layers=2, bidirectional=True, batch_first=True)
    x = torch.randn(3, 11, 20)
    with torch.no_grad():
        ref, (h, c) = m(x)
        y, (h2, c2) = M.lstm(x, {'r.' + k: v for k, v in m.state_dict().items()}, 'r', 2, True)
    assert float((y - ref).abs().max()) < 1e-6 and float((h - h2).abs().max()) < 1e-6 and float((c - c2).abs().max()) < 1e-6


# ---- round 6: teacher-forced forward, text losses + parameter gradients, external conditioning, a long sentence --------------------------
def _f2ps(z):
    out, o = [], 0
    for n in z['f2p_len']:
        out.append([int(v) for v in z['f2p_flat'][o:o + int(n)]])
        o += int(n)
    return out


@pytest.mark.parametrize('name', ['languasito2_train_a', 'languasito2_train_b'])
def test_languasito2_forward_losses_and_gradients_match_reference(golden_dir, name):
    """Languasito2.forward (modules.py:996-999) + cubegan.py:94-112 + autograd through the REFERENCE module (tools/gen_golden_training.py)
    vs autograd through the oracle's restatement: outputs, both losses, and the gradient fingerprint of every parameter."""
    from oracle.fingerprint import compare
    z, sd = _load(golden_dir, name)
    cfg = json.loads(str(z['cfg']))
    names = json.loads(str(z['grad_names']))
    for k in names:
        sd[k].requires_grad_(True)
    y_pitch = torch.from_numpy(z['y_pitch'])
    p_dur, p_pitch, p_vuv, cond = M.languasito2_forward(sd, torch.from_numpy(z['x_char']), torch.from_numpy(z['x_speaker']), _f2ps(z),
                                                        y_pitch, cfg['max_pitch'])
    for got, key in ((p_dur, 'p_dur'), (p_pitch, 'p_pitch'), (p_vuv, 'p_vuv'), (cond, 'conditioning')):
        assert got.shape == z[key].shape, key
        assert float((got.detach() - torch.from_numpy(z[key])).abs().max()) < 2e-5, key
    l_dur, l_pitch = M.text_losses(p_dur, p_pitch, p_vuv, torch.from_numpy(z['y_dur']), y_pitch, cfg['max_pitch'], cfg['max_duration'])
    assert abs(float(l_dur) - float(z['loss_duration'])) < 1e-5 and abs(float(l_pitch) - float(z['loss_pitch'])) < 1e-5
    l_cond = (cond * torch.from_numpy(z['cond_probe'])).sum() / cond.numel()
    (l_dur + l_pitch + l_cond).backward()
    worst = {}
    for k in names:
        fp = {f: z['grad/%s/%s' % (k, f)] for f in ('norm', 'sum', 'probe', 'idx', 'samples', 'size')}
        dev = compare(sd[k].grad.numpy(), k, fp)
        worst[k] = max(dev.values())
    bad = {k: v for k, v in worst.items() if v > 1e-4}
    assert not bad, bad


def test_languasito2_external_conditioning_matches_reference(golden_dir):
    """cond_type='fasttext': the `_lm_t/_lm_g` BiLSTMs over x_words and `_get_cond_selection` (modules.py:932-940, 1079-1082)."""
    z, sd = _load(golden_dir, 'languasito2_ft_a')
    cfg = json.loads(str(z['cfg']))
    with torch.no_grad():
        cond, durs, pitch = M.languasito2_inference(sd, torch.from_numpy(z['x_char']), torch.from_numpy(z['x_speaker']), cfg['max_pitch'],
                                                    x_words=torch.from_numpy(z['x_words']), x_phon2word=torch.from_numpy(z['x_phon2word']))
    assert list(durs) == list(z['durs'])
    assert cond.shape == z['cond'].shape and float((cond - torch.from_numpy(z['cond'])).pow(2).mean().sqrt()) < 1e-5
    assert float((pitch - torch.from_numpy(z['pitch'])).abs().max()) < 1e-3


def test_languasito2_long_sentence_matches_reference(golden_dir):
    """B = 1, 64 phonemes -> 576 frames (the two earlier goldens are 17 and 5 phonemes)."""
    z, sd = _load(golden_dir, 'languasito2_long')
    cfg = json.loads(str(z['cfg']))
    with torch.no_grad():
        cond, durs, pitch = M.languasito2_inference(sd, torch.from_numpy(z['x_char']), torch.from_numpy(z['x_speaker']), cfg['max_pitch'])
    assert list(durs) == list(z['durs']) and cond.shape[1] == 576
    assert float((cond - torch.from_numpy(z['cond'])).pow(2).mean().sqrt()) < 1e-5
