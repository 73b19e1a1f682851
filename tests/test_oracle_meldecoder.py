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
