"""Oracle pinning (CPU): the HiFi-GAN restatement vs golden vectors from an independent
implementation (tools/gen_golden_hifigan.py; SURVEY.md §8c surrogate oracle)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import hifigan_ref as R


def _load(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name))
    h = json.loads(str(z['cfg_json']))
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith('sd/')}
    Ts = sorted(int(k[4:]) for k in z.files if k.startswith('mel/'))
    return z, h, sd, Ts


@pytest.mark.parametrize('name', ['hifigan_c64_r5344.npz', 'hifigan_c32_r3544.npz'])
def test_oracle_matches_independent_golden(golden_dir, name):
    z, h, sd, Ts = _load(golden_dir, name)
    w = R.fold_state_dict(sd)
    for T in Ts:
        mel = torch.from_numpy(z['mel/%d' % T])
        ref = torch.from_numpy(z['wav/%d' % T])
        out = R.generator_forward(w, h, mel).squeeze(1)
        assert out.shape == ref.shape
        assert out.shape[1] == R.out_len(h, T)
        rms = float((out - ref).pow(2).mean().sqrt())
        assert rms < 1e-6, (T, rms)


def test_out_len_reference_configs():
    # 240*T+64 for rates [5,3,4,4], 240*T+96 for neb-noft [3,5,4,4]
    h = dict(R.CONFIG_V1)
    assert R.out_len(h, 300) == 72064
    assert R.out_len(h, 800) == 192064
    h2 = dict(R.CONFIG_V1, upsample_rates=[3, 5, 4, 4])
    assert R.out_len(h2, 100) == 24096


def test_weight_norm_fold_matches_torch():
    torch.manual_seed(0)
    conv = torch.nn.utils.weight_norm(torch.nn.Conv1d(6, 4, 3))
    convt = torch.nn.utils.weight_norm(torch.nn.ConvTranspose1d(6, 4, 4, 2))
    with torch.no_grad():
        conv.weight_g.mul_(1.7)
        convt.weight_g.mul_(0.3)
    for m in (conv, convt):
        x = torch.randn(1, 6, 9)
        y = m(x)
        w = R.fold_weight_norm(m.weight_g.detach(), m.weight_v.detach())
        torch.nn.utils.remove_weight_norm(m)
        assert torch.allclose(m.weight, w, atol=1e-7)
        assert torch.allclose(m(x), y, atol=1e-6)
