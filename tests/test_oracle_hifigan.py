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


def test_oracle_full_v1_matches_surrogate_layer_by_layer(golden_dir):
    """The FULL config_v1 generator (512 channels) stage by stage against the independent implementation: conv_pre output, every
    upsampler's output, every stage's block mean and the waveform (tools/gen_golden_hifigan.py::full_v1).  The reference's own
    generator source is absent, so this is the strongest pin available for oracle/hifigan_ref.py."""
    import json
    import torch
    import torch.nn.functional as F
    z = np.load(os.path.join(golden_dir, 'hifigan_full_v1.npz'))
    h = json.loads(str(z['cfg_json']))
    assert h['upsample_initial_channel'] == 512
    w = R.fold_state_dict(R.synthetic_state_dict(h, seed=int(z['seed']), weight_norm=True))
    assert abs(float(sum(v.double().abs().sum() for v in w.values())) - float(z['weights_abs_sum'])) < 1e-6 * float(z['weights_abs_sum'])
    rel = lambda a, b: float((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt())
    for T in (7, 30):
        mel = torch.from_numpy(z['mel/%d' % T])
        wav, stages, ups = R.generator_forward(w, h, mel, return_stages='all')
        assert float((wav[:, 0] - torch.from_numpy(z['wav/%d' % T])).pow(2).mean().sqrt()) < 2e-6
        if T != 7:
            continue
        assert rel(stages[0], torch.from_numpy(z['stage/conv_pre'])) < 1e-6
        n = len(h['upsample_rates'])
        for i in range(n):
            assert rel(ups[i], torch.from_numpy(z['stage/ups_out.%d' % i])) < 2e-6, i
            slope = 0.1 if i < n - 1 else 0.01
            assert rel(F.leaky_relu(stages[i + 1], slope), torch.from_numpy(z['stage/stage_lrelu.%d' % i])) < 2e-6, i
