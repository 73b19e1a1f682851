"""Generate HiFi-GAN generator golden vectors from an INDEPENDENT implementation.

Runs in the build container only (needs `transformers`).  The reference's own
``hifigan/`` source is absent, so the surrogate oracle named in SURVEY.md
§8(c) — ``transformers.models.speecht5.modeling_speecht5.SpeechT5HifiGan`` — produces the
vectors.  Weights are seeded + variance-scaled so activations are O(1); saved under the
reference checkpoint key names (conv_pre / ups.N / resblocks.N.convs{1,2}.M / conv_post).

    python tools/gen_golden_hifigan.py     ->  tests/golden/hifigan_*.npz
"""
import os
import sys
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import hifigan_ref as R  # only for synthetic weights (shared seed recipe)


def build_surrogate(h, folded):
    from transformers import SpeechT5HifiGanConfig
    from transformers.models.speecht5.modeling_speecht5 import SpeechT5HifiGan
    cfg = SpeechT5HifiGanConfig(
        model_in_dim=h['num_mels'], upsample_rates=h['upsample_rates'],
        upsample_kernel_sizes=h['upsample_kernel_sizes'],
        upsample_initial_channel=h['upsample_initial_channel'],
        resblock_kernel_sizes=h['resblock_kernel_sizes'],
        resblock_dilation_sizes=h['resblock_dilation_sizes'],
        leaky_relu_slope=0.1, normalize_before=False)
    m = SpeechT5HifiGan(cfg).eval()
    sd = {}
    for k, v in folded.items():
        k2 = k.replace('ups.', 'upsampler.')
        sd[k2] = v
    sd['mean'] = torch.zeros(h['num_mels'])
    sd['scale'] = torch.ones(h['num_mels'])
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all(k in ('mean', 'scale') for k in missing), missing
    return m


def main():
    out_dir = os.path.join(ROOT, 'tests', 'golden')
    os.makedirs(out_dir, exist_ok=True)
    cases = [
        ('hifigan_c64_r5344', dict(R.CONFIG_V1, upsample_initial_channel=64), [1, 7, 50], 1234),
        # data/models/vocoder/neb-noft/config.json uses rates [3,5,4,4]
        ('hifigan_c32_r3544', dict(R.CONFIG_V1, upsample_initial_channel=32, upsample_rates=[3, 5, 4, 4]), [3, 20], 99),
    ]
    for name, h, Ts, seed in cases:
        sd = R.synthetic_state_dict(h, seed=seed, weight_norm=True)
        folded = R.fold_state_dict(sd)
        m = build_surrogate(h, folded)
        blob = {'cfg_json': np.array(__import__('json').dumps(h))}
        for k, v in sd.items():
            blob['sd/' + k] = v.numpy().astype(np.float32)
        for T in Ts:
            mel = R.synthetic_mel(2, T, h['num_mels'], seed=seed + T)
            with torch.no_grad():
                wav = m(mel.transpose(1, 2))  # surrogate takes [B,T,80] -> [B,L]
            blob['mel/%d' % T] = mel.numpy()
            blob['wav/%d' % T] = wav.numpy().astype(np.float32)
            print(name, 'T', T, '->', tuple(wav.shape), 'rms', float(wav.pow(2).mean().sqrt()))
        np.savez_compressed(os.path.join(out_dir, name + '.npz'), **blob)


if __name__ == '__main__':
    main()
