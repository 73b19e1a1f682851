"""Golden vectors for the TRAINING-side functions of the hot path and for the external-conditioning branch, produced by the
REFERENCE ITSELF (imported from /root/reference; build container only — the reference never travels).

    python tools/gen_golden_training.py   ->  tests/golden/languasito2_train_*.npz, languasito2_ft_*.npz, languasito2_long.npz,
                                               vocoder_step_*.npz

What is pinned (VERDICT r5 "What's missing" #1, #2):
  * `Languasito2.forward` teacher-forced (cube/networks/modules.py:996-999) + the text losses of `Cubegan.training_step`
    (cube/networks/cubegan.py:94-112, restated call for call on the reference module's outputs — `Cubegan` itself cannot be imported:
    `hifigan/` is empty) + the gradient of (loss_duration + loss_pitch) and of a fixed linear functional of the conditioning with respect to
    EVERY parameter, by torch autograd through the reference module.
  * `Languasito2(cond_type='fasttext').inference` on random `x_words [1, Nw, 300]` / `x_phon2word` (modules.py:932-940, 1079-1082).
  * a B = 1, >= 60-phoneme `Languasito2.inference`.
  * `CubenetVocoder.training_step` (cube/networks/vocoder.py:136-156) for TWO consecutive steps from fixed weights: losses, gradient norms before
    clipping, every parameter after each step, the learning rate.

Gradients of the 13.5 M-parameter text model would be a 54 MB fixture; every tensor is therefore stored as a FINGERPRINT: L2 norm, sum, the dot
product with a seeded N(0,1) vector (`probe_vector`), and 128 strided samples.  A wrong gradient anywhere in the tensor moves the norm and the
probe; the samples localise it."""
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
from cube.networks.vocoder import CubenetVocoder  # noqa: E402
from oracle import meldecoder_ref as M  # noqa: E402
from oracle import wavernn_ref as O  # noqa: E402
from oracle.fingerprint import fingerprint  # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden')


def _lang(seed, num_phones, num_speakers, max_pitch, max_duration, cond_type=None):
    torch.manual_seed(0)
    net = Languasito2(num_phones, num_speakers, max_pitch, max_duration, cond_type=cond_type)
    shapes = M.named_shapes(net)
    net.load_state_dict(M.fill_state_dict(shapes, seed), strict=True)
    return net, shapes


def _batch(rng, B, nphs, num_phones, num_speakers, max_pitch, max_duration, dur_hi):
    """A CubeganCollate-shaped batch (cube/io_utils/io_cubegan.py:219-231): x_char 0-padded, y_dur padded with the ignore index,
    y_frame2phone per utterance, y_pitch [B, F] 0-padded."""
    N = max(nphs)
    ignore = int(max(max_pitch, max_duration) + 1)
    x_char = np.zeros((B, N), dtype=np.int64)
    y_dur = np.full((B, N), ignore, dtype=np.int64)
    f2ps = []
    for b, n in enumerate(nphs):
        x_char[b, :n] = rng.randint(1, num_phones + 1, size=n)
        d = rng.randint(1, dur_hi, size=n)
        y_dur[b, :n] = np.minimum(d, max_duration)
        f2ps.append([p for p, k in enumerate(d) for _ in range(k)])
    F_ = max(len(f) for f in f2ps)
    y_pitch = np.zeros((B, F_), dtype=np.int64)
    for b, f in enumerate(f2ps):
        voiced = rng.uniform(size=len(f)) > 0.3
        y_pitch[b, :len(f)] = np.where(voiced, rng.randint(60, max_pitch, size=len(f)), 0)
    x_speaker = rng.randint(1, num_speakers + 1, size=(B, 1)).astype(np.int64)
    return x_char, x_speaker, y_dur, f2ps, y_pitch


def gen_languasito_train(name, seed, nphs, num_phones=50, num_speakers=3, max_pitch=300, max_duration=12):
    net, shapes = _lang(seed, num_phones, num_speakers, max_pitch, max_duration)
    net.train()
    rng = np.random.RandomState(seed)
    B = len(nphs)
    x_char, x_speaker, y_dur, f2ps, y_pitch = _batch(rng, B, nphs, num_phones, num_speakers, max_pitch, max_duration, 7)
    X = {'x_char': torch.from_numpy(x_char), 'x_speaker': torch.from_numpy(x_speaker), 'y_frame2phone': [list(f) for f in f2ps],
         'y_pitch': torch.from_numpy(y_pitch), 'y_dur': torch.from_numpy(y_dur)}
    p_dur, p_pitch, p_vuv, conditioning = net(X)
    # ---- cubegan.py:94-112, on the reference module's outputs
    t_dur = X['y_dur']
    t_pitch = X['y_pitch']
    t_vuv = (t_pitch > 1).float()
    m_size = min(t_dur.shape[1], p_dur.shape[1])
    t_dur = t_dur[:, :m_size]
    p_dur_ = p_dur[:, :m_size, :]
    m_size = min(t_pitch.shape[1], p_pitch.shape[1])
    t_pitch = t_pitch[:, :m_size]
    p_pitch_ = p_pitch[:, :m_size]
    t_vuv = t_vuv[:, :m_size]
    p_vuv_ = p_vuv[:, :m_size]
    loss_duration = net._loss_cross(p_dur_.reshape(-1, p_dur_.shape[2]), t_dur.reshape(-1))
    loss_pitch = (torch.abs(t_pitch / net._max_pitch - p_pitch_) * t_vuv).mean() + torch.abs(t_vuv - p_vuv_).mean()
    # ---- a fixed linear functional of the conditioning stands in for the generator losses that reach the `_g` stack (cubegan.py:131-170)
    R = torch.from_numpy(rng.randn(*conditioning.shape).astype(np.float32))
    loss_cond = (conditioning * R).sum() / conditioning.numel()
    (loss_duration + loss_pitch + loss_cond).backward()
    out = dict(seed=seed, shapes=json.dumps(shapes), x_char=x_char, x_speaker=x_speaker, y_dur=y_dur, y_pitch=y_pitch,
               f2p_flat=np.concatenate([np.asarray(f) for f in f2ps]), f2p_len=np.asarray([len(f) for f in f2ps]),
               p_dur=p_dur.detach().numpy(), p_pitch=p_pitch.detach().numpy(), p_vuv=p_vuv.detach().numpy(),
               conditioning=conditioning.detach().numpy(), cond_probe=R.numpy(),
               loss_duration=float(loss_duration), loss_pitch=float(loss_pitch), loss_cond=float(loss_cond),
               cfg=json.dumps(dict(num_phones=num_phones, num_speakers=num_speakers, max_pitch=max_pitch, max_duration=max_duration)))
    names = []
    for k, p in net.named_parameters():
        if k.startswith('_lm_'):
            continue          # the Linear(1, 1) placeholders of cond_type=None never see a gradient
        assert p.grad is not None, k
        names.append(k)
        for fk, fv in fingerprint(p.grad.numpy(), k).items():
            out['grad/%s/%s' % (k, fk)] = fv
    out['grad_names'] = json.dumps(names)
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **out)
    print(name, 'B', B, 'frames', conditioning.shape[1], 'loss_dur %.5f loss_pitch %.5f loss_cond %.3e' %
          (float(loss_duration), float(loss_pitch), float(loss_cond)), len(names), 'gradient tensors')


def gen_languasito_fasttext(name, seed, nph, nw, num_phones=50, num_speakers=3, max_pitch=300, max_duration=12):
    net, shapes = _lang(seed, num_phones, num_speakers, max_pitch, max_duration, cond_type='fasttext')
    net.eval()
    rng = np.random.RandomState(seed)
    x_char = torch.from_numpy(rng.randint(1, num_phones + 1, size=(1, nph))).long()
    x_speaker = torch.tensor([[1]]).long()
    x_words = torch.from_numpy(rng.randn(1, nw, 300).astype(np.float32) * 0.3)
    p2w = np.sort(rng.randint(0, nw, size=(1, nph)), axis=1).astype(np.int64)      # monotone phoneme -> word map
    X = {'x_char': x_char, 'x_speaker': x_speaker, 'x_words': x_words, 'x_phon2word': torch.from_numpy(p2w), 'y_frame2phone': [[0]]}
    with torch.no_grad():
        cond = net.inference(X)
    f2p = X['y_frame2phone'][0]
    durs = np.bincount(np.asarray(f2p, dtype=np.int64), minlength=nph) if len(f2p) else np.zeros(nph, dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, name + '.npz'), seed=seed, shapes=json.dumps(shapes), x_char=x_char.numpy(),
                        x_speaker=x_speaker.numpy(), x_words=x_words.numpy(), x_phon2word=p2w, cond=cond.numpy(), durs=durs,
                        pitch=X['y_pitch'].numpy(),
                        cfg=json.dumps(dict(num_phones=num_phones, num_speakers=num_speakers, max_pitch=max_pitch,
                                            max_duration=max_duration)))
    print(name, 'frames', cond.shape[1], 'cond rms', float(cond.pow(2).mean().sqrt()), 'durs', durs[:10])


def gen_languasito_long(name, seed, nph, num_phones=50, num_speakers=3, max_pitch=300, max_duration=12):
    net, shapes = _lang(seed, num_phones, num_speakers, max_pitch, max_duration)
    net.eval()
    rng = np.random.RandomState(seed)
    x_char = torch.from_numpy(rng.randint(1, num_phones + 1, size=(1, nph))).long()
    x_speaker = torch.tensor([[3]]).long()
    X = {'x_char': x_char, 'x_speaker': x_speaker, 'y_frame2phone': [[0]]}
    with torch.no_grad():
        cond = net.inference(X)
    f2p = X['y_frame2phone'][0]
    durs = np.bincount(np.asarray(f2p, dtype=np.int64), minlength=nph)
    np.savez_compressed(os.path.join(OUT, name + '.npz'), seed=seed, shapes=json.dumps(shapes), x_char=x_char.numpy(),
                        x_speaker=x_speaker.numpy(), cond=cond.numpy(), durs=durs, pitch=X['y_pitch'].numpy(),
                        cfg=json.dumps(dict(num_phones=num_phones, num_speakers=num_speakers, max_pitch=max_pitch,
                                            max_duration=max_duration)))
    print(name, 'phonemes', nph, 'frames', cond.shape[1], 'cond rms', float(cond.pow(2).mean().sqrt()))


def gen_vocoder_step(name, seed, H, N, B, L, steps=2, lr=1e-3, out_gain=1.0):
    """CubenetVocoder.training_step, unmodified (vocoder.py:136-156).  The stub LightningModule has no trainer: `optimizers()` is supplied
    on the instance from the reference's own configure_optimizers (vocoder.py:167-171); `torch.nn.utils.clip_grad_norm` (removed from
    torch 2.x) is aliased to `clip_grad_norm_` in torch's namespace — the reference source is untouched."""
    if not hasattr(torch.nn.utils, 'clip_grad_norm'):
        torch.nn.utils.clip_grad_norm = torch.nn.utils.clip_grad_norm_
    torch.manual_seed(0)
    voc = CubenetVocoder(num_layers_lr=N, layer_size_lr=H, num_layers_hr=N, layer_size_hr=H, upsample=240, upsample_low=10,
                         learning_rate=lr, output='mulaw')
    sd = {}
    for pre, low, s in (('_wavernn_hr.', True, seed), ('_wavernn_lr.', False, seed + 100)):
        for k, v in O.synthetic_state_dict(H=H, num_layers=N, use_lowres=low, seed=s).items():
            sd[pre + k] = torch.from_numpy(v)
            if k.startswith('_output.linear_layer.weight'):
                sd[pre + k] = sd[pre + k] * out_gain     # sharper logits -> gradient norms beyond 5, so that clip_grad_norm(5) acts
    voc.load_state_dict(sd, strict=True)
    voc.train()
    opts = voc.configure_optimizers()
    voc.optimizers = lambda: opts
    norms = []
    real_clip = torch.nn.utils.clip_grad_norm

    def recording_clip(params, max_norm):
        n = real_clip(params, max_norm)
        norms.append(float(n))
        return n

    torch.nn.utils.clip_grad_norm = recording_clip
    rng = np.random.RandomState(seed)
    out = dict(seed=seed, H=H, N=N, lr=lr, steps=steps, out_gain=out_gain)
    try:
        for s in range(steps):
            x = (rng.uniform(-1, 1, size=(B, L)) * 0.9).astype(np.float32)
            x_low = (rng.uniform(-1, 1, size=(B, L // 10)) * 0.9).astype(np.float32)
            mel = np.clip(rng.randn(B, L // 240 + 1, 80) - 2, -5, 1).astype(np.float32)
            batch = {'x': torch.from_numpy(x), 'x_low': torch.from_numpy(x_low), 'mel': torch.from_numpy(mel)}
            loss = voc.training_step(batch, s)
            out['x%d' % s], out['x_low%d' % s], out['mel%d' % s] = x, x_low, mel
            out['loss_lr%d' % s], out['loss_hr%d' % s], out['alpha%d' % s] = float(loss['lr']), float(loss['hr']), float(loss['alpha'])
            out['norm_lr%d' % s], out['norm_hr%d' % s] = norms[2 * s], norms[2 * s + 1]
            # every parameter after the step: as fingerprints, and in full after the last step
            for k, v in voc.state_dict().items():
                if s == steps - 1:
                    out['p%d/%s' % (s, k)] = v.detach().numpy().copy()
                else:
                    for fk, fv in fingerprint(v.detach().numpy(), k).items():
                        out['fp%d/%s/%s' % (s, k, fk)] = fv
            print(name, 'step', s, 'loss lr %.5f hr %.5f' % (float(loss['lr']), float(loss['hr'])), 'norms', norms[2 * s:2 * s + 2],
                  'alpha', float(loss['alpha']))
    finally:
        torch.nn.utils.clip_grad_norm = real_clip
    out['keys'] = json.dumps(list(voc.state_dict().keys()))
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **out)


if __name__ == '__main__':
    os.makedirs(OUT, exist_ok=True)
    gen_languasito_train('languasito2_train_a', 51, [23])
    gen_languasito_train('languasito2_train_b', 52, [19, 11])
    gen_languasito_fasttext('languasito2_ft_a', 61, 21, 6)
    gen_languasito_long('languasito2_long', 71, 64)
    gen_vocoder_step('vocoder_step_h64', 81, 64, 1, 2, 720)
    gen_vocoder_step('vocoder_step_h64_clipped', 82, 64, 2, 3, 480, out_gain=12.0)
