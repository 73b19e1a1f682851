"""GPU parity against vectors made by the REFERENCE ITSELF for the training-side functions and the external-conditioning branch
(tools/gen_golden_training.py, build container; VERDICT r5 "What's missing" #1-2):

  * Languasito2.forward teacher-forced (cube/networks/modules.py:996-999) + the text losses (cube/networks/cubegan.py:94-112) + the gradient
    of every parameter: the HIP autograd path (networks/training.py::languasito_forward_train, text_losses) vs the reference's own autograd;
  * Languasito2(cond_type='fasttext').inference (modules.py:932-940, 1079-1082) and a 64-phoneme sentence;
  * CubenetVocoder.training_step (cube/networks/vocoder.py:136-156), two consecutive steps: losses, gradient norms before clipping,
    learning rate, every parameter afterwards.

Gates: gradients <= 1e-4 relative (norm, seeded probe and 128 samples of every tensor: oracle/fingerprint.py), identical durations,
outputs <= 1e-4."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import meldecoder_ref as M
from oracle import wavernn_ref as O
from oracle.fingerprint import compare

pytestmark = pytest.mark.gpu


def _load(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name + '.npz'))
    shapes = [(k, tuple(s)) for k, s in json.loads(str(z['shapes']))]
    return z, shapes, M.fill_state_dict(shapes, int(z['seed']))


def _f2ps(z):
    out, o = [], 0
    for n in z['f2p_len']:
        out.append([int(v) for v in z['f2p_flat'][o:o + int(n)]])
        o += int(n)
    return out


def _lang(z, shapes, sd, cond_type=None):
    from ttscube_amd.networks.modules import Languasito2
    cfg = json.loads(str(z['cfg']))
    net = Languasito2(cfg['num_phones'], cfg['num_speakers'], cfg['max_pitch'], cfg['max_duration'], cond_type=cond_type)
    assert M.named_shapes(net) == shapes           # state_dict layout == the reference's (names, shapes, order)
    net.load_state_dict(sd, strict=True)
    return net.cuda(), cfg


@pytest.mark.parametrize('name', ['languasito2_train_a', 'languasito2_train_b'])
def test_languasito2_training_forward_losses_and_gradients_match_the_reference(golden_dir, name):
    from ttscube_amd.networks import training as T
    z, shapes, sd = _load(golden_dir, name)
    net, cfg = _lang(z, shapes, sd)
    net.train()
    X = {'x_char': torch.from_numpy(z['x_char']), 'x_speaker': torch.from_numpy(z['x_speaker']), 'y_frame2phone': _f2ps(z),
         'y_pitch': torch.from_numpy(z['y_pitch']), 'y_dur': torch.from_numpy(z['y_dur'])}
    p_dur, p_pitch, p_vuv, cond = net(X)                   # Languasito2.forward -> the differentiable HIP path (training mode, grad enabled)
    assert p_dur.requires_grad and cond.requires_grad
    for got, key in ((p_dur, 'p_dur'), (p_pitch, 'p_pitch'), (p_vuv, 'p_vuv'), (cond, 'conditioning')):
        assert got.shape == z[key].shape, key
        assert float((got.detach().cpu() - torch.from_numpy(z[key])).abs().max()) < 1e-4, key
    l_dur, l_pitch = T.text_losses(p_dur, p_pitch, p_vuv, X['y_dur'].cuda(), X['y_pitch'].cuda(), cfg['max_pitch'],
                                   int(max(cfg['max_pitch'], cfg['max_duration']) + 1))
    assert abs(float(l_dur.detach()) - float(z['loss_duration'])) < 1e-4 and abs(float(l_pitch.detach()) - float(z['loss_pitch'])) < 1e-4
    l_cond = (cond * torch.from_numpy(z['cond_probe']).cuda()).sum() / cond.numel()
    (l_dur + l_pitch + l_cond).backward()
    torch.cuda.synchronize()
    params = dict(net.named_parameters())
    bad = {}
    for k in json.loads(str(z['grad_names'])):
        assert params[k].grad is not None, k
        fp = {f: z['grad/%s/%s' % (k, f)] for f in ('norm', 'sum', 'probe', 'idx', 'samples', 'size')}
        dev = compare(params[k].grad.cpu().numpy(), k, fp)
        if max(dev.values()) > 1e-4:
            bad[k] = dev
    assert not bad, bad


def test_languasito2_teacher_forced_forward_on_the_inference_kernels_matches_the_reference(golden_dir):
    """eval mode / no grad: Languasito2.forward runs the inference kernels (validation, forced-alignment synthesis: cubegan.py:65-72)"""
    z, shapes, sd = _load(golden_dir, 'languasito2_train_a')
    net, cfg = _lang(z, shapes, sd)
    net.eval()
    X = {'x_char': torch.from_numpy(z['x_char']), 'x_speaker': torch.from_numpy(z['x_speaker']), 'y_frame2phone': _f2ps(z),
         'y_pitch': torch.from_numpy(z['y_pitch'])}
    with torch.no_grad():
        outs = net(X)
    for got, key in zip(outs, ('p_dur', 'p_pitch', 'p_vuv', 'conditioning')):
        assert got.shape == z[key].shape, key
        assert float((got.cpu() - torch.from_numpy(z[key])).abs().max()) < 1e-4, key


def test_languasito2_external_conditioning_matches_the_reference(golden_dir):
    """cond_type='fasttext': the `_lm_t/_lm_g` BiLSTMs over the word vectors + `_get_cond_selection` (modules.py:932-940, 1079-1082)"""
    z, shapes, sd = _load(golden_dir, 'languasito2_ft_a')
    net, cfg = _lang(z, shapes, sd, cond_type='fasttext')
    net.eval()
    X = {'x_char': torch.from_numpy(z['x_char']), 'x_speaker': torch.from_numpy(z['x_speaker']), 'x_words': torch.from_numpy(z['x_words']),
         'x_phon2word': torch.from_numpy(z['x_phon2word']), 'y_frame2phone': [[0]]}
    cond = net.inference(X).cpu()
    durs = np.bincount(np.asarray(X['y_frame2phone'][0], dtype=np.int64), minlength=z['x_char'].shape[1])
    assert list(durs) == list(z['durs'])
    assert cond.shape == z['cond'].shape
    assert float((cond - torch.from_numpy(z['cond'])).pow(2).mean().sqrt()) < 1e-4
    assert float((X['y_pitch'].cpu() - torch.from_numpy(z['pitch'])).abs().max()) < 1e-2


def test_languasito2_long_sentence_matches_the_reference(golden_dir):
    z, shapes, sd = _load(golden_dir, 'languasito2_long')
    net, cfg = _lang(z, shapes, sd)
    net.eval()
    X = {'x_char': torch.from_numpy(z['x_char']), 'x_speaker': torch.from_numpy(z['x_speaker']), 'y_frame2phone': [[0]]}
    cond = net.inference(X).cpu()
    durs = np.bincount(np.asarray(X['y_frame2phone'][0], dtype=np.int64), minlength=z['x_char'].shape[1])
    assert list(durs) == list(z['durs']) and cond.shape == z['cond'].shape
    assert float((cond - torch.from_numpy(z['cond'])).pow(2).mean().sqrt()) < 1e-4
    assert float((X['y_pitch'].cpu() - torch.from_numpy(z['pitch'])).abs().max()) < 1e-2


@pytest.mark.parametrize('name', ['vocoder_step_h64', 'vocoder_step_h64_clipped'])
def test_vocoder_training_step_matches_the_reference(golden_dir, name, monkeypatch):
    """CubenetVocoder.training_step through the HIP GRU / GEMM / convolution kernels behind autograd vs the reference's own two steps
    (second case: gradient norms of 100-190, so clip_grad_norm(5) acts)."""
    from ttscube_amd.networks.vocoder import CubenetVocoder
    z = np.load(os.path.join(golden_dir, name + '.npz'))
    H, N, steps, lr = int(z['H']), int(z['N']), int(z['steps']), float(z['lr'])
    voc = CubenetVocoder(num_layers_lr=N, layer_size_lr=H, num_layers_hr=N, layer_size_hr=H, upsample=240, upsample_low=10, learning_rate=lr,
                         output='mulaw')
    sd = {}
    for pre, low, s in (('_wavernn_hr.', True, int(z['seed'])), ('_wavernn_lr.', False, int(z['seed']) + 100)):
        for k, v in O.synthetic_state_dict(H=H, num_layers=N, use_lowres=low, seed=s).items():
            sd[pre + k] = torch.from_numpy(v) * (float(z['out_gain']) if k == '_output.linear_layer.weight' else 1.0)
    assert list(voc.state_dict().keys()) == json.loads(str(z['keys']))
    voc.load_state_dict(sd, strict=True)
    voc = voc.cuda().train()
    norms = []
    real = torch.nn.utils.clip_grad_norm_

    def recording(params, max_norm, *a, **k):
        n = real(params, max_norm, *a, **k)
        norms.append(float(n))
        return n

    monkeypatch.setattr(torch.nn.utils, 'clip_grad_norm_', recording)
    for s in range(steps):
        batch = {k: torch.from_numpy(z['%s%d' % (k, s)]) for k in ('x', 'x_low', 'mel')}
        out = voc.training_step(batch, s)
        for key, got in (('loss_lr', out['lr']), ('loss_hr', out['hr']), ('norm_lr', norms[2 * s]), ('norm_hr', norms[2 * s + 1])):
            ref = float(z['%s%d' % (key, s)])
            assert abs(float(got) - ref) < 2e-4 * max(1.0, abs(ref)), (s, key, float(got), ref)
        assert abs(out['alpha'] - float(z['alpha%d' % s])) < 1e-15
    # every parameter after the last step.  One Adam step moves a weight by ~lr = 1e-3 whatever the gradient's size, so an element whose gradient
    # is at the rounding floor may land anywhere within a step: the bulk is held to 2 % of a step, stragglers to one step.
    got = voc.state_dict()
    for k in json.loads(str(z['keys'])):
        d = (got[k].cpu() - torch.from_numpy(z['p%d/%s' % (steps - 1, k)])).abs()
        assert float(d.max()) < 1.1 * lr * steps, (k, float(d.max()))
        assert float((d > 2e-5).float().mean()) < 2e-3, (k, float((d > 2e-5).float().mean()), float(d.max()))
