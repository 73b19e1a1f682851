"""Oracle pinning (CPU): the C WaveRNN restatement vs vectors produced by the reference itself
(tools/gen_golden_wavernn.py imports /root/reference and replays torch's real sampler noise)."""
import os

import numpy as np
import pytest

from oracle import wavernn_ref as O

CASES = ['wavernn_hr_h64_n1', 'wavernn_hr_h64_n2', 'wavernn_lr_h64_n1', 'wavernn_hr_h512_n1', 'wavernn_hr_h64_raw']


def _case(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name + '.npz'))
    kw = dict(num_layers=int(z['N']), H=int(z['H']), use_lowres=bool(z['use_lowres']),
              upsample=240 if bool(z['use_lowres']) else 24, output=str(z['output']))
    sd = O.synthetic_state_dict(H=kw['H'], num_layers=kw['num_layers'], use_lowres=kw['use_lowres'], seed=int(z['seed']))
    return z, sd, kw


def test_mulaw_known_answers(golden_dir):
    lut = O.mulaw_lut(golden_dir)
    # SURVEY.md §4 known answers captured by import from cube/networks/loss.py:236-269
    assert lut[0] == -1.0 and lut[255] == 1.0
    assert float(lut[1]).hex() == '-0x1.ea1fd00000000p-1' and float(lut[254]).hex() == '0x1.ea1fd00000000p-1'
    assert float(lut[127]).hex() == '-0x1.69991a0000000p-14' and float(lut[128]).hex() == '0x1.699a9a0000000p-14'
    assert list(O.mulaw_encode(np.array([1, .9, 0, -.9, -1], dtype=np.float32))) == [255, 253, 128, 2, 0]
    z = np.load(os.path.join(golden_dir, 'mulaw_kat.npz'))
    assert np.array_equal(O.mulaw_encode(z['x']), z['enc'])
    assert np.array_equal(O.raw_encode(z['x']), z['enc_raw'])
    # encode(decode(i)) == i for every code: the LUT is a right inverse of the encoder
    assert np.array_equal(O.mulaw_encode(lut), np.arange(256))


@pytest.mark.parametrize('name', CASES)
def test_sampled_indices_match_reference(golden_dir, name):
    """Same weights, same inputs, the reference's own sampler noise -> identical samples, every step."""
    z, sd, kw = _case(golden_dir, name)
    idx, wav, _ = O.decode(sd, z['mel'], z['x_low'] if kw['use_lowres'] else None, mode=O.MODE_NOISE, noise=z['gumbel'], **kw)
    assert idx.shape == z['idx'].shape
    assert np.array_equal(wav, z['wav']), 'first mismatch at %s' % (np.argwhere(wav != z['wav'])[:3],)
    if kw['output'] == 'mulaw':
        assert np.array_equal(idx, z['idx'])


@pytest.mark.parametrize('name', CASES)
def test_teacher_forced_logits_match_reference(golden_dir, name):
    """forced feedback == WaveRNN._train_forward (modules.py:505-539): logits within 1e-4 max-abs."""
    z, sd, kw = _case(golden_dir, name)
    _, _, logits = O.decode(sd, z['mel'], z['x_low'] if kw['use_lowres'] else None, mode=O.MODE_ARGMAX,
                            forced_x=z['audio'], want_logits=True, **kw)
    err = float(np.abs(logits - z['logits_tf']).max())
    assert err < 1e-4, err
    # MULAWOutput.loss / RAWOutput.loss = mean CE over B*L (loss.py:222-225)
    tgt = O.mulaw_encode(z['audio']) if kw['output'] == 'mulaw' else O.raw_encode(z['audio'])
    lg = logits.astype(np.float64)
    lse = np.log(np.exp(lg - lg.max(-1, keepdims=True)).sum(-1)) + lg.max(-1)
    ce = (lse - np.take_along_axis(lg, tgt[..., None], -1)[..., 0]).mean()
    assert abs(ce - float(z['loss_tf'])) < 1e-4


def test_shared_math_vs_libm():
    L = O.lib()
    xs = np.linspace(-20, 20, 4001).astype(np.float32)
    for f, ref in (('wr_expf', np.exp), ('wr_tanhf', np.tanh), ('wr_sigmoidf', lambda v: 1 / (1 + np.exp(-v)))):
        got = np.array([getattr(L, f)(float(v)) for v in xs], dtype=np.float64)
        want = ref(xs.astype(np.float64))
        assert np.max(np.abs(got - want) / np.maximum(np.abs(want), 1.0)) < 1e-6, f
    xs = np.exp(np.linspace(-18, 18, 2001)).astype(np.float32)
    got = np.array([L.wr_logf(float(v)) for v in xs], dtype=np.float64)
    assert np.max(np.abs(got - np.log(xs.astype(np.float64)))) < 2e-6


def test_philox_known_answer():
    import ctypes as C
    L = O.lib()
    out = (C.c_uint32 * 4)()
    L.wr_philox(0, 0, 0, 0, 0, 0, out)  # Random123 KAT, philox4x32-10, zero counter/key
    assert [hex(v) for v in out] == ['0x6627e8d5', '0xe169c58d', '0xbc57ac4c', '0x9b00dbd8']
    L.wr_philox(0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff, out)
    assert [hex(v) for v in out] == ['0x408f276d', '0x41c83b0e', '0xa20bc7c6', '0x6d5451fd']


def test_vocoder_fold_matches_reference(golden_dir):
    z = np.load(os.path.join(golden_dir, 'vocoder_fold.npz'))
    f = O.inference_batch(z['mel'], z['x_low'], num_batches=20)
    assert np.array_equal(f['mel'], z['fold_mel']) and np.array_equal(f['x_low'], z['fold_x_low'])
    assert np.array_equal(O.compose_batched_inference(z['hr']), z['composed'])


# ---- continuous outputs (MOL is the reference's default, cube/networks/modules.py:398) -----------------------------------
CONT = ['wavernn_hr_h64_mol', 'wavernn_hr_h512_mol', 'wavernn_lr_h64_gm', 'wavernn_hr_h64_beta']


def _case_cont(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name + '.npz'))
    out = str(z['output'])
    kw = dict(num_layers=int(z['N']), H=int(z['H']), use_lowres=bool(z['use_lowres']),
              upsample=240 if bool(z['use_lowres']) else 24, output=out)
    sd = O.synthetic_state_dict(H=kw['H'], num_layers=kw['num_layers'], use_lowres=kw['use_lowres'], seed=int(z['seed']),
                                S=O.SAMPLE_SIZE[out])
    sd['_output.linear_layer.weight'], sd['_output.linear_layer.bias'] = z['out_w'], z['out_b']   # (rescaled by the generator)
    return z, sd, kw


@pytest.mark.parametrize('name', CONT[:3])
def test_continuous_samples_match_reference(golden_dir, name):
    """MOLOutput.sample / GaussianOutput.sample (loss.py:163-201,50-52) with the reference's own random terms replayed:
    the mixture index of every step is identical and the samples agree to 1e-5 over the whole autoregressive run (the only
    arithmetic difference is one exp() per sample: shared fp32 definition here, libm in torch)."""
    z, sd, kw = _case_cont(golden_dir, name)
    idx, wav, _ = O.decode(sd, z['mel'], z['x_low'] if kw['use_lowres'] else None, mode=O.MODE_NOISE, noise=z['noise'], **kw)
    assert wav.shape == z['wav'].shape
    if kw['output'] == 'mol':
        assert np.array_equal(idx, z['idx']), 'first mixture-index mismatch at %s' % (np.argwhere(idx != z['idx'])[:3],)
    assert float(np.abs(wav - z['wav']).max()) < 1e-5


@pytest.mark.parametrize('name', CONT)
def test_continuous_teacher_forced_outputs_match_reference(golden_dir, name):
    z, sd, kw = _case_cont(golden_dir, name)
    _, _, logits = O.decode(sd, z['mel'], z['x_low'] if kw['use_lowres'] else None, mode=O.MODE_ARGMAX, forced_x=z['audio'],
                            want_logits=True, **kw)
    assert logits.shape == z['logits_tf'].shape
    assert float(np.abs(logits - z['logits_tf']).max()) < 1e-4


def test_normal_quantile_and_beta_sampler_distribution():
    """The Beta sampler cannot replay torch's rejection sampler draw for draw; it is pinned distributionally: Marsaglia-Tsang
    gammas from the counter RNG -> Kolmogorov-Smirnov against scipy's Beta for shapes on both sides of 1 (loss.py:83-92)."""
    import ctypes as C
    from scipy import stats
    L = O.lib()
    ps = np.concatenate([np.linspace(1e-6, 1 - 1e-6, 2001), [1e-7, 0.02425, 0.97575]]).astype(np.float32)
    got = np.array([L.wr_normal_icdf(float(p)) for p in ps])
    assert np.max(np.abs(got - stats.norm.ppf(ps.astype(np.float64)))) < 5e-4   # fp32 evaluation of the rational approximation
    nz = (C.c_float * 18)()
    for a_, b_ in ((0.4, 0.7), (1.0, 1.0), (2.5, 0.8), (5.0, 9.0), (30.0, 30.0)):
        y = (C.c_float * 2)(float(np.log(a_)), float(np.log(b_)))
        xs = []
        for t in range(6000):
            L.wr_noise_beta(t, 3, C.c_uint64(12345), nz)
            xs.append(L.wr_sample_beta(y, nz))
        xs = (np.asarray(xs, dtype=np.float64) + 1.0) / 2.0          # back to (0, 1)
        assert 0.0 <= xs.min() and xs.max() <= 1.0
        ks = stats.kstest(xs, stats.beta(a_, b_).cdf)
        assert ks.pvalue > 1e-3, (a_, b_, ks)


@pytest.mark.parametrize('name', ['vocoder_step_h64', 'vocoder_step_h64_clipped'])
def test_vocoder_training_step_restatement_matches_reference(golden_dir, name):
    """CubenetVocoder.training_step run UNMODIFIED from the imported reference (tools/gen_golden_training.py: two consecutive steps from fixed
    weights; the second case has gradient norms of 100-190 so that clip_grad_norm(5) acts) vs oracle/vocoder_step_ref.py: losses, gradient
    norms before clipping, learning rate, every parameter after every step."""
    import json
    import torch
    from oracle import vocoder_step_ref as V
    from oracle.fingerprint import compare
    z = np.load(os.path.join(golden_dir, name + '.npz'))
    H, N, steps = int(z['H']), int(z['N']), int(z['steps'])
    sd = {}
    for pre, low, s in (('_wavernn_hr.', True, int(z['seed'])), ('_wavernn_lr.', False, int(z['seed']) + 100)):
        for k, v in O.synthetic_state_dict(H=H, num_layers=N, use_lowres=low, seed=s).items():
            sd[pre + k] = torch.from_numpy(v) * (float(z['out_gain']) if k == '_output.linear_layer.weight' else 1.0)
    assert list(sd.keys()) == json.loads(str(z['keys']))
    batches = [{k: torch.from_numpy(z['%s%d' % (k, s)]) for k in ('x', 'x_low', 'mel')} for s in range(steps)]
    recs = V.vocoder_training_steps(sd, batches[:1], float(z['lr']))
    for k in sd:        # after the first step: fingerprints
        fp = {f: z['fp0/%s/%s' % (k, f)] for f in ('norm', 'sum', 'probe', 'idx', 'samples', 'size')}
        assert max(compare(sd[k].numpy(), k, fp).values()) < 1e-5, k
    # second step continues from the reference's state after the first only through the oracle's own Adam moments: run both from scratch
    sd2 = {}
    for pre, low, s in (('_wavernn_hr.', True, int(z['seed'])), ('_wavernn_lr.', False, int(z['seed']) + 100)):
        for k, v in O.synthetic_state_dict(H=H, num_layers=N, use_lowres=low, seed=s).items():
            sd2[pre + k] = torch.from_numpy(v) * (float(z['out_gain']) if k == '_output.linear_layer.weight' else 1.0)
    recs = V.vocoder_training_steps(sd2, batches, float(z['lr']))
    for s, r in enumerate(recs):
        for key in ('loss_lr', 'loss_hr', 'norm_lr', 'norm_hr'):
            assert abs(r[key] - float(z['%s%d' % (key, s)])) < 2e-4 * max(1.0, abs(float(z['%s%d' % (key, s)]))), (s, key, r[key])
        assert abs(r['alpha'] - float(z['alpha%d' % s])) < 1e-15
    for k in sd2:
        ref = torch.from_numpy(z['p%d/%s' % (steps - 1, k)])
        # one Adam step moves a weight by ~lr = 1e-3: the update itself is held to 1 %
        assert float((sd2[k] - ref).abs().max()) < 2e-5, (k, float((sd2[k] - ref).abs().max()))
