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
