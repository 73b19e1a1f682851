"""GPU: the BASELINE.json configurations AT SIZE (VERDICT r2 "next" #1) — the other GPU tests pin the arithmetic on small
cases; these run the shapes the metric is quoted on and check them against the oracles:

  C3  WaveRNN decode, B = 256 utterances, H = 512, T = 100 frames = 24 000 autoregressive steps (cube/networks/modules.py:453-503):
      sample indices AND waveform bit-exact against oracle/wavernn_ref.c on utterances {0, 7, 128, 255} over the WHOLE decode (the
      counter-based noise carries the utterance index, so the oracle runs those four alone); the same for a two-layer net (the
      reference class default, modules.py:392-400), also on the tile kernel.
  C5  per-GPU share of the end-to-end path: 64 ragged sentences (20..120 phonemes) through Cubegan.inference (cubegan.py:74-83):
      {shortest, longest, 2 random} against the meldecoder_ref -> hifigan_ref oracle chain (identical durations, <= 4 LSB int16),
      all 64 against their solo runs.
  C4  per-GPU share of train_cubegan.py: ONE Cubegan.training_step (cubegan.py:85-189) at b = 16 (the reference script's default) AND at b = 128 (the
      per-GPU batch configs[3] names) whose losses equal the torch-op formulation of the same step (torch.nn.LSTM, F.conv1d generator and
      discriminators, torch.stft mel) within 1e-4 relative.
  C1  the reference's own CPU-runnable case: ONE utterance of 300 frames (3 s) through Generator.forward — all 72 064 samples against the oracle.
"""
import random
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest
import torch

from oracle import hifigan_ref as R
from oracle import meldecoder_ref as M
from oracle import wavernn_ref as O

pytestmark = pytest.mark.gpu


def _wavernn(H, N, sd):
    from ttscube_amd.networks.modules import WaveRNN
    net = WaveRNN(num_layers=N, layer_size=H, upsample=240, upsample_low=10, use_lowres=True, output='mulaw')
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    return net.cuda().eval()


@pytest.mark.parametrize('N,T,check,kernel', [(1, 100, (0, 7, 128, 255), 'tile'), (2, 100, (3, 200), 'tile')])
def test_c3_wavernn_b256_h512_full_length_bit_exact(N, T, check, kernel):
    B, H, seed = 256, 512, 0x5EED00C3
    sd = O.synthetic_state_dict(H=H, num_layers=N, use_lowres=True, seed=31 + N)
    net = _wavernn(H, N, sd)
    mel, x_low = O.synthetic_inputs(B, T, seed=6)
    assert mel.shape == (B, T, 80) and x_low.shape == (B, T * 24)
    idx, wav, _ = net.decode({'mel': torch.from_numpy(mel), 'x_low': torch.from_numpy(x_low)}, mode='philox', seed=seed)
    assert net.last_kernel == kernel
    idx, wav = idx.cpu().numpy(), wav.cpu().numpy()
    assert idx.shape == (B, T * 240)

    def oracle(b):   # ctypes releases the GIL: the utterances run on separate host cores
        return O.decode(sd, mel[b:b + 1], x_low[b:b + 1], num_layers=N, H=H, mode=O.MODE_PHILOX, seed=seed, b_offset=b)

    with ThreadPoolExecutor(len(check)) as ex:
        refs = list(ex.map(oracle, check))
    for b, (ridx, rwav, _) in zip(check, refs):
        bad = np.flatnonzero(idx[b] != ridx[0])
        assert bad.size == 0, 'utterance %d: first differing step %d of %d' % (b, bad[0], ridx.shape[1])
        assert np.array_equal(wav[b], rwav[0])
    # not a degenerate decode: the sampled utterances differ from each other and use the index range
    assert len({idx[b].tobytes() for b in check}) == len(check) and np.unique(idx[list(check)]).size > 64


class _Enc:
    phon2int = {'p%d' % i: i for i in range(50)}
    speaker2int = {'s0': 0}
    max_pitch = 300
    max_duration = 12   # synthetic weights give ~uniform durations: ~6 frames per phoneme


def _c5_model():
    from ttscube_amd.networks.cubegan import Cubegan
    torch.manual_seed(0)
    tts = Cubegan(_Enc(), conditioning=None, train=False)
    lsd = M.fill_state_dict(M.named_shapes(tts._languasito), 5)
    gsd = R.synthetic_state_dict(dict(R.CONFIG_V1), seed=1234)
    sd = tts.state_dict()
    sd.update({'_languasito.' + k: v for k, v in lsd.items()})
    sd.update({'_generator.' + k: v for k, v in gsd.items()})
    tts.load_state_dict(sd)
    return tts.cuda().eval(), lsd, gsd


def test_c5_64_ragged_sentences_match_oracle_chain_and_solo_runs():
    from ttscube_amd.io_utils.synthetic import synthetic_sentences
    tts, lsd, gsd = _c5_model()
    xc, lens = synthetic_sentences(64, seed=1234)
    assert lens.min() >= 20 and lens.max() <= 120
    X = {'x_char': torch.from_numpy(xc), 'x_speaker': torch.ones((64, 1), dtype=torch.long)}
    wav, wl = tts.inference(X, return_lengths=True)
    wav = wav.cpu().numpy()
    assert wav.shape[0] == 64 and np.isfinite(wav).all() and min(wl) > 0
    to16 = lambda a: np.asarray(a * 32767, dtype=np.int16)   # TTSCube.__call__'s conversion (api.py:65)
    # every sentence alone == the sentence inside the padded batch (same kernels inside the resident ranges)
    for b in range(64):
        solo = tts.inference({'x_char': torch.from_numpy(xc[b:b + 1, :lens[b]]), 'x_speaker': torch.ones((1, 1), dtype=torch.long)})
        assert solo.shape[2] == wl[b], (b, solo.shape, wl[b])
        assert np.array_equal(to16(solo.cpu().numpy()[0, 0]), to16(wav[b, 0, :wl[b]])), 'sentence %d: batch != solo' % b
    rs = np.random.RandomState(7)
    picks = sorted({int(np.argmin(lens)), int(np.argmax(lens)), *map(int, rs.choice(64, size=2, replace=False))})
    w = R.fold_state_dict(gsd)
    for b in picks:
        with torch.no_grad():
            cond, durs, _ = M.languasito2_inference(lsd, torch.from_numpy(xc[b:b + 1, :lens[b]]), torch.tensor([[1]]), _Enc.max_pitch)
            ref = R.generator_forward(w, dict(R.CONFIG_V1), cond.permute(0, 2, 1))
        assert wl[b] == 240 * sum(durs) + 64, 'sentence %d: durations differ from the oracle' % b
        d = np.abs(to16(wav[b, 0, :wl[b]]).astype(np.int32) - to16(ref.numpy().squeeze()).astype(np.int32))
        assert d.max() <= 4, (b, int(d.max()))   # 1e-4 of full scale = 3.3 LSB


def test_c1_single_3s_utterance_whole_output_matches_the_oracle():
    """BASELINE configs[0] shape on the GPU: [1, 80, 300] -> [1, 1, 72064], the call shape of cube/io_utils/runtime.py:78 and cubegan.py:83 —
    every sample against oracle/hifigan_ref.py (1e-4 RMS gate of north_star; max error reported too), both precisions, and the int16 the API
    hands out within 4 LSB"""
    from ttscube_amd.hifigan.env import AttrDict
    from ttscube_amd.hifigan.models import Generator
    h = dict(R.CONFIG_V1)
    sd = R.synthetic_state_dict(h, seed=1234)
    g = Generator(AttrDict(h))
    g.load_state_dict(sd)
    g = g.cuda().eval()
    mel = R.synthetic_mel(1, 300, seed=77)
    with torch.no_grad():
        ref = R.generator_forward(R.fold_state_dict(sd), h, mel)
    assert tuple(ref.shape) == (1, 1, 72064)
    for prec in ('f16x3', 'fp32'):
        g.set_precision(prec)
        with torch.no_grad():
            out = g(mel.cuda()).cpu()
        assert out.shape == ref.shape
        rms = float((out - ref).pow(2).mean().sqrt())
        assert rms < 1e-4 and float((out - ref).abs().max()) < 1e-4, (prec, rms, float((out - ref).abs().max()))
        to16 = lambda a: np.asarray(a.numpy().squeeze() * 32767, dtype=np.int16).astype(np.int32)
        assert int(np.abs(to16(out) - to16(ref)).max()) <= 4


@pytest.mark.parametrize('b', [16, 128])
def test_c4_training_step_losses_match_torch_formulation(monkeypatch, b):
    from ttscube_amd.io_utils.io_cubegan import CubeganCollate
    from ttscube_amd.io_utils.synthetic import synthetic_encodings, synthetic_examples
    from ttscube_amd.networks import training as T
    from ttscube_amd.networks.cubegan import Cubegan
    enc = synthetic_encodings()
    torch.manual_seed(1234)
    model = Cubegan(enc, conditioning=None, train=True).cuda()
    model.train()
    twin = Cubegan(enc, conditioning=None, train=True).cuda()   # (weight-normed modules cannot be deep-copied)
    twin.load_state_dict(model.state_dict())
    twin.train()
    batch = CubeganCollate(enc).collate_fn(list(synthetic_examples(b, 777, min_ph=30, max_ph=50)))
    assert batch['x_char'].shape[0] == b
    out = model.training_step(batch, 0, rng=random.Random(99))   # the class surface (cubegan.py:85) -> networks/training.py::cubegan_training_step
    assert all(np.isfinite(v) for v in out.values())
    # the same step with every native piece swapped for its torch-op formulation, same weights, same crops
    from tests import torch_reference
    torch_reference.install(monkeypatch)   # generator, discriminators, LSTMs, text stacks, mel loss, GAN losses, AdamW: all torch ops
    ref = T.cubegan_training_step(twin, batch, T.cubegan_configure_optimizers(twin), rng=random.Random(99))
    for k in ('loss_d', 'loss_g', 'loss_t', 'loss_mel'):
        assert abs(out[k] - ref[k]) <= 1e-4 * abs(ref[k]), (k, out[k], ref[k])
