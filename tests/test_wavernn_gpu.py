"""GPU parity (through the C ABI): persistent WaveRNN kernel vs the C oracle and the reference-generated goldens.
Bar: uint8 µ-law indices bit-exact under injected noise; logits bit-exact vs the oracle and <=1e-4 vs the reference."""
import os

import numpy as np
import pytest
import torch

from oracle import wavernn_ref as O

pytestmark = pytest.mark.gpu



@pytest.fixture(params=['tile', 'stream'], autouse=True)
def wr_kernel(request, monkeypatch):
    """Every test runs twice: with the multi-workgroup tile kernel allowed (its default: one-layer nets, B <= 256) and with the single-workgroup streaming kernel forced (TTSC_WR_TILE=0) — both must be bit-exact."""
    if request.param == 'tile':
        monkeypatch.delenv('TTSC_WR_TILE', raising=False)
    else:
        monkeypatch.setenv('TTSC_WR_TILE', '0')
    return request.param


CASES = ['wavernn_hr_h64_n1', 'wavernn_hr_h64_n2', 'wavernn_lr_h64_n1', 'wavernn_hr_h512_n1', 'wavernn_hr_h64_raw']


def _net(H, N, use_lowres, sd, output='mulaw'):
    from ttscube_amd.networks.modules import WaveRNN
    net = WaveRNN(num_layers=N, layer_size=H, upsample=240 if use_lowres else 24, upsample_low=10, use_lowres=use_lowres,
                  output=output)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    return net.cuda().eval()


@pytest.mark.parametrize('name', CASES)
def test_reference_goldens_bit_exact(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name + '.npz'))
    H, N, lowres, out = int(z['H']), int(z['N']), bool(z['use_lowres']), str(z['output'])
    sd = O.synthetic_state_dict(H=H, num_layers=N, use_lowres=lowres, seed=int(z['seed']))
    net = _net(H, N, lowres, sd, out)
    X = {'mel': torch.from_numpy(z['mel'])}
    if lowres:
        X['x_low'] = torch.from_numpy(z['x_low'])
    idx, wav, _ = net.decode(X, mode='noise', noise=z['gumbel'])
    assert np.array_equal(wav.cpu().numpy(), z['wav'])  # the reference's own samples, every step
    if out == 'mulaw':
        assert np.array_equal(idx.cpu().numpy(), z['idx'])
    # module-level API parity: forward() returns numpy [B, L, 1]
    y = net._inference(X, mode='noise', noise=z['gumbel'])
    assert isinstance(y, np.ndarray) and y.shape == z['wav'].shape + (1,)
    # teacher-forced logits vs the reference's _train_forward
    xin = np.concatenate([np.zeros_like(z['audio'][:, :1]), z['audio'][:, :-1]], axis=1)
    Xt = dict(X)
    Xt['x'] = torch.from_numpy(xin).cuda()
    logits = net(Xt).cpu().numpy()
    assert float(np.abs(logits - z['logits_tf']).max()) < 1e-4


@pytest.mark.parametrize('H,N,lowres,B,T,mode', [
    (512, 1, True, 3, 2, 'noise'), (512, 2, True, 2, 1, 'noise'), (512, 1, False, 4, 8, 'noise'),
    (64, 1, True, 5, 3, 'philox'), (512, 1, True, 2, 1, 'philox'), (512, 1, True, 9, 1, 'noise'), (256, 1, False, 6, 4, 'philox'), (128, 2, False, 2, 5, 'argmax'), (512, 1, True, 2, 1, 'argmax'),
])
def test_matches_oracle_bit_exact(H, N, lowres, B, T, mode, wr_kernel):
    sd = O.synthetic_state_dict(H=H, num_layers=N, use_lowres=lowres, seed=100 + H + N)
    net = _net(H, N, lowres, sd)
    mel, x_low = O.synthetic_inputs(B, T, seed=7 + T, upsample=240 if lowres else 24)
    X = {'mel': torch.from_numpy(mel)}
    if lowres:
        X['x_low'] = torch.from_numpy(x_low)
    up = 240 if lowres else 24
    L = T * up
    noise = None
    if mode == 'noise':
        u = np.random.RandomState(5).uniform(1e-6, 1 - 1e-6, size=(B, L, 256))
        noise = (-np.log(-np.log(u))).astype(np.float32)
    omode = {'noise': O.MODE_NOISE, 'philox': O.MODE_PHILOX, 'argmax': O.MODE_ARGMAX}[mode]
    ridx, rwav, rlog = O.decode(sd, mel, x_low if lowres else None, num_layers=N, H=H, use_lowres=lowres, upsample=up,
                                mode=omode, noise=noise, seed=0xC0FFEE12345, want_logits=True)
    idx, wav, logits = net.decode(X, mode=mode, noise=noise, seed=0xC0FFEE12345, want_logits=True)
    assert net.last_kernel == ('tile' if (wr_kernel == 'tile' and N <= 2 and B <= 256) else 'stream')
    assert np.array_equal(idx.cpu().numpy(), ridx), 'first index mismatch at %s' % (np.argwhere(idx.cpu().numpy() != ridx)[:3],)
    assert np.array_equal(wav.cpu().numpy(), rwav)
    assert np.array_equal(logits.cpu().numpy(), rlog)  # logits themselves are bit-exact


@pytest.mark.parametrize('bt,B,H,N,mode', [(2, 5, 64, 1, 'noise'), (4, 7, 64, 2, 'philox'), (8, 13, 128, 1, 'noise'), (2, 3, 512, 1, 'philox'),
                                           (4, 6, 512, 2, 'argmax'), (8, 9, 512, 1, 'noise'), (0, 300, 64, 1, 'philox'), (0, 600, 64, 1, 'argmax')])
def test_utterance_tiles_bit_exact(bt, B, H, N, mode, monkeypatch, wr_kernel):
    """wr_decode_kernel<2/4/8> (several utterances per workgroup sharing one weight stream): forced through TTSC_WR_BT with a
    batch that is NOT a multiple of the tile, and selected by the dispatcher itself at B=300 (BT=2) / B=600 (BT=4)."""
    if wr_kernel == 'tile':
        pytest.skip('the utterance-tile template belongs to the streaming kernel')
    if bt:
        monkeypatch.setenv('TTSC_WR_BT', str(bt))
    else:
        monkeypatch.delenv('TTSC_WR_BT', raising=False)
    sd = O.synthetic_state_dict(H=H, num_layers=N, use_lowres=True, seed=500 + H + N + bt)
    net = _net(H, N, True, sd)
    T = 1
    mel, x_low = O.synthetic_inputs(B, T, seed=13 + B)
    X = {'mel': torch.from_numpy(mel), 'x_low': torch.from_numpy(x_low)}
    L = T * 240
    noise = None
    if mode == 'noise':
        u = np.random.RandomState(8).uniform(1e-6, 1 - 1e-6, size=(B, L, 256))
        noise = (-np.log(-np.log(u))).astype(np.float32)
    omode = {'noise': O.MODE_NOISE, 'philox': O.MODE_PHILOX, 'argmax': O.MODE_ARGMAX}[mode]
    ridx, rwav, rlog = O.decode(sd, mel, x_low, num_layers=N, H=H, mode=omode, noise=noise, seed=0xBEEF, want_logits=True)
    idx, wav, logits = net.decode(X, mode=mode, noise=noise, seed=0xBEEF, want_logits=True)
    assert net.last_kernel == 'stream'
    assert np.array_equal(idx.cpu().numpy(), ridx), 'first index mismatch at %s' % (np.argwhere(idx.cpu().numpy() != ridx)[:3],)
    assert np.array_equal(wav.cpu().numpy(), rwav)
    assert np.array_equal(logits.cpu().numpy(), rlog)


def test_ragged_tile_and_batch_independence():
    """A batch of 7 (one utterance per workgroup at this size; the multi-utterance tiles are covered by
    test_utterance_tiles_bit_exact): utterance b alone == utterance b inside the batch."""
    H, N = 64, 1
    sd = O.synthetic_state_dict(H=H, num_layers=N, use_lowres=True, seed=9)
    net = _net(H, N, True, sd)
    mel, x_low = O.synthetic_inputs(7, 1, seed=3)
    X = {'mel': torch.from_numpy(mel), 'x_low': torch.from_numpy(x_low)}
    idx, _, _ = net.decode(X, mode='philox', seed=42)
    # philox counters carry the utterance index, so compare against the oracle rather than a B=1 rerun
    ridx, _, _ = O.decode(sd, mel, x_low, num_layers=N, H=H, mode=O.MODE_PHILOX, seed=42)
    assert np.array_equal(idx.cpu().numpy(), ridx)
    a, _, _ = net.decode(X, mode='argmax')
    b, _, _ = net.decode({'mel': X['mel'][3:4], 'x_low': X['x_low'][3:4]}, mode='argmax')
    assert torch.equal(a[3:4], b)


def test_wavernn_errors():
    from ttscube_amd._lib import TTSCError
    from ttscube_amd.networks.modules import WaveRNN
    with pytest.raises(ValueError):
        WaveRNN(output='dirac')
    net = WaveRNN(num_layers=1, layer_size=64, upsample=240, output='mulaw')
    with pytest.raises(TTSCError):
        net({'mel': torch.zeros(1, 2, 80), 'x_low': torch.zeros(1, 48)})  # parameters on CPU: no CPU path
    net = net.cuda()
    with pytest.raises(TTSCError):
        net.decode({'mel': torch.zeros(1, 2, 80), 'x_low': torch.zeros(1, 48)}, mode='noise', noise=None)


def test_cubenet_vocoder_fold_and_decode(golden_dir):
    """CubenetVocoder (vocoder.py:96-131): fold/unfold equal the reference's own outputs (fixture made by import), and
    the two-network decode equals lr-oracle -> fold -> hr-oracle bit-exactly in argmax mode."""
    from ttscube_amd.networks.vocoder import CubenetVocoder
    z = np.load(os.path.join(golden_dir, 'vocoder_fold.npz'))
    voc = CubenetVocoder(num_layers_lr=1, layer_size_lr=64, num_layers_hr=1, layer_size_hr=64, upsample=240, upsample_low=10,
                         output='mulaw')
    sd_hr = O.synthetic_state_dict(H=64, num_layers=1, use_lowres=True, seed=21)
    sd_lr = O.synthetic_state_dict(H=64, num_layers=1, use_lowres=False, seed=22)
    sd = {'_wavernn_hr.' + k: torch.from_numpy(v) for k, v in sd_hr.items()}
    sd.update({'_wavernn_lr.' + k: torch.from_numpy(v) for k, v in sd_lr.items()})
    voc.load_state_dict(sd, strict=True)
    voc = voc.cuda().eval()
    f = voc._inference_batch(torch.from_numpy(z['mel']).cuda(), torch.from_numpy(z['x_low']).cuda(), num_batches=20)
    assert np.array_equal(f['mel'].cpu().numpy(), z['fold_mel']) and np.array_equal(f['x_low'].cpu().numpy(), z['fold_x_low'])
    assert np.array_equal(voc._compose_batched_inference(torch.from_numpy(z['hr']).cuda()).cpu().numpy(), z['composed'])
    mel = z['mel'][:, :40]
    x_lr, x_hr = voc({'mel': torch.from_numpy(mel)}, mode='argmax') if False else voc._inference({'mel': torch.from_numpy(mel)}, mode='argmax')
    _, r_lr, _ = O.decode(sd_lr, mel, None, num_layers=1, H=64, use_lowres=False, upsample=24, mode=O.MODE_ARGMAX)
    fo = O.inference_batch(mel, r_lr, num_batches=20)
    _, r_hr, _ = O.decode(sd_hr, fo['mel'], fo['x_low'], num_layers=1, H=64, mode=O.MODE_ARGMAX)
    assert x_lr.shape == (1, 960, 1) and x_hr.shape == (1, 6800)  # T=40 -> lr 960, hr 6800
    assert np.array_equal(x_lr[:, :, 0], r_lr)
    assert np.array_equal(x_hr, O.compose_batched_inference(r_hr))


# ---- continuous output distributions (MOL = the reference's default, modules.py:398) ---------------------------------------
CONT = ['wavernn_hr_h64_mol', 'wavernn_hr_h512_mol', 'wavernn_lr_h64_gm', 'wavernn_hr_h64_beta']


def _cont_net(z):
    out = str(z['output'])
    H, N, lowres = int(z['H']), int(z['N']), bool(z['use_lowres'])
    sd = O.synthetic_state_dict(H=H, num_layers=N, use_lowres=lowres, seed=int(z['seed']), S=O.SAMPLE_SIZE[out])
    sd['_output.linear_layer.weight'], sd['_output.linear_layer.bias'] = z['out_w'], z['out_b']
    X = {'mel': torch.from_numpy(z['mel'])}
    if lowres:
        X['x_low'] = torch.from_numpy(z['x_low'])
    kw = dict(num_layers=N, H=H, use_lowres=lowres, upsample=240 if lowres else 24, output=out)
    return _net(H, N, lowres, sd, out), sd, X, kw


@pytest.mark.parametrize('name', CONT)
def test_continuous_outputs_match_reference_and_oracle(golden_dir, name, wr_kernel):
    """mol / gm / beta in the persistent kernels (tile kernel for one- and two-layer nets, streaming kernel otherwise): with the reference's own random terms injected the samples follow the
    reference run to 1e-5 (mixture index identical at every step) and equal the C oracle BIT FOR BIT (one shared arithmetic
    definition) in noise, Philox and arg-max mode; teacher-forced outputs vs the reference's _train_forward <= 1e-4."""
    z = np.load(os.path.join(golden_dir, name + '.npz'))
    net, sd, X, kw = _cont_net(z)
    want_kernel = 'tile' if (wr_kernel == 'tile' and kw['num_layers'] <= 2) else 'stream'
    xl = z['x_low'] if kw['use_lowres'] else None
    if 'noise' in z.files:
        idx, wav, _ = net.decode(X, mode='noise', noise=z['noise'])
        assert net.last_kernel == want_kernel
        assert float(np.abs(wav.cpu().numpy() - z['wav']).max()) < 1e-5
        if kw['output'] == 'mol':
            assert np.array_equal(idx.cpu().numpy(), z['idx'])
        ridx, rwav, _ = O.decode(sd, z['mel'], xl, mode=O.MODE_NOISE, noise=z['noise'], **kw)
        assert np.array_equal(wav.cpu().numpy(), rwav) and np.array_equal(idx.cpu().numpy(), ridx)
    for mode, omode in (('philox', O.MODE_PHILOX), ('argmax', O.MODE_ARGMAX)):
        ridx, rwav, rlog = O.decode(sd, z['mel'], xl, mode=omode, seed=0xABCDEF0123, want_logits=True, **kw)
        idx, wav, logits = net.decode(X, mode=mode, seed=0xABCDEF0123, want_logits=True)
        assert np.array_equal(wav.cpu().numpy(), rwav), (mode, np.argwhere(wav.cpu().numpy() != rwav)[:3])
        assert np.array_equal(idx.cpu().numpy(), ridx) and np.array_equal(logits.cpu().numpy(), rlog)
    xin = np.concatenate([np.zeros_like(z['audio'][:, :1]), z['audio'][:, :-1]], axis=1)
    Xt = dict(X)
    Xt['x'] = torch.from_numpy(xin).cuda()
    lg = net(Xt)
    assert float(np.abs(lg.cpu().numpy() - z['logits_tf']).max()) < 1e-4
    # the loss the training step minimises, evaluated by the mirrored output class on the kernel's outputs (loss.py)
    loss = float(net._output_functions.loss(lg.cpu(), torch.from_numpy(z['audio'])))
    assert abs(loss - float(z['loss_tf'])) < 2e-3 * max(1.0, abs(float(z['loss_tf'])))


def test_reference_default_constructor_decodes():
    """`WaveRNN()` with the reference's default arguments (2 x 512, upsample 100, output='mol', modules.py:391-398) and
    `CubenetVocoder(...)` with its default output construct, decode in-kernel-RNG mode and stay inside [-1, 1]."""
    from ttscube_amd.networks.modules import WaveRNN
    from ttscube_amd.networks.vocoder import CubenetVocoder
    torch.manual_seed(0)
    net = WaveRNN().cuda().eval()
    assert net._output_functions.sample_size == 30
    mel = torch.randn(3, 2, 80)
    y = net({'mel': mel, 'x_low': torch.rand(3, 20) * 2 - 1})
    assert y.shape == (3, 200, 1) and np.isfinite(y).all() and np.abs(y).max() <= 1.0
    voc = CubenetVocoder(num_layers_lr=1, layer_size_lr=64, num_layers_hr=1, layer_size_hr=64, upsample=240, upsample_low=10).cuda().eval()
    x_lr, x_hr = voc({'mel': torch.randn(1, 40, 80)})
    assert x_lr.shape == (1, 960, 1) and x_hr.shape == (1, 6800) and np.isfinite(x_hr).all()


def test_default_kernel_choice_by_batch(monkeypatch):
    """no env override: the tile kernel takes one-layer networks while every member can be resident (B <= 256),
    a partial last tile included; the streaming kernel takes the rest"""
    monkeypatch.delenv('TTSC_WR_TILE', raising=False)
    sd = O.synthetic_state_dict(H=64, num_layers=1, use_lowres=True, seed=77)
    net = _net(64, 1, True, sd)
    for B, want in ((3, 'tile'), (200, 'tile'), (300, 'stream')):
        mel, x_low = O.synthetic_inputs(B, 1, seed=B)
        idx, _, _ = net.decode({'mel': torch.from_numpy(mel), 'x_low': torch.from_numpy(x_low)}, mode='philox', seed=9)
        assert net.last_kernel == want
        ridx, _, _ = O.decode(sd, mel, x_low, num_layers=1, H=64, mode=O.MODE_PHILOX, seed=9)
        assert np.array_equal(idx.cpu().numpy(), ridx)
    # two-layer networks (the reference class default) run on the tile kernel too; TTSC_WR_TILE2=0 is the A/B switch back
    sd2 = O.synthetic_state_dict(H=128, num_layers=2, use_lowres=True, seed=78)
    net2 = _net(128, 2, True, sd2)
    mel, x_low = O.synthetic_inputs(11, 1, seed=5)
    X = {'mel': torch.from_numpy(mel), 'x_low': torch.from_numpy(x_low)}
    ridx, _, rlog = O.decode(sd2, mel, x_low, num_layers=2, H=128, mode=O.MODE_PHILOX, seed=3, want_logits=True)
    idx, _, logits = net2.decode(X, mode='philox', seed=3, want_logits=True)
    assert net2.last_kernel == 'tile' and np.array_equal(idx.cpu().numpy(), ridx) and np.array_equal(logits.cpu().numpy(), rlog)
    monkeypatch.setenv('TTSC_WR_TILE2', '0')
    idx, _, _ = net2.decode(X, mode='philox', seed=3)
    assert net2.last_kernel == 'stream' and np.array_equal(idx.cpu().numpy(), ridx)
