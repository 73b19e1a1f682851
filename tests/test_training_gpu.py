"""GPU: training steps — the differentiable forwards agree with the HIP inference kernels, one
Cubegan GAN step and one CubenetVocoder step run, update every parameter group and write the reference checkpoint layout."""
import numpy as np
import pytest
import torch

from oracle import hifigan_ref as R
from oracle import meldecoder_ref as M
from oracle import wavernn_ref as O

pytestmark = pytest.mark.gpu


class _Enc:
    phon2int = {str(i): i for i in range(20)}
    speaker2int = {'a': 0, 'b': 1}
    max_pitch = 300
    max_duration = 10


def _batch(B, nph, rng):
    from ttscube_amd.io_utils.io_cubegan import CubeganCollate, CubeganEncodings
    enc = CubeganEncodings()
    enc.phon2int, enc.speaker2int, enc.max_pitch, enc.max_duration = _Enc.phon2int, _Enc.speaker2int, _Enc.max_pitch, _Enc.max_duration
    ex = []
    for b in range(B):
        durs = rng.randint(3, 9, size=nph)
        f2p = [p for p, d in enumerate(durs) for _ in range(d)]
        F_ = len(f2p)
        ex.append({'meta': {'phones': [str(v) for v in rng.randint(0, 20, size=nph)], 'speaker': 'a' if b % 2 else 'b',
                            'frame2phon': f2p, 'phon2word': [0] * nph},
                   'mgc': np.clip(rng.randn(F_, 80) - 2, -5, 1), 'pitch': rng.randint(0, 300, size=F_).astype(np.float64),
                   'audio': rng.uniform(-0.5, 0.5, size=F_ * 240)})
    return CubeganCollate(enc).collate_fn(ex), enc


def test_training_forward_matches_hip_inference_kernels():
    from ttscube_amd.hifigan.env import AttrDict
    from ttscube_amd.hifigan.models import Generator
    from tests.torch_reference import generator_forward_train
    h = dict(R.CONFIG_V1, upsample_initial_channel=128)
    g = Generator(AttrDict(h))
    g.load_state_dict(R.synthetic_state_dict(h, seed=2))
    g = g.cuda()
    mel = R.synthetic_mel(2, 11, seed=3).cuda()
    with torch.no_grad():
        y_hip = g(mel)
    y_tr = g(mel.requires_grad_(True))            # grad mode -> differentiable path
    assert y_tr.requires_grad
    assert float((y_hip - y_tr.detach()).pow(2).mean().sqrt()) < 1e-4
    y_tr.pow(2).mean().backward()
    assert g.conv_pre.weight_g.grad is not None and g.resblocks[5].convs2[1].weight_v.grad is not None


def _rel(a, b):
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.mark.parametrize('cfg', [
    dict(Cin=16, Cout=32, K=7, padding=3, dilation=1),
    dict(Cin=32, Cout=32, K=11, padding=25, dilation=5),
    dict(Cin=48, Cout=20, K=3, padding=3, dilation=3),
    dict(Cin=32, Cout=1, K=7, padding=3, dilation=1),
    dict(Cin=5, Cout=70, K=1, padding=0, dilation=1),
    dict(Cin=64, Cout=32, K=16, stride=5, padding=5, transposed=True),
    dict(Cin=32, Cout=16, K=16, stride=3, padding=6, transposed=True),
    dict(Cin=16, Cout=8, K=4, stride=4, padding=0, transposed=True),
    dict(Cin=8, Cout=12, K=5, stride=2, padding=1, transposed=True),
])
def test_hip_conv_autograd_matches_torch(cfg):
    """forward / dgrad (gated by the fused leaky-relu) / wgrad / bias grad / residual grad of one layer vs torch autograd."""
    import torch.nn.functional as F
    from ttscube_amd.hifigan.autograd import TrainConv, hip_conv
    g = torch.Generator().manual_seed(5)
    tr = cfg.get('transposed', False)
    Cin, Cout, K = cfg['Cin'], cfg['Cout'], cfg['K']
    tc = TrainConv(**cfg)
    B, L = 3, 157
    x = torch.randn(B, Cin, L, generator=g).cuda().requires_grad_(True)
    w = (torch.randn((Cin, Cout, K) if tr else (Cout, Cin, K), generator=g) / (Cin * K) ** 0.5).cuda().requires_grad_(True)
    b = torch.randn(Cout, generator=g).cuda().requires_grad_(True)
    sc, sl = 1.0 / 3, 0.1

    def ref(x, w, b, r):
        a = F.leaky_relu(x * sc, sl)
        if tr:
            y = F.conv_transpose1d(a, w, b, stride=cfg['stride'], padding=cfg['padding'])
        else:
            y = F.conv1d(a, w, b, dilation=cfg['dilation'], padding=cfg['padding'])
        return y + r if r is not None else y

    Lo = ref(x, w, b, None).shape[2]
    r = torch.randn(B, Cout, Lo, generator=g).cuda().requires_grad_(True)
    gy = torch.randn(B, Cout, Lo, generator=g).cuda()
    y0 = ref(x, w, b, r)
    g0 = torch.autograd.grad(y0, (x, w, b, r), gy)
    for rep in range(2):      # second pass re-uses the handles / index maps
        y1 = hip_conv(tc, x, w, b, resid=r, in_scale=sc, in_slope=sl)
        g1 = torch.autograd.grad(y1, (x, w, b, r), gy)
        assert _rel(y1, y0) < 2e-6
        for a_, b_ in zip(g1, g0):
            assert a_.shape == b_.shape and _rel(a_, b_) < 5e-6
    # no input gradient requested (conv_pre on a detached mel) and no residual
    y2 = hip_conv(tc, x.detach(), w, b, in_scale=sc, in_slope=sl)
    gw, = torch.autograd.grad(y2, (w,), gy)
    assert _rel(gw, g0[1]) < 5e-6


def test_native_generator_gradients_match_torch_autograd():
    from ttscube_amd.hifigan.env import AttrDict
    from ttscube_amd.hifigan.models import Generator
    from tests.torch_reference import generator_forward_train
    from ttscube_amd.hifigan.autograd import generator_forward_with_grad
    h = dict(R.CONFIG_V1, upsample_initial_channel=128)
    g = Generator(AttrDict(h))
    g.load_state_dict(R.synthetic_state_dict(h, seed=4))
    g = g.cuda()
    mel = R.synthetic_mel(3, 9, seed=6).cuda().requires_grad_(True)
    tgt = torch.randn(3, 1, 9 * 240 + 64, generator=torch.Generator().manual_seed(1)).cuda() * 0.3
    params = [p for p in g.parameters() if p.requires_grad]
    y0 = generator_forward_train(g, mel)
    g0 = torch.autograd.grad((y0 - tgt).abs().mean(), [mel] + params)
    y1 = generator_forward_with_grad(g, mel)
    g1 = torch.autograd.grad((y1 - tgt).abs().mean(), [mel] + params)
    assert _rel(y1, y0) < 1e-5
    worst = max(_rel(a, b) for a, b in zip(g1, g0))
    assert worst < 1e-4, worst


def test_cubegan_training_step_runs_and_updates_all_groups(tmp_path):
    from ttscube_amd.networks.cubegan import Cubegan
    from ttscube_amd.networks import training as T
    rng = np.random.RandomState(0)
    batch, enc = _batch(2, 12, rng)
    torch.manual_seed(0)
    model = Cubegan(enc, conditioning=None, train=True).cuda()
    model.train()
    opts = T.cubegan_configure_optimizers(model)
    g, d, t = T.cubegan_param_groups(model)
    before = [p.detach().clone() for p in (g[0], d[0], t[0])]
    import random
    out = T.cubegan_training_step(model, batch, opts, rng=random.Random(1))
    assert all(np.isfinite(v) for v in out.values())
    assert model._global_step == 1 and abs(out['lr'] - 2e-4 / (1 + 1e-5)) < 1e-12
    for p0, p in zip(before, (g[0], d[0], t[0])):
        assert not torch.equal(p0, p.detach())
    # checkpoint layout of train_cubegan.py:38-66
    base = str(tmp_path / 'cubegan')
    model.save(base + '.last')
    torch.save({**{str(i): o.state_dict() for i, o in enumerate(opts)}, 'global_step': model._global_step}, base + '.opt.last')
    sd = torch.load(base + '.last', map_location='cpu')
    assert {k.split('.')[0] for k in sd} == {'_generator', '_mpd', '_msd', '_languasito', '_dummy'}
    m2 = Cubegan(enc, conditioning=None, train=True).cuda()   # (the optimizers refuse CPU parameters: no CPU path)
    m2.load(base + '.last')
    m2._loaded_optimizer_states = torch.load(base + '.opt.last', map_location='cpu')
    o2 = T.cubegan_configure_optimizers(m2)
    assert o2[0].state_dict()['state'] and m2._loaded_optimizer_states is None   # optimizer state actually restored
    # the trained weights drive the HIP inference path straight away (same parameter tensors)
    model.eval()
    wav = model.inference({'x_char': batch['x_char'][:1, :12], 'x_speaker': batch['x_speaker'][:1]})
    assert wav.dim() == 3 and bool(torch.isfinite(wav).all())


def test_reference_shaped_step_body_runs_the_same_kernels_as_the_step_function():
    """The drop-in boundary of row a9: a training step written the way the reference writes it — `self._languasito(batch)`,
    `self._generator(cond)`, `self._mpd(y, y_g_hat.detach())`, `discriminator_loss(...)`, `mel_spectrogram(...)`, `feature_loss`, `generator_loss`,
    `self.optimizers()`, cube/networks/cubegan.py:93,131-180, with the names imported from hifigan/discriminators.py as cubegan.py:18-21 imports them
    from hifigan.models / hifigan.meldataset — must land on the HIP kernels by itself: its losses are bit-identical to
    `Cubegan.training_step` (= networks/training.py::cubegan_training_step) over two steps (the second step's losses see the first one's updates of all
    three parameter groups), and a CPU tensor anywhere raises instead of falling back to torch ops."""
    import random
    import torch.nn.functional as F
    from ttscube_amd.hifigan.discriminators import discriminator_loss, feature_loss, generator_loss, mel_spectrogram
    from ttscube_amd.networks.cubegan import Cubegan
    batch, enc = _batch(2, 12, np.random.RandomState(3))
    torch.manual_seed(0)
    a = Cubegan(enc, conditioning=None, train=True).cuda().train()
    b = Cubegan(enc, conditioning=None, train=True).cuda().train()
    b.load_state_dict(a.state_dict())

    def reference_shaped_step(self, batch, rng):
        opt_g, opt_d, opt_t, opt_b = self.optimizers()
        dev = self.get_device()
        p_dur, p_pitch, p_vuv, conditioning = self._languasito(batch)
        t_dur = batch['y_dur'].to(dev)
        t_pitch = batch['y_pitch'].to(dev)
        t_vuv = (t_pitch > 1).float()
        m_size = min(t_dur.shape[1], p_dur.shape[1])
        t_dur, p_dur = t_dur[:, :m_size], p_dur[:, :m_size, :]
        m_size = min(t_pitch.shape[1], p_pitch.shape[1])
        t_pitch, p_pitch, t_vuv, p_vuv = t_pitch[:, :m_size], p_pitch[:, :m_size], t_vuv[:, :m_size], p_vuv[:, :m_size]
        ignore = int(max(self._encodings.max_pitch, self._encodings.max_duration) + 1)
        loss_duration = F.cross_entropy(p_dur.reshape(-1, p_dur.shape[2]), t_dur.reshape(-1), ignore_index=ignore)
        loss_pitch = (torch.abs(t_pitch / self._languasito._max_pitch - p_pitch) * t_vuv).mean() + torch.abs(t_vuv - p_vuv).mean()
        y = batch['y_audio'].to(dev)
        if y.shape[1] > 12000 - 240:
            y_t_list, c_list = [], []
            for ii in range(y.shape[0]):
                max_frame = len(batch['y_frame2phone'][ii])
                r = rng.randint(0, max_frame - 50 - 1) if max_frame > 51 else 0
                c_list.append(conditioning[ii, r:r + 50, :].unsqueeze(0))
                y_t_list.append(y[ii, r * 240:r * 240 + 12000].unsqueeze(0))
            y = torch.cat(y_t_list, dim=0)
            conditioning = torch.cat(c_list, dim=0)
        y = y.unsqueeze(1)
        y_g_hat = self._generator(conditioning.permute(0, 2, 1))
        m_size = min(y.shape[2], y_g_hat.shape[2])
        y, y_g_hat = y[:, :, :m_size], y_g_hat[:, :, :m_size]
        y_mel = mel_spectrogram(y.squeeze(1), 1024, 80, 24000, 240, 1024, 0, 12000)
        y_g_hat_mel = mel_spectrogram(y_g_hat.squeeze(1), 1024, 80, 24000, 240, 1024, 0, 12000)
        opt_b.zero_grad()
        opt_d.zero_grad()
        y_df_hat_r, y_df_hat_g, _, _ = self._mpd(y, y_g_hat.detach())
        loss_disc_f, _, _ = discriminator_loss(y_df_hat_r, y_df_hat_g)
        y_ds_hat_r, y_ds_hat_g, _, _ = self._msd(y, y_g_hat.detach())
        loss_disc_s, _, _ = discriminator_loss(y_ds_hat_r, y_ds_hat_g)
        loss_disc_all = loss_disc_s + loss_disc_f
        loss_disc_all.backward()
        opt_d.step()
        opt_g.zero_grad()
        loss_mel = self._loss_l1(y_mel, y_g_hat_mel) * 45
        y_df_hat_r, y_df_hat_g, fmap_f_r, fmap_f_g = self._mpd(y, y_g_hat)
        y_ds_hat_r, y_ds_hat_g, fmap_s_r, fmap_s_g = self._msd(y, y_g_hat)
        loss_fm_f = feature_loss(fmap_f_r, fmap_f_g)
        loss_fm_s = feature_loss(fmap_s_r, fmap_s_g)
        loss_gen_f, _ = generator_loss(y_df_hat_g)
        loss_gen_s, _ = generator_loss(y_ds_hat_g)
        loss_gen_all = loss_gen_s + loss_gen_f + loss_fm_s + loss_fm_f + loss_mel
        loss_gen_all.backward(retain_graph=True)
        opt_g.step()
        opt_t.zero_grad()
        loss_text = loss_pitch + loss_duration
        loss_text.backward(retain_graph=True)
        opt_t.step()
        opt_b.step()
        self._global_step += 1
        self._current_lr = self._compute_lr(self._learning_rate, 1e-5, self._global_step)
        for o in (opt_d, opt_g, opt_t):
            o.param_groups[0]['lr'] = self._current_lr
        return {'loss_g': float(loss_gen_all), 'loss_t': float(loss_text), 'loss_d': float(loss_disc_all)}

    ra, rb = random.Random(5), random.Random(5)
    for step in range(2):
        oa = a.training_step(batch, step, rng=ra)
        ob = reference_shaped_step(b, batch, rb)
        for k in ('loss_d', 'loss_g', 'loss_t'):
            assert oa[k] == ob[k], (step, k, oa[k], ob[k])
        assert oa['loss'] == oa['loss_g'] + oa['loss_d'] + oa['loss_t']
    # same parameters after two steps, bit for bit, in all three groups
    from ttscube_amd.networks import training as T
    for ga, gb in zip(T.cubegan_param_groups(a), T.cubegan_param_groups(b)):
        assert all(torch.equal(p.detach(), q.detach()) for p, q in zip(ga, gb))
    from ttscube_amd._lib import TTSCError
    with pytest.raises(TTSCError):
        a._mpd(torch.zeros(1, 1, 600), torch.zeros(1, 1, 600))
    with pytest.raises(TTSCError):
        discriminator_loss([torch.zeros(2, 3)], [torch.zeros(2, 3)])
    # validation through the class surface (cubegan.py:191-273): finite mel-L1, _val_loss from validation_epoch_end
    a.eval()
    v = a.validation_step(batch, 0, rng=random.Random(1))
    a.validation_epoch_end([v])
    assert np.isfinite(v['loss_mel']) and a._val_loss == v['loss_mel']


def test_vocoder_training_step_and_teacher_forced_logits_agree():
    from ttscube_amd.networks.vocoder import CubenetVocoder
    from ttscube_amd.networks import training as T
    torch.manual_seed(0)
    voc = CubenetVocoder(num_layers_lr=1, layer_size_lr=64, num_layers_hr=1, layer_size_hr=64, upsample=240, upsample_low=10,
                         learning_rate=1e-3, output='mulaw').cuda()
    rng = np.random.RandomState(0)
    B, T_ = 2, 3
    x = torch.from_numpy(rng.uniform(-0.9, 0.9, size=(B, T_ * 240)).astype(np.float32))
    batch = {'x': x, 'x_low': x[:, ::10].contiguous(), 'mel': torch.from_numpy(np.clip(rng.randn(B, T_ + 1, 80) - 2, -5, 1).astype(np.float32))}
    # differentiable teacher-forced logits == the decode kernel's forced-feedback logits (both are _train_forward)
    net = voc._wavernn_hr
    xin = torch.nn.functional.pad(x[:, :-1], (1, 0)).cuda()
    X = {'x': xin, 'x_low': batch['x_low'].cuda(), 'mel': batch['mel'].cuda()}
    with torch.no_grad():
        lt = T.wavernn_logits_train(net, X)
    lk = net(X)
    assert lt.shape == lk.shape and float((lt - lk).abs().max()) < 1e-4
    opts = (torch.optim.Adam(voc._wavernn_lr.parameters(), lr=1e-3), torch.optim.Adam(voc._wavernn_hr.parameters(), lr=1e-3))
    l0 = T.vocoder_training_step(voc, batch, opts)
    for _ in range(5):
        l1 = T.vocoder_training_step(voc, batch, opts)
    assert np.isfinite(l1['loss']) and l1['loss'] < l0['loss']          # same batch: the loss goes down
    assert abs(l1['alpha'] - 1e-3 / (1 + 5e-5 * 6)) < 1e-12
    # validation through the HIP kernel (CubenetVocoder.forward with 'x' in X -> losses)
    v = voc({k: t.cuda() for k, t in batch.items()})
    assert abs(float(v['hr']) - T.wavernn_train_loss(net, {k: t.cuda() for k, t in batch.items()}).item()) < 1e-3
    sd = voc.state_dict()
    assert any(k.startswith('_wavernn_hr._rnns.0.') for k in sd) and any(k.startswith('_wavernn_lr._skip.') for k in sd)


def test_fused_weight_norm_and_bias_grad_match_torch():
    from ttscube_amd.hifigan.autograd import HipWeightNormFn, TrainConv, _bias_grad
    g_ = torch.Generator().manual_seed(9)
    for shape in [(512, 256, 16), (32, 32, 11), (1, 32, 7), (7, 3, 1)]:
        v = torch.randn(shape, generator=g_).cuda().requires_grad_(True)
        g = (torch.rand(shape[0], 1, 1, generator=g_) + 0.5).cuda().requires_grad_(True)
        gw = torch.randn(shape, generator=g_).cuda()
        w0 = g * v / v.reshape(shape[0], -1).norm(dim=1).reshape(-1, 1, 1)
        r0 = torch.autograd.grad(w0, (v, g), gw)
        w1 = HipWeightNormFn.apply(v, g)
        r1 = torch.autograd.grad(w1, (v, g), gw)
        assert _rel(w1, w0) < 1e-6 and _rel(r1[0], r0[0]) < 1e-5 and _rel(r1[1], r0[1]) < 1e-5
    tc = TrainConv(4, 4, 1)
    for B, C_, L in [(16, 32, 12064), (3, 5, 17), (16, 512, 50), (2, 1, 100000)]:
        dy = torch.randn(B, C_, L, generator=g_).cuda()
        for rep in range(2):          # second call re-uses the ticket workspace
            db = _bias_grad(tc, dy)
            assert _rel(db, dy.double().sum(dim=(0, 2)).float()) < 1e-5


@pytest.mark.parametrize('cfg', [(2, True, 64, 37, 3, 9), (1, False, 32, 20, 2, 5), (2, True, 256, 641, 2, 30), (2, True, 256, 80, 16, 24),
                                 (1, False, 512, 96, 3, 17), (1, True, 128, 40, 200, 4)])
def test_hip_lstm_autograd_matches_torch(cfg):
    """forward + backward-through-time kernels of a stacked (bi)LSTM vs torch.nn.LSTM autograd (same parameters): the
    single-workgroup kernels (H = 64 / 32, or too many sequences for the CUs) and the split kernels (2..4 workgroups per
    (utterance, direction) exchanging the state every step)."""
    from ttscube_amd.networks.lstm_autograd import lstm_forward_train
    from ttscube_amd import _lib
    layers, bi, H, nin, B, T = cfg
    torch.manual_seed(3)
    m = torch.nn.LSTM(nin, H, num_layers=layers, bidirectional=bi, batch_first=True).cuda()
    x = torch.randn(B, T, nin, device='cuda', requires_grad=True)
    gy = torch.randn(B, T, H * (2 if bi else 1), device='cuda')
    params = list(m.parameters())
    y0, _ = m(x)
    g0 = torch.autograd.grad(y0, [x] + params, gy)
    y1 = lstm_forward_train(m, x)
    assert _lib.lib().ttsc_lstm_split_status() == 0
    g1 = torch.autograd.grad(y1, [x] + params, gy)
    assert _lib.lib().ttsc_lstm_split_status() == 0
    assert _rel(y1, y0) < 1e-5
    for a_, b_, p in zip(g1, g0, [x] + params):
        assert a_.shape == b_.shape and _rel(a_, b_) < 1e-4, (tuple(p.shape), _rel(a_, b_))


@pytest.mark.parametrize('cfg', [(64, 102, 3, 50), (512, 81, 2, 33), (128, 7, 1, 300), (512, 102, 16, 40), (256, 20, 300, 5)])
def test_hip_gru_autograd_matches_torch(cfg):
    """forward + backward-through-time GRU kernels vs torch.nn.GRU autograd (same parameters): single-workgroup kernels
    (H=64, or B=300 > CUs / 2) and the split kernels (2..8 workgroups per utterance exchanging the state every step)."""
    from ttscube_amd.networks.gru_autograd import gru_forward_train
    from ttscube_amd import _lib
    H, nin, B, T = cfg
    torch.manual_seed(4)
    m = torch.nn.GRU(nin, H, num_layers=1, batch_first=True).cuda()
    x = torch.randn(B, T, nin, device='cuda', requires_grad=True)
    gy = torch.randn(B, T, H, device='cuda')
    params = list(m.parameters())
    y0, _ = m(x)
    g0 = torch.autograd.grad(y0, [x] + params, gy)
    y1 = gru_forward_train(m, x)
    assert _lib.lib().ttsc_gru_split_status() == 0
    g1 = torch.autograd.grad(y1, [x] + params, gy)
    assert _lib.lib().ttsc_gru_split_status() == 0
    assert _rel(y1, y0) < 1e-5
    for a_, b_, p in zip(g1, g0, [x] + params):
        assert a_.shape == b_.shape and _rel(a_, b_) < 1e-4, (tuple(p.shape), _rel(a_, b_))


def test_flat_adamw_matches_torch_adamw_and_speaks_its_state_dict():
    """ttsc_adamw_step over flat arenas == torch.optim.AdamW(betas=(0.8, 0.99)) step for step (cubegan.py:275-298), parameters and
    gradients alias the arenas, version counters move (weight caches of the inference handles see the update), and the state_dict
    round-trips through torch's optimizer."""
    from ttscube_amd.optim import FlatAdamW
    torch.manual_seed(3)
    mk = lambda: torch.nn.Sequential(torch.nn.Linear(37, 129), torch.nn.Tanh(), torch.nn.Linear(129, 5)).cuda()
    a, b = mk(), mk()
    b.load_state_dict(a.state_dict())
    extra = torch.nn.Parameter(torch.zeros(3, device='cuda'))     # never differentiated: must stay untouched, as torch leaves it
    oa = FlatAdamW(list(a.parameters()) + [extra], lr=2e-4, betas=(0.8, 0.99))
    ob = torch.optim.AdamW(b.parameters(), lr=2e-4, betas=(0.8, 0.99))
    x = torch.randn(16, 37, device='cuda')
    for step in range(5):
        for net, opt in ((a, oa), (b, ob)):
            opt.zero_grad()
            (net(x).tanh().pow(2).mean() * (1 + step)).backward()
            opt.step()
            opt.param_groups[0]['lr'] = 2e-4 / (1 + 1e-5 * (step + 1))
        for p, q in zip(a.parameters(), b.parameters()):
            assert float((p - q).abs().max()) < 2e-7, step
    assert extra.grad is None and float(extra.abs().sum()) == 0
    v0 = [p._version for p in a.parameters()]
    oa.zero_grad(); a(x).sum().backward(); oa.step()
    assert all(p._version > v for p, v in zip(a.parameters(), v0))
    # state_dict: torch layout -> torch optimizer -> back
    sd = oa.state_dict()
    assert set(sd['state'].keys()) == {0, 1, 2, 3} and float(sd['state'][0]['step']) == 6.0
    oc = torch.optim.AdamW(a.parameters(), lr=1.0)
    oc.load_state_dict({'state': sd['state'], 'param_groups': [dict(sd['param_groups'][0], params=[0, 1, 2, 3])]})
    assert torch.equal(oc.state_dict()['state'][2]['exp_avg'], sd['state'][2]['exp_avg'])
    c = mk()
    c.load_state_dict(a.state_dict())
    od = FlatAdamW(list(c.parameters()), lr=1.0)
    od.load_state_dict(sd)                                       # before the arenas exist: applied at the first step
    for net, opt in ((a, oa), (c, od)):
        opt.zero_grad(); net(x).pow(2).mean().backward(); opt.step()
    for p, q in zip(a.parameters(), c.parameters()):
        assert torch.equal(p, q)
    with pytest.raises(Exception):
        FlatAdamW([torch.nn.Parameter(torch.zeros(4))]).step()   # CPU parameters: no CPU path


def test_fused_gan_losses_match_torch_formulations():
    from tests import torch_reference as D                # torch-op formulations (test infrastructure)
    from ttscube_amd.hifigan import discriminators as H   # the drop-in names: gan_loss_kernel underneath
    g = torch.Generator().manual_seed(5)
    shapes = [(4, 32, 700, 2), (4, 128, 234, 2), (4, 1, 77, 3), (4, 1024, 9, 5), (4, 16, 12000)]
    mk = lambda: [[torch.randn(s, generator=g).cuda().requires_grad_(True) for s in shapes[:3]], [torch.randn(s, generator=g).cuda().requires_grad_(True) for s in shapes[3:]]]
    fr, fg = mk(), mk()
    fr2 = [[t.detach() for t in sub] for sub in fr]               # generator step: the real branch carries no graph
    flat = lambda ll: [t for sub in ll for t in sub]
    l0 = D.feature_loss(fr2, fg)
    g0 = torch.autograd.grad(l0 * 1.7, flat(fg))
    l1 = H.feature_loss(fr2, fg)
    g1 = torch.autograd.grad(l1 * 1.7, flat(fg))
    assert abs(float(l0) - float(l1)) < 1e-5 * abs(float(l0))
    for u, v in zip(g0, g1):
        assert float((u - v).abs().max()) <= 1e-6 * float(u.abs().max()) + 1e-12
    outs_r, outs_g = flat(mk()), flat(mk())
    d0 = D.discriminator_loss(outs_r, outs_g)[0]
    d1 = H.discriminator_loss(outs_r, outs_g)[0]
    assert abs(float(d0) - float(d1)) < 1e-5 * abs(float(d0))
    for u, v in zip(torch.autograd.grad(d0, outs_r + outs_g), torch.autograd.grad(d1, outs_r + outs_g)):
        assert float((u - v).abs().max()) <= 1e-6 * float(u.abs().max()) + 1e-12
    n0, n1 = D.generator_loss(outs_g)[0], H.generator_loss(outs_g)[0]
    assert abs(float(n0) - float(n1)) < 1e-5 * abs(float(n0))
    for u, v in zip(torch.autograd.grad(n0, outs_g), torch.autograd.grad(n1, outs_g)):
        assert float((u - v).abs().max()) <= 1e-6 * float(u.abs().max()) + 1e-12
    # slices of a batched tensor (the discriminator step runs real + generated as one batch): views must work
    big = torch.randn(8, 64, 100, generator=g).cuda().requires_grad_(True)
    s0 = D.discriminator_loss([big[:4]], [big[4:]])[0]
    s1 = H.discriminator_loss([big[:4]], [big[4:]])[0]
    ga, gb = torch.autograd.grad(s0, big)[0], torch.autograd.grad(s1, big)[0]
    assert abs(float(s0) - float(s1)) < 1e-5 * abs(float(s0)) and float((ga - gb).abs().max()) < 1e-9


def test_text_stack_training_ops_match_torch():
    """networks/text_autograd.py: embedding gather / ordered scatter-add, the three GEMMs of a Linear and the char-CNN on the conv
    kernels — values and gradients against the torch ops they replace in languasito_forward_train (modules.py:916-999)."""
    import torch.nn.functional as F
    from ttscube_amd.networks.modules import Languasito2
    from ttscube_amd.networks.text_autograd import char_cnn_train, hip_embedding, hip_linear
    g = torch.Generator().manual_seed(12)
    emb = torch.nn.Embedding(41, 64, padding_idx=0).cuda()
    idx = torch.randint(0, 41, (5, 37), generator=g).cuda()
    r0, r1 = emb(idx), hip_embedding(emb, idx)
    assert torch.equal(r0, r1)
    gy = torch.randn(r0.shape, generator=g).cuda()
    g0, = torch.autograd.grad(r0, emb.weight, gy)
    g1, = torch.autograd.grad(r1, emb.weight, gy)
    assert _rel(g1, g0) < 1e-6 and float(g1[0].abs().sum()) == 0          # padding row: no gradient
    x = torch.randn(7, 53, 512, generator=g).cuda().requires_grad_(True)
    w = (torch.randn(13, 512, generator=g) / 22).cuda().requires_grad_(True)
    b = torch.randn(13, generator=g).cuda().requires_grad_(True)
    y0, y1 = F.linear(x, w, b), hip_linear(x, w, b)
    assert y1.shape == y0.shape and _rel(y1, y0) < 2e-6
    gy = torch.randn(y0.shape, generator=g).cuda()
    for u, v in zip(torch.autograd.grad(y1, (x, w, b), gy), torch.autograd.grad(y0, (x, w, b), gy)):
        assert u.shape == v.shape and _rel(u, v) < 5e-6
    torch.manual_seed(3)
    lang = Languasito2(40, 2, 300, 12).cuda()
    h = torch.randn(3, 64, 29, generator=g).cuda().requires_grad_(True)
    ref = h
    for layer in lang._char_cnn_t:
        if hasattr(layer, 'conv'):
            ref = torch.tanh(F.conv1d(ref, layer.conv.weight, layer.conv.bias, padding=1))
    out = char_cnn_train(lang, '_char_cnn_t', h)
    assert _rel(out, ref) < 2e-6
    ps = [h] + [p for p in lang._char_cnn_t.parameters()]
    gy = torch.randn(ref.shape, generator=g).cuda()
    for u, v in zip(torch.autograd.grad(out, ps, gy), torch.autograd.grad(ref, ps, gy)):
        assert _rel(u, v) < 1e-5


def test_cubegan_step_is_reproducible_with_eight_hardware_queues():
    """Round 4: with the runtime's default four hardware queues several of the step's streams share a queue and serialise by accident; with
    eight, every stream really runs beside the others — which exposed two latent races (inputs of side-stream jobs given back to the allocator
    while a side stream still read them; one gradient tensor shared by the three ResBlock branches and accumulated into in place by the last of
    them).  The queue count is read when the HIP runtime starts, so this runs tools/probes/train_determinism_poisoned.py in a process of its own:
    three 3-step runs from identical weights, fresh allocations poisoned with NaN / large values, must give identical parameters."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GPU_MAX_HW_QUEUES='8', PROBE_STEPS='3', PROBE_PRIO='1')
    out = subprocess.run([sys.executable, os.path.join(root, 'tools', 'probes', 'train_determinism_poisoned.py')], env=env, capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('run ')]
    assert len(lines) == 2 and all(' 0 of ' in l for l in lines), out.stdout[-2000:]


def test_guarded_adamw_skips_itself_on_a_set_word_and_matches_the_plain_launch_otherwise():
    """ttsc_adamw_step_guarded: a non-zero device word leaves parameters and moments untouched; a zero word gives ttsc_adamw_step's bits"""
    from ttscube_amd.optim import FlatAdamW
    g = torch.Generator().manual_seed(3)
    def make():
        ps = [torch.nn.Parameter(torch.randn(n, generator=g).cuda()) for n in (1000, 37, 4096)]
        for p in ps:
            p.grad = torch.randn(p.shape, generator=g).cuda()
        return ps
    g.manual_seed(3)
    pa = make()
    g.manual_seed(3)
    pb = make()
    oa, ob = FlatAdamW(pa, 1e-3, betas=(0.8, 0.99)), FlatAdamW(pb, 1e-3, betas=(0.8, 0.99))
    word = torch.zeros(1, dtype=torch.int32, device='cuda')
    oa.step()
    ob.step(guard=word)
    for a, b in zip(pa, pb):
        assert torch.equal(a.detach(), b.detach())
    word.fill_(4)
    before = [p.detach().clone() for p in pb]
    m0, v0 = ob.m.clone(), ob.v.clone()
    ob.step(guard=word)
    torch.cuda.synchronize()
    for p0, p in zip(before, pb):
        assert torch.equal(p0, p.detach())
    assert torch.equal(m0, ob.m) and torch.equal(v0, ob.v)


def test_split_status_collect_is_a_launch_not_a_wait():
    """ttsc_split_status_collect: after recurrences ran on a stream the call looks at their sticky words with one launch and leaves 0 in the
    destination when every hand-off completed; a null destination is refused"""
    from ttscube_amd import _lib
    from ttscube_amd.networks.lstm_autograd import lstm_forward_train
    L = _lib.lib()
    assert L.ttsc_split_status_collect(_lib.current_stream(), None, 0) < 0
    lstm = torch.nn.LSTM(64, 256, num_layers=1, bidirectional=True, batch_first=True).cuda()
    x = torch.randn(4, 50, 64, device='cuda', requires_grad=True)
    lstm_forward_train(lstm, x).sum().backward()
    word = torch.zeros(1, dtype=torch.int32, device='cuda')
    n = L.ttsc_split_status_collect(_lib.current_stream(), word.data_ptr(), 1)
    assert n >= 0
    assert int(word.item()) == 0
    _lib.check_split_status('test')


def test_lazy_step_result_equals_the_eager_one_bit_for_bit():
    """TTSC_STEP_LAZY: the device-guarded updates and the deferred read-back change nothing — losses and every parameter after three steps are
    the bits of the host-checked step; the result is a mapping that fills itself on first access"""
    import random
    from ttscube_amd.networks.cubegan import Cubegan
    from ttscube_amd.networks import training as T
    rng = np.random.RandomState(0)
    batch, enc = _batch(2, 12, rng)
    outs, params = [], []
    old = T.STEP_LAZY
    try:
        for lazy in (True, False):
            T.STEP_LAZY = lazy
            torch.manual_seed(0)
            model = Cubegan(enc, conditioning=None, train=True).cuda()
            model.train()
            opts = T.cubegan_configure_optimizers(model)
            r = random.Random(1)
            res = [T.cubegan_training_step(model, batch, opts, rng=r) for _ in range(3)]
            if lazy:
                assert isinstance(res[-1], T.StepLosses) and res[-1].pending
                via_class = model.training_step(batch, 0, rng=random.Random(5))
                assert via_class.pending and set(via_class) >= {'loss_g', 'loss_d', 'loss_t', 'loss_v', 'loss', 'lr'}
                assert abs(via_class['loss'] - (via_class['loss_g'] + via_class['loss_d'] + via_class['loss_t'])) < 1e-6
            else:
                assert isinstance(res[-1], dict)
                model.training_step(batch, 0, rng=random.Random(5))
            outs.append([dict(o) for o in res])
            params.append([p.detach().clone() for p in model.parameters()])
    finally:
        T.STEP_LAZY = old
    assert outs[0] == outs[1], (outs[0], outs[1])
    assert all(torch.equal(a, b) for a, b in zip(*params))


@pytest.mark.parametrize('switch', ['fmap_raw', 'wbank_reuse'])
def test_step_shortcuts_do_not_change_a_bit(switch):
    """TTSC_FMAP_RAW (the feature-matching loss applies the discriminators' leaky-relu inside its launch instead of reading materialised
    activations) and TTSC_WBANK_REUSE (a weight bank whose parameters did not move since its last preparation is not prepared again): losses
    and every parameter after three steps are the bits of the long way round"""
    import random
    from ttscube_amd.hifigan import wbank as W
    from ttscube_amd.networks.cubegan import Cubegan
    from ttscube_amd.networks import training as T
    rng = np.random.RandomState(0)
    batch, enc = _batch(2, 12, rng)
    outs, params = [], []
    old = (T.FMAP_RAW, W.REUSE)
    try:
        for on in (True, False):
            T.FMAP_RAW, W.REUSE = old
            if switch == 'fmap_raw':
                T.FMAP_RAW = on
            else:
                W.REUSE = on
            torch.manual_seed(0)
            model = Cubegan(enc, conditioning=None, train=True).cuda()
            model.train()
            opts = T.cubegan_configure_optimizers(model)
            r = random.Random(1)
            outs.append([dict(T.cubegan_training_step(model, batch, opts, rng=r)) for _ in range(3)])
            params.append([p.detach().clone() for p in model.parameters()])
    finally:
        T.FMAP_RAW, W.REUSE = old
    assert outs[0] == outs[1], (outs[0], outs[1])
    assert all(torch.equal(a, b) for a, b in zip(*params))


def test_feature_loss_over_raw_feature_maps_equals_the_activated_one():
    """losses_hip.feature_loss over RawFmap pairs (leaky-relu inside the launch) against the same loss over materialised activations: value and
    both gradients, bit for bit (x * slope is the activation's own product; its derivative is the factor torch's backward applies)"""
    from ttscube_amd.hifigan.losses_hip import RawFmap, feature_loss
    g = torch.Generator().manual_seed(11)
    shapes = [(2, 32, 700), (2, 128, 233), (2, 1, 51)]
    xr = [torch.randn(s, generator=g).cuda() for s in shapes]
    xg = [torch.randn(s, generator=g).cuda().requires_grad_(True) for s in shapes]
    slopes = [0.1, 0.1, 1.0]
    l_raw = feature_loss([[RawFmap(a, s) for a, s in zip(xr, slopes)]], [[RawFmap(b, s) for b, s in zip(xg, slopes)]])
    g_raw = torch.autograd.grad(l_raw * 3.0, xg)
    act = lambda t, s: torch.nn.functional.leaky_relu(t, s) if s != 1.0 else t
    l_act = feature_loss([[act(a, s) for a, s in zip(xr, slopes)]], [[act(b, s) for b, s in zip(xg, slopes)]])
    g_act = torch.autograd.grad(l_act * 3.0, xg)
    assert torch.equal(l_raw, l_act)
    for a, b in zip(g_raw, g_act):
        assert torch.equal(a, b)
