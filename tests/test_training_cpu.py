"""CPU: the native training entry points fail loudly without a HIP device (no CPU fallback), and the host-side index
maps of the ConvTranspose1d backward (phase de-interleave) are consistent with torch's own convolution arithmetic."""
import numpy as np
import pytest
import torch

from ttscube_amd import _lib


def test_training_paths_refuse_cpu_tensors():
    from ttscube_amd.hifigan.autograd import generator_forward_with_grad
    from ttscube_amd.hifigan.env import AttrDict
    from ttscube_amd.hifigan.models import Generator
    from ttscube_amd.networks.gru_autograd import gru_forward_train
    from ttscube_amd.networks.lstm_autograd import lstm_forward_train
    from oracle import hifigan_ref as R
    h = dict(R.CONFIG_V1, upsample_initial_channel=32)
    g = Generator(AttrDict(h))
    with pytest.raises(_lib.TTSCError):
        generator_forward_with_grad(g, torch.zeros(1, 80, 4))
    with pytest.raises(_lib.TTSCError):
        lstm_forward_train(torch.nn.LSTM(8, 8, batch_first=True), torch.zeros(1, 3, 8))
    with pytest.raises(_lib.TTSCError):
        gru_forward_train(torch.nn.GRU(8, 8, batch_first=True), torch.zeros(1, 3, 8))


@pytest.mark.parametrize('cfg', [dict(K=16, stride=5, padding=5), dict(K=16, stride=3, padding=6), dict(K=4, stride=4, padding=0),
                                 dict(K=5, stride=2, padding=1), dict(K=7, stride=3, padding=0)])
def test_conv_transpose_phase_maps_reproduce_torch_gradients(cfg):
    """The ConvTranspose1d backward runs as stride-1 problems on the phase-de-interleaved output gradient
    (hifigan/autograd.py: TrainConv.taps_t / maps_t / deinterleave).  Evaluate exactly that formulation with torch CPU ops
    and compare with torch autograd of conv_transpose1d: validates the index maps without a GPU."""
    import torch.nn.functional as F
    from ttscube_amd.hifigan.autograd import TrainConv
    Ci, Co, K, u, p = 6, 5, cfg['K'], cfg['stride'], cfg['padding']
    tc = TrainConv.__new__(TrainConv)      # host-side logic only: no device handle
    tc.Cin, tc.Cout, tc.K, tc.stride, tc.padding, tc.dilation, tc.transposed = Ci, Co, K, u, p, 1, True
    tc._maps = None
    g = torch.Generator().manual_seed(0)
    B, Lin = 2, 11
    x = torch.randn(B, Ci, Lin, generator=g, requires_grad=True)
    w = torch.randn(Ci, Co, K, generator=g, requires_grad=True)
    y = F.conv_transpose1d(x, w, stride=u, padding=p)
    gy = torch.randn(y.shape, generator=g)
    gx0, gw0 = torch.autograd.grad(y, (x, w), gy)
    m_lo, _, M = tc.taps_t()
    dyp = tc.deinterleave(gy, Lin)                                     # [B, u*Co, Lin+M-1]
    w_map, g_map = tc.maps_t(torch.device('cpu'))
    wz = torch.cat([w.detach().reshape(-1), torch.zeros(1)])
    wd = wz[w_map]                                                     # [Ci, u*Co, M]: data-gradient weights
    gx1 = F.conv1d(dyp, wd)                                            # stride-1 conv, no padding -> [B, Ci, Lin]
    assert gx1.shape == gx0.shape and float((gx1 - gx0).abs().max()) < 1e-4
    # weight gradient: G[v, ci, j] = sum_{b,t} dyP[b,v,t] * x[b,ci,t - j]
    G = torch.zeros(u * Co, Ci, M)
    xd = x.detach()
    for jj in range(M):
        xs = F.pad(xd, (jj, M - 1 - jj))                               # xs[t] = x[t - jj]
        G[:, :, jj] = torch.einsum('bvt,bct->vc', dyp, xs)
    gw1 = G.reshape(-1)[g_map]
    assert gw1.shape == gw0.shape and float((gw1 - gw0).abs().max()) < 1e-3


def test_flat_adamw_gathers_gradients_into_the_arena():
    """FlatAdamW (round 4): zero_grad drops the gradients, autograd hands its own tensors over, gather() brings them into the arena — whole,
    or range by range as the exchange chunks ask for them, with a parameter cut by a range boundary and one without a gradient (zeros) —
    and afterwards `.grad` is the arena view again.  Layout logic only: runs on CPU tensors (step() itself needs the HIP kernel)."""
    from ttscube_amd.optim import FlatAdamW
    torch.manual_seed(0)
    a, b, c = (torch.nn.Parameter(torch.randn(n)) for n in (70, 130, 5))
    opt = FlatAdamW([a, b, c], lr=1e-3)
    x = torch.randn(3)

    def backward(with_c=True):
        loss = (a * x[0]).sum() + (b * b).sum() * x[1] + ((c * x[2]).sum() if with_c else 0.0)
        loss.backward()

    backward()
    opt.ensure_built()                       # first backward decides who is live; its gradients are copied in
    assert opt.live == [0, 1, 2] and opt.offsets == [0, 128, 320] and not opt._dirty
    assert a.grad.data_ptr() == opt.g.data_ptr() and torch.equal(opt.g[128:258], b.grad)
    want = [p.grad.clone() for p in (a, b, c)]
    opt.zero_grad()
    assert a.grad is None and b.grad is None and opt._dirty
    backward()
    assert a.grad.data_ptr() != opt.g.data_ptr()      # autograd's own tensor, not an accumulation into the arena
    opt.g.fill_(7.0)
    opt.gather(0, 200)                                # a chunk boundary inside b
    assert torch.equal(opt.g[:70], want[0]) and torch.equal(opt.g[128:200], want[1][:72]) and bool((opt.g[200:258] == 7).all())
    opt.gather(200, opt.numel)
    opt.grads_in_arena()
    for p, w in zip((a, b, c), want):
        assert torch.equal(p.grad, w) and opt.g.data_ptr() <= p.grad.data_ptr() < opt.g.data_ptr() + opt.g.numel() * 4
    assert not opt._dirty
    # a live parameter that gets no gradient in some step contributes zeros; gathering twice changes nothing
    opt.zero_grad()
    backward(with_c=False)
    opt.gather()
    opt.gather()
    opt.grads_in_arena()
    assert torch.equal(a.grad, want[0]) and torch.equal(b.grad, want[1]) and bool((c.grad == 0).all())
