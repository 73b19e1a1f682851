"""ORACLE (test infrastructure — never imported by the product path).

CPU restatement of the WaveRNN TRAINING path in plain differentiable torch algebra (explicit GRU recurrence, no nn.GRU / nn.Module / optimizer
objects), so that autograd through it and ten lines of Adam reproduce the reference's whole optimisation step:

  WaveRNN._train_forward                 cube/networks/modules.py:505-539   (repeat-upsampled mel, low-resolution branch, GRU stack, two Linears)
  WaveRNN.training_step                  cube/networks/modules.py:553-563   (teacher input = target shifted right by one, 0 first)
  MULAWOutput.loss / encode              cube/networks/loss.py:222-225, 236-254
  CubenetVocoder.training_step           cube/networks/vocoder.py:136-156   (both nets, clip_grad_norm 5, Adam x 2, lr0 / (1 + 5e-5 * step))

Pinned against the reference itself by tools/gen_golden_training.py -> tests/golden/vocoder_step_*.npz -> tests/test_oracle_wavernn.py.
Only tests/ may import this."""
import numpy as np
import torch
import torch.nn.functional as F


def gru(x, w_ih, w_hh, b_ih, b_hh):
    """nn.GRU(batch_first=True, num_layers=1) from zero state: gate order r, z, n; n = tanh(W_in x + b_in + r * (W_hn h + b_hn)); h' = (1 - z) n + z h."""
    B, T, _ = x.shape
    H = w_hh.shape[1]
    h = torch.zeros(B, H)
    xg = x @ w_ih.t() + b_ih
    ys = []
    for t in range(T):
        hg = h @ w_hh.t() + b_hh
        r = torch.sigmoid(xg[:, t, :H] + hg[:, :H])
        z = torch.sigmoid(xg[:, t, H:2 * H] + hg[:, H:2 * H])
        n = torch.tanh(xg[:, t, 2 * H:] + r * hg[:, 2 * H:])
        h = (1 - z) * n + z * h
        ys.append(h)
    return torch.stack(ys, dim=1)


def mulaw_encode(x):
    """loss.py:236-254 (torch branch)"""
    mu = torch.tensor([255.0])
    x_mu = torch.sign(x) * torch.log1p(mu * torch.abs(x)) / torch.log1p(mu)
    return torch.clip(((x_mu + 1) / 2 * mu + 0.5).long(), 0, 255)


def train_logits(sd, mel, x_in, x_low, upsample, upsample_low=10):
    """modules.py:505-539.  sd: one WaveRNN's state_dict; x_low None for the low-resolution net (use_lowres=False)."""
    up = mel.repeat_interleave(upsample, dim=1)                      # UpsampleNetR (modules.py:378-389): frame t -> rows t*r .. t*r + r-1
    if x_low is not None:
        interp = F.interpolate(x_low.unsqueeze(1), upsample_low * x_low.shape[1], mode='linear').squeeze(1)
        h = x_low.unsqueeze(1)
        for i in range(3):
            h = torch.tanh(F.conv1d(h, sd['_lowres_conv.%d.conv.weight' % i], sd['_lowres_conv.%d.conv.bias' % i], padding=3))
        ux = h.repeat_interleave(upsample_low, dim=2).permute(0, 2, 1)
        m = min(up.shape[1], x_in.shape[1], ux.shape[1], interp.shape[1])
        hidden = torch.cat([up[:, :m], ux[:, :m], interp.unsqueeze(2), x_in[:, :m].unsqueeze(2)], dim=-1)   # (interp is NOT truncated: modules.py:527)
    else:
        m = min(up.shape[1], x_in.shape[1])
        hidden = torch.cat([up[:, :m], x_in[:, :m].unsqueeze(2)], dim=-1)
    layer = 0
    while '_rnns.%d.weight_ih_l0' % layer in sd:
        p = '_rnns.%d.' % layer
        hidden = gru(hidden, sd[p + 'weight_ih_l0'], sd[p + 'weight_hh_l0'], sd[p + 'bias_ih_l0'], sd[p + 'bias_hh_l0'])
        layer += 1
    pre = torch.tanh(hidden @ sd['_preoutput.linear_layer.weight'].t() + sd['_preoutput.linear_layer.bias'])
    return pre @ sd['_output.linear_layer.weight'].t() + sd['_output.linear_layer.bias']


def train_loss(sd, mel, x, x_low, upsample):
    """modules.py:553-563 + loss.py:222-225"""
    x_in = F.pad(x[:, :-1], (1, 0), mode='constant', value=0)
    out = train_logits(sd, mel, x_in, x_low, upsample)
    return F.cross_entropy(out.reshape(out.shape[0] * out.shape[1], -1), mulaw_encode(x).reshape(-1))


class Adam:
    """torch.optim.Adam defaults (betas 0.9 / 0.999, eps 1e-8, no weight decay), spelled out"""

    def __init__(self, keys):
        self.m = {k: 0.0 for k in keys}
        self.v = {k: 0.0 for k in keys}
        self.t = 0

    def step(self, sd, grads, lr):
        self.t += 1
        for k, g in grads.items():
            self.m[k] = 0.9 * self.m[k] + 0.1 * g
            self.v[k] = 0.999 * self.v[k] + 0.001 * g * g
            mhat = self.m[k] / (1 - 0.9 ** self.t)
            den = (self.v[k] / (1 - 0.999 ** self.t)).sqrt() + 1e-8
            sd[k] = (sd[k] - lr * mhat / den).detach()


def clip_(grads, max_norm):
    """torch.nn.utils.clip_grad_norm_: total 2-norm; scale by max_norm / (norm + 1e-6) when that is below 1.  -> the norm before clipping"""
    norm = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())).float()
    coef = torch.clamp(max_norm / (norm + 1e-6), max=1.0)
    for k in grads:
        grads[k] = grads[k] * coef
    return float(norm)


def vocoder_training_steps(sd, batches, lr0, upsample=240, upsample_low=10):
    """vocoder.py:136-156 for consecutive batches.  sd: CubenetVocoder state_dict ('_wavernn_hr.' / '_wavernn_lr.' prefixes).
    -> per step dict(loss_lr, loss_hr, norm_lr, norm_hr, alpha), and sd is updated in place.  `_skip` (dead layer, modules.py:424) has no gradient:
    torch's Adam skips parameters whose .grad is None, so it never moves."""
    nets = {}
    for pre in ('_wavernn_lr.', '_wavernn_hr.'):
        keys = [k for k in sd if k.startswith(pre) and '._skip.' not in k]
        nets[pre] = (keys, Adam(keys))
    lr = lr0
    out = []
    for step, b in enumerate(batches, 1):
        rec = {}
        grads_all = {}
        for pre, low, x, name in (('_wavernn_lr.', None, b['x_low'], 'lr'), ('_wavernn_hr.', b['x_low'], b['x'], 'hr')):
            keys = nets[pre][0]
            local = {k[len(pre):]: sd[k].detach().clone().requires_grad_(True) for k in keys}
            loss = train_loss(local, b['mel'], x, low, upsample if low is not None else upsample // upsample_low)
            gs = torch.autograd.grad(loss, [local[k[len(pre):]] for k in keys])
            grads_all[pre] = dict(zip(keys, gs))
            rec["loss_" + name] = float(loss.detach())
        for pre, name in (('_wavernn_lr.', 'lr'), ('_wavernn_hr.', 'hr')):
            rec['norm_' + name] = clip_(grads_all[pre], 5)
            nets[pre][1].step(sd, grads_all[pre], lr)
        lr = lr0 / (1 + 5e-5 * step)
        rec['alpha'] = lr
        out.append(rec)
    return out
