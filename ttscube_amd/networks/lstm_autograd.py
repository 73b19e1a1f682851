"""Native training path of the mel-decoder LSTMs (SURVEY.md §8 rows a5/a6 inside a9).

The reference trains `Languasito2` / `CubenetTextcoder` through torch.nn.LSTM (cube/networks/modules.py:873-905,
textcoder.py:55-92), which on ROCm is MIOpen's per-time-step kernel chain: ~8 000 launches per Cubegan step at the GAN
crop sizes.  Here one layer (both directions) is one `torch.autograd.Function`:

  forward   ttsc_linear_forward            x W_ih^T + (b_ih + b_hh) for all steps (fp32 MFMA GEMM)
            ttsc_lstm_seq_forward_train    persistent recurrence kernel, saves gates + cell states
  backward  ttsc_lstm_seq_backward         persistent backward-through-time kernel -> gate gradients of all steps
            ttsc_gemm (NN, TN split-K)     dx = dG W_ih,  dW_ih = dG^T x,  dW_hh = dG^T h_prev (h_prev by row shift);  ttsc_colsum: biases

W_hh is re-packed on the device every step (`ttsc_lstm_pack_whh_device`, forward and transposed layouts)."""
import torch

from .. import _lib
from ..hip_layers import colsum_hip, gemm_hip, linear_hip


def _pack(whh, transpose):
    nd, H4, H = whh.shape
    out = torch.empty(nd * H4 * H, dtype=torch.float32, device=whh.device)
    with _lib.on_device(whh.device):
        _lib.check(_lib.lib().ttsc_lstm_pack_whh_device(_lib.dev_ptr(whh), nd, H, int(transpose), _lib.dev_ptr(out),
                                                        _lib.current_stream()), 'ttsc_lstm_pack_whh_device')
    return out


class HipLSTMLayerFn(torch.autograd.Function):
    """One (bi)directional LSTM layer over a padded batch: x [B,T,in] -> y [B,T,ndir*H].
    params = (w_ih, w_hh, b_ih, b_hh) per direction, torch.nn.LSTM layout (gate order i,f,g,o)."""

    @staticmethod
    def forward(ctx, x, nd, *params):
        x = x.contiguous().float()
        B, T, _ = x.shape
        w_ih = [params[4 * d].detach() for d in range(nd)]
        w_hh = [params[4 * d + 1].detach() for d in range(nd)]
        bias = torch.cat([params[4 * d + 2].detach() + params[4 * d + 3].detach() for d in range(nd)])
        H = w_hh[0].shape[1]
        wih = torch.cat(w_ih, dim=0).contiguous()                      # [nd*4H, in]
        whh = torch.stack(w_hh, dim=0).contiguous()                    # [nd, 4H, H]
        xg = linear_hip(x, wih, bias)                                  # [B, T, nd*4H]
        y = torch.empty((B, T, nd * H), dtype=torch.float32, device=x.device)
        gates = torch.empty((B, T, nd * 4 * H), dtype=torch.float32, device=x.device)
        cst = torch.empty((B, T, nd * H), dtype=torch.float32, device=x.device)
        with _lib.on_device(x.device):
            _lib.check(_lib.lib().ttsc_lstm_seq_forward_train(_lib.dev_ptr(xg), _lib.dev_ptr(_pack(whh, False)), _lib.dev_ptr(y), None,
                                                              B, T, H, nd, nd * H, 0, _lib.dev_ptr(gates), _lib.dev_ptr(cst),
                                                              _lib.current_stream()), 'ttsc_lstm_seq_forward_train')
        ctx.save_for_backward(x, wih, whh, gates, cst, y)
        ctx.nd, ctx.H = nd, H
        return y

    @staticmethod
    def backward(ctx, dy):
        x, wih, whh, gates, cst, y = ctx.saved_tensors
        nd, H = ctx.nd, ctx.H
        B, T, _ = x.shape
        dy = dy.contiguous()
        dG = torch.empty_like(gates)
        with _lib.on_device(x.device):
            _lib.check(_lib.lib().ttsc_lstm_seq_backward(_lib.dev_ptr(dy), _lib.dev_ptr(gates), _lib.dev_ptr(cst),
                                                         _lib.dev_ptr(_pack(whh, True)), _lib.dev_ptr(dG), None, B, T, H, nd, nd * H, 0,
                                                         _lib.current_stream()), 'ttsc_lstm_seq_backward')
        dG2 = dG.reshape(B * T, nd * 4 * H)
        dx = gemm_hip(dG2, wih).reshape(x.shape) if ctx.needs_input_grad[0] else None   # dG . W_ih                          (NN)
        dwih = gemm_hip(dG2, x.reshape(B * T, -1), trans_a=True)                         # dG^T . x, split over the B*T rows   (TN) [nd*4H, in]
        db = colsum_hip(dG2)
        y2 = y.reshape(B * T, nd * H)
        grads = []
        for d in range(nd):
            sl = slice(d * 4 * H, (d + 1) * 4 * H)
            # dG_d^T . h_prev: the forward direction's previous step is t - 1, the reverse direction's t + 1 — a row shift of the
            # direction's column slice of y inside every sequence (zero at the sequence's first step), nothing is materialised
            dwhh = gemm_hip(dG2[:, sl], y2[:, d * H:(d + 1) * H], trans_a=True, b_row_shift=-1 if d == 0 else 1, b_period=T)
            grads += [dwih[sl], dwhh, db[sl], db[sl]]
        return (dx, None) + tuple(grads)


def lstm_forward_train(m, x):
    """Differentiable forward of a torch.nn.LSTM parameter set `m` (batch_first, zero initial state) on the HIP kernels."""
    if not x.is_cuda:
        raise _lib.TTSCError('LSTM training needs a HIP device; no CPU path')
    nd = 2 if m.bidirectional else 1
    assert m.batch_first and float(m.dropout) == 0.0
    h = x
    for l in range(m.num_layers):
        ps = []
        for sfx in ['', '_reverse'][:nd]:
            ps += [getattr(m, 'weight_ih_l%d%s' % (l, sfx)), getattr(m, 'weight_hh_l%d%s' % (l, sfx)),
                   getattr(m, 'bias_ih_l%d%s' % (l, sfx)), getattr(m, 'bias_hh_l%d%s' % (l, sfx))]
        h = HipLSTMLayerFn.apply(h, nd, *ps)
    return h
