"""Native training path of the WaveRNN GRU (`WaveRNN._train_forward` modules.py:505-539).

The reference runs torch.nn.GRU over the teacher-forced sequence (24 000 steps per utterance) and lets autograd walk it
back; on ROCm that is MIOpen's per-time-step chain.  Here one GRU layer is one `torch.autograd.Function`:

  forward   ttsc_linear_forward      W_ih x + b_ih for all steps (fp32 MFMA GEMM)
            ttsc_gru_seq_forward     persistent recurrence kernel, saves r, z, n and W_hn h + b_hn
  backward  ttsc_gru_seq_backward    persistent backward-through-time kernel -> per-step gate gradients
            ttsc_gemm x3             dx = dGi W_ih (NN),  dW_ih = dGi^T x,  dW_hh = dGh^T h_prev (TN, split-K, h_prev by row shift)
            ttsc_colsum x2           bias gradients"""
import torch

from .. import _lib
from ..hip_layers import colsum_hip, gemm_hip, linear_hip


def _pack(whh, transpose):
    H = whh.shape[1]
    out = torch.empty(3 * H * H, dtype=torch.float32, device=whh.device)
    with _lib.on_device(whh.device):
        _lib.check(_lib.lib().ttsc_gru_pack_whh_device(_lib.dev_ptr(whh), H, int(transpose), _lib.dev_ptr(out), _lib.current_stream()),
                   'ttsc_gru_pack_whh_device')
    return out


class HipGRUFn(torch.autograd.Function):
    """x [B,T,in] -> y [B,T,H], zero initial state; parameters in torch.nn.GRU layout (gate order r,z,n)."""

    @staticmethod
    def forward(ctx, x, w_ih, w_hh, b_ih, b_hh):
        x = x.contiguous().float()
        B, T, _ = x.shape
        wih, whh = w_ih.detach().contiguous(), w_hh.detach().contiguous()
        H = whh.shape[1]
        xg = linear_hip(x, wih, b_ih.detach())
        y = torch.empty((B, T, H), dtype=torch.float32, device=x.device)
        saved = torch.empty((B, T, 4 * H), dtype=torch.float32, device=x.device)
        with _lib.on_device(x.device):
            _lib.check(_lib.lib().ttsc_gru_seq_forward(_lib.dev_ptr(xg), _lib.dev_ptr(_pack(whh, False)), _lib.dev_ptr(b_hh.detach().contiguous()),
                                                       _lib.dev_ptr(y), _lib.dev_ptr(saved), None, B, T, H, _lib.current_stream()),
                       'ttsc_gru_seq_forward')
        ctx.save_for_backward(x, wih, whh, saved, y)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, wih, whh, saved, y = ctx.saved_tensors
        B, T, _ = x.shape
        H = whh.shape[1]
        dy = dy.contiguous()
        dgi = torch.empty((B, T, 3 * H), dtype=torch.float32, device=x.device)
        dgh = torch.empty_like(dgi)
        with _lib.on_device(x.device):
            _lib.check(_lib.lib().ttsc_gru_seq_backward(_lib.dev_ptr(dy), _lib.dev_ptr(saved), _lib.dev_ptr(y), None, _lib.dev_ptr(_pack(whh, True)),
                                                        _lib.dev_ptr(dgi), _lib.dev_ptr(dgh), B, T, H, _lib.current_stream()),
                       'ttsc_gru_seq_backward')
        gi2, gh2 = dgi.reshape(B * T, 3 * H), dgh.reshape(B * T, 3 * H)
        dx = gemm_hip(gi2, wih).reshape(x.shape) if ctx.needs_input_grad[0] else None       # dGi . W_ih                       (NN)
        dwih = gemm_hip(gi2, x.reshape(B * T, -1), trans_a=True)                             # dGi^T . x, split over the B*T rows (TN)
        # dGh^T . h_prev with h_prev[b, t] = y[b, t-1] (0 at t = 0): the GEMM reads y one row up inside every sequence
        dwhh = gemm_hip(gh2, y.reshape(B * T, H), trans_a=True, b_row_shift=-1, b_period=T)
        return dx, dwih, dwhh, colsum_hip(gi2), colsum_hip(gh2)


def gru_forward_train(m, x):
    """Differentiable forward of a single-layer unidirectional torch.nn.GRU parameter set `m` (batch_first) on the HIP kernels."""
    if not x.is_cuda:
        raise _lib.TTSCError('GRU training needs a HIP device; no CPU path')
    assert m.batch_first and not m.bidirectional and m.num_layers == 1
    return HipGRUFn.apply(x, m.weight_ih_l0, m.weight_hh_l0, m.bias_ih_l0, m.bias_hh_l0)
