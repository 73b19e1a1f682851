"""Native training path of the mel decoder's non-recurrent layers: embeddings, the three-layer char-CNN
(tanh(Conv1d k=3)) and the output Linears of `Languasito2` (cube/networks/modules.py:869-914, 916-999) as `autograd.Function`s
over the HIP kernels — row gather / ordered scatter-add, `conv_mfma_kernel` + `conv_wgrad_kernel` (hifigan/autograd.py::HipConvFn)
and `gemm_nt_kernel` for the three GEMMs of a Linear.  The LSTMs are in lstm_autograd.py."""
import torch

from .. import _lib
from ..hifigan.autograd import TrainConv, hip_conv
from ..hip_layers import colsum_hip, gemm_hip, linear_hip


class HipLinearFn(torch.autograd.Function):
    """y = x W^T + b on gemm_nt_kernel; backward dx = dy W (NN) and dW = dy^T x (TN, split-K) on gemm_general_kernel — no transposed
    copies —, db = fixed-order column sums (colsum kernels)"""

    @staticmethod
    def forward(ctx, x, w, b):
        shp = x.shape
        x2 = x.reshape(-1, shp[-1]).contiguous()
        ctx.save_for_backward(x2, w)
        ctx.shp, ctx.has_b = shp, b is not None
        y = linear_hip(x2, w.detach().contiguous(), b.detach() if b is not None else None)
        return y.view(*shp[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, w = ctx.saved_tensors
        dy2 = dy.reshape(-1, dy.shape[-1]).contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = gemm_hip(dy2, w.detach()).view(ctx.shp)                   # dy [M, N] . W [N, K]                      (NN)
        if ctx.needs_input_grad[1]:
            dw = gemm_hip(dy2, x2, trans_a=True)                           # dy^T [N, M] . x [M, K], split over the M rows (TN)
        if ctx.has_b and ctx.needs_input_grad[2]:
            db = colsum_hip(dy2)
        return dx, dw, db


def hip_linear(x, w, b=None):
    return HipLinearFn.apply(x, w, b)


class HipEmbeddingFn(torch.autograd.Function):
    """table[idx] (nn.Embedding with padding_idx: that row gets no gradient)"""

    @staticmethod
    def forward(ctx, table, idx, padding_idx, sorted_idx=False):
        i32 = idx.reshape(-1).to(torch.int32).contiguous()
        V, Cc = table.shape
        out = torch.empty((i32.numel(), Cc), dtype=torch.float32, device=table.device)
        with _lib.on_device(table.device):
            _lib.check(_lib.lib().ttsc_rows_gather(_lib.dev_ptr(table.detach().contiguous()), _lib.dev_ptr(i32), _lib.dev_ptr(out), i32.numel(), Cc, V,
                                                   _lib.current_stream()), 'ttsc_rows_gather')
        ctx.save_for_backward(i32)
        ctx.V, ctx.Cc, ctx.pad = V, Cc, -1 if padding_idx is None else int(padding_idx)
        ctx.sorted_idx = bool(sorted_idx) and padding_idx is None   # non-decreasing index list: the adjoint is a segmented sum, O(n C)
        return out.view(*idx.shape, Cc)

    @staticmethod
    def backward(ctx, g):
        i32, = ctx.saved_tensors
        g2 = g.reshape(-1, ctx.Cc).contiguous()
        gt = torch.empty((ctx.V, ctx.Cc), dtype=torch.float32, device=g.device)
        with _lib.on_device(g.device):
            if ctx.sorted_idx:
                _lib.check(_lib.lib().ttsc_rows_segment_sum(_lib.dev_ptr(g2), _lib.dev_ptr(i32), _lib.dev_ptr(gt), i32.numel(), ctx.Cc, ctx.V,
                                                            _lib.current_stream()), 'ttsc_rows_segment_sum')
            else:
                _lib.check(_lib.lib().ttsc_rows_scatter_add(_lib.dev_ptr(g2), _lib.dev_ptr(i32), _lib.dev_ptr(gt), i32.numel(), ctx.Cc, ctx.V, ctx.pad,
                                                            _lib.current_stream()), 'ttsc_rows_scatter_add')
        return gt, None, None, None


def hip_embedding(emb, idx):
    return HipEmbeddingFn.apply(emb.weight, idx, emb.padding_idx)


def char_cnn_train(lang, name, h):
    """h [B, C, N] through the (ConvNorm k=3 pad=1, Tanh) x 3 stack `name` of `lang`; one TrainConv (forward / dgrad / wgrad handles) per layer"""
    cache = lang.__dict__.setdefault('_train_cnn', {})
    ml = getattr(lang, name)
    for i, layer in enumerate(ml):
        if not hasattr(layer, 'conv'):
            continue
        c = layer.conv
        tc = cache.get((name, i))
        if tc is None:
            tc = cache[(name, i)] = TrainConv(c.in_channels, c.out_channels, c.kernel_size[0], padding=c.padding[0], dilation=c.dilation[0])
        h = torch.tanh(hip_conv(tc, h, c.weight, c.bias))
    return h
