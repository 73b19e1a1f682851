"""Thin Python handles over the C-ABI layer objects of libttscube_hip.so (no compute in Python).

``Conv1dHip`` stands where the reference uses torch.nn.Conv1d / ConvTranspose1d inside ConvNorm
(cube/networks/modules.py:37-55), PostNet (modules.py:117-145), the WaveRNN low-res convs (modules.py:416-420)
and the char CNNs (textcoder.py:44-53, modules.py:850-871)."""
import ctypes as C
import os

import torch

from . import _lib

ACT = {None: _lib.ACT_NONE, 'none': _lib.ACT_NONE, 'tanh': _lib.ACT_TANH, 'relu': _lib.ACT_RELU,
       'sigmoid': _lib.ACT_SIGMOID}


class Conv1dHip:
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, transposed=False, groups=1):
        L = _lib.lib()
        _lib.require_gpu()
        self.groups = int(groups)
        self.cfg = _lib.Conv1dCfg(in_channels, out_channels, kernel_size, stride, padding, dilation, int(transposed), self.groups)
        self._h = C.c_void_p()
        _lib.check(L.ttsc_conv1d_create(C.byref(self.cfg), C.byref(self._h)), 'ttsc_conv1d_create')

    def set_precision(self, precision):
        """'fp32' (exact fp32 MFMA) or 'f16x3' (split-precision fp16 MFMA, ~2^-21 relative)."""
        mode = {'fp32': _lib.PREC_FP32, 'f16x3': _lib.PREC_F16X3}[precision]
        _lib.check(_lib.lib().ttsc_conv1d_set_precision(self._h, mode), 'ttsc_conv1d_set_precision')
        return self

    def set_activation_scale(self, scale):
        """power-of-two pre-scale of this layer's input on the split-precision path (ttsc_conv1d_set_activation_scale)"""
        _lib.check(_lib.lib().ttsc_conv1d_set_activation_scale(self._h, float(scale)), 'ttsc_conv1d_set_activation_scale')
        return self

    def set_weight(self, weight, bias=None):
        w = weight.detach().float().cpu().contiguous()
        exp = ((self.cfg.in_channels, self.cfg.out_channels) if self.cfg.transposed else
               (self.cfg.out_channels, self.cfg.in_channels // self.groups)) + (self.cfg.kernel_size,)
        if tuple(w.shape) != exp:
            raise _lib.TTSCError('Conv1dHip.set_weight: expected weight shape %s, got %s' % (exp, tuple(w.shape)))
        b = None
        if bias is not None:
            b = bias.detach().float().cpu().contiguous()
            if b.numel() != self.cfg.out_channels:
                raise _lib.TTSCError('Conv1dHip.set_weight: bias must have %d elements' % self.cfg.out_channels)
        _lib.check(_lib.lib().ttsc_conv1d_set_weight(self._h, C.c_void_p(w.data_ptr()),
                                                     C.c_void_p(b.data_ptr()) if b is not None else None),
                   'ttsc_conv1d_set_weight')

    def set_weight_device(self, weight, bias=None):
        """Training: (re)pack the fragments from the live fp32 device tensors on the current stream (no host copy)."""
        exp = ((self.cfg.in_channels, self.cfg.out_channels) if self.cfg.transposed else
               (self.cfg.out_channels, self.cfg.in_channels // self.groups)) + (self.cfg.kernel_size,)
        if tuple(weight.shape) != exp or not weight.is_cuda or weight.dtype != torch.float32 or not weight.is_contiguous():
            raise _lib.TTSCError('Conv1dHip.set_weight_device: need a contiguous fp32 device tensor of shape %s' % (exp,))
        if bias is not None and (bias.numel() != self.cfg.out_channels or not bias.is_cuda or bias.dtype != torch.float32):
            raise _lib.TTSCError('Conv1dHip.set_weight_device: bias must be an fp32 device tensor with %d elements' % self.cfg.out_channels)
        with _lib.on_device(weight.device):
            _lib.check(_lib.lib().ttsc_conv1d_set_weight_device(self._h, _lib.dev_ptr(weight),
                                                                _lib.dev_ptr(bias.contiguous()) if bias is not None else None,
                                                                _lib.current_stream()), 'ttsc_conv1d_set_weight_device')

    def set_weight_device_dgrad(self, fwd_weight):
        """Training: this handle is the data gradient of a Conv1d whose weight is `fwd_weight` [Cin(this), Cout(this), K]; the
        flipped / transposed view is read directly by the packing kernel."""
        exp = (self.cfg.in_channels, self.cfg.out_channels // self.groups, self.cfg.kernel_size)   # the forward layer's [Cout, Cin / groups, K]
        if tuple(fwd_weight.shape) != exp or not fwd_weight.is_cuda or fwd_weight.dtype != torch.float32 or not fwd_weight.is_contiguous():
            raise _lib.TTSCError('Conv1dHip.set_weight_device_dgrad: need a contiguous fp32 device tensor of shape %s' % (exp,))
        with _lib.on_device(fwd_weight.device):
            _lib.check(_lib.lib().ttsc_conv1d_set_weight_device_dgrad(self._h, _lib.dev_ptr(fwd_weight), _lib.current_stream()),
                       'ttsc_conv1d_set_weight_device_dgrad')

    def out_len(self, Lin):
        return int(_lib.lib().ttsc_conv1d_out_len(self._h, Lin))

    def __call__(self, x, resid=None, out=None, in_scale=1.0, in_slope=1.0, out_scale=1.0, act=None, accumulate=False,
                 gate=None, gate_slope=1.0):
        if not x.is_cuda:
            raise _lib.TTSCError('Conv1dHip: input must live on a HIP device; no CPU path')
        x = x.float().contiguous()
        B, Cin, Lin = x.shape
        if Cin != self.cfg.in_channels:
            raise _lib.TTSCError('Conv1dHip: expected %d input channels, got %d' % (self.cfg.in_channels, Cin))
        Lout = self.out_len(Lin)
        if out is None:
            if accumulate:
                raise _lib.TTSCError('Conv1dHip: accumulate=True needs an `out` tensor')
            out = torch.empty((B, self.cfg.out_channels, Lout), dtype=torch.float32, device=x.device)
        assert out.is_contiguous() and tuple(out.shape) == (B, self.cfg.out_channels, Lout)
        if resid is not None:
            resid = resid.float().contiguous()
            assert tuple(resid.shape) == tuple(out.shape)
        if gate is not None:
            assert gate.is_contiguous() and gate.dtype == torch.float32 and tuple(gate.shape) == tuple(out.shape)
        ep = _lib.Conv1dEpilogue(in_scale, in_slope, out_scale, ACT[act], int(accumulate),
                                 gate.data_ptr() if gate is not None else None, gate_slope)
        with _lib.on_device(x.device):
            _lib.check(_lib.lib().ttsc_conv1d_forward(self._h, _lib.dev_ptr(x), B, Lin, _lib.dev_ptr(out),
                                                      _lib.dev_ptr(resid) if resid is not None else None,
                                                      C.byref(ep), _lib.current_stream()), 'ttsc_conv1d_forward')
        return out

    def __del__(self):
        try:
            _lib.lib().ttsc_conv1d_destroy(self._h)
        except Exception:
            pass


SPLIT_GEMM = os.environ.get('TTSC_GEMM_SPLIT', '1') != '0'


def linear_hip(x, weight, bias=None, act=None, out=None, accumulate=False, split=False, lengths_dev=None, period=0):
    """y[..., :N] = act(x[..., :K] @ weight[N,K]^T + bias) on the MFMA GEMM (ttsc_linear_forward).
    x: device tensor [..., K] (last dim contiguous, rows at a constant stride); weight/bias: device tensors.
    split: run on the f16 matrix pipe with fp32-class accuracy (ttsc_linear_forward_split: hi / lo fp16 halves, three products — operands
    must lie inside the fp16 range, checked on the device and reported by _lib.check_split_status); shapes the split kernel does not take
    run on the exact fp32 kernel.  lengths_dev (int32 device tensor [B]) with period = T for x [B, T, K]: rows t >= lengths[b] are padding
    nobody reads — whole row tiles of padding are skipped and stay UNWRITTEN (split kernel only)."""
    if not x.is_cuda:
        raise _lib.TTSCError('linear_hip: input must live on a HIP device; no CPU path')
    K = x.shape[-1]
    N = weight.shape[0]
    assert weight.shape[1] == K, (tuple(weight.shape), K)
    x2 = x.float().contiguous().reshape(-1, K)
    M = x2.shape[0]
    w = weight.detach().float().contiguous()
    b = bias.detach().float().contiguous() if bias is not None else None
    if out is None:
        out2 = torch.empty((M, N), dtype=torch.float32, device=x.device)
        ldy = N
    else:
        assert out.is_cuda and out.dtype == torch.float32 and out.stride(-1) == 1
        out2 = out
        ldy = out.stride(-2) if out.dim() > 1 else N
    L = _lib.lib()
    with _lib.on_device(x.device):
        if split and SPLIT_GEMM and L.ttsc_linear_split_supported(M, N, K, K) and (x2.data_ptr() | w.data_ptr()) % 16 == 0:
            _lib.check(L.ttsc_linear_forward_split(_lib.dev_ptr(x2), _lib.dev_ptr(w), _lib.dev_ptr(b) if b is not None else None,
                                                   C.c_void_p(out2.data_ptr()), M, N, K, K, ldy, ACT[act], int(accumulate),
                                                   _lib.dev_ptr(lengths_dev) if lengths_dev is not None else None, int(period),
                                                   _lib.current_stream()), 'ttsc_linear_forward_split')
        else:
            _lib.check(L.ttsc_linear_forward(_lib.dev_ptr(x2), _lib.dev_ptr(w), _lib.dev_ptr(b) if b is not None else None,
                                             C.c_void_p(out2.data_ptr()), M, N, K, K, ldy, ACT[act], int(accumulate),
                                             _lib.current_stream()), 'ttsc_linear_forward')
    if out is None:
        return out2.reshape(tuple(x.shape[:-1]) + (N,))
    return out


def gemm_hip(a, b, trans_a=False, trans_b=False, out=None, accumulate=False, b_row_shift=0, b_period=0):
    """out[M, N] (+)= op(a) @ op(b) on the fp32 MFMA GEMM (ttsc_gemm): a is [M, K] ([K, M] when trans_a), b is [K, N] ([N, K] when
    trans_b); 2-D device tensors whose last dimension is contiguous (row strides are passed as leading dimensions, so column slices
    of a wider tensor work without a copy).  b_row_shift / b_period: see include/ttscube_hip.h (h_prev of a recurrence without
    materialising it).  Deterministic (split-K partials are added in a fixed order)."""
    for t in (a, b):
        if not (t.is_cuda and t.dtype == torch.float32 and t.dim() == 2 and t.stride(1) == 1):
            raise _lib.TTSCError('gemm_hip: operands must be 2-D fp32 device tensors with a contiguous last dimension')
    M, K = (a.shape[1], a.shape[0]) if trans_a else (a.shape[0], a.shape[1])
    N, Kb = (b.shape[0], b.shape[1]) if trans_b else (b.shape[1], b.shape[0])
    if Kb != K:
        raise _lib.TTSCError('gemm_hip: contraction lengths differ (%d vs %d)' % (K, Kb))
    if out is None:
        if accumulate:
            raise _lib.TTSCError('gemm_hip: accumulate=True needs an `out` tensor')
        out = torch.empty((M, N), dtype=torch.float32, device=a.device)
    assert out.is_cuda and out.dtype == torch.float32 and tuple(out.shape) == (M, N) and out.stride(1) == 1
    L = _lib.lib()
    wsb = int(L.ttsc_gemm_workspace_bytes(M, N, K))
    ws = torch.empty((wsb // 4,), dtype=torch.float32, device=a.device) if wsb else None
    with _lib.on_device(a.device):
        _lib.check(L.ttsc_gemm(int(trans_a), int(trans_b), M, N, K, C.c_void_p(a.data_ptr()), a.stride(0), C.c_void_p(b.data_ptr()), b.stride(0),
                               C.c_void_p(out.data_ptr()), out.stride(0), int(accumulate), int(b_row_shift), int(b_period),
                               C.c_void_p(ws.data_ptr()) if ws is not None else None, wsb, _lib.current_stream()), 'ttsc_gemm')
    return out


def colsum_hip(x, out=None, accumulate=False):
    """out[c] (+)= sum_r x[r, c] for a 2-D fp32 device tensor with a contiguous last dimension (ttsc_colsum; fixed summation order)"""
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1):
        raise _lib.TTSCError('colsum_hip: need a 2-D fp32 device tensor with a contiguous last dimension')
    R, Cn = x.shape
    if out is None:
        out = torch.empty((Cn,), dtype=torch.float32, device=x.device)
    L = _lib.lib()
    wsb = int(L.ttsc_colsum_workspace_bytes(R, Cn))
    ws = torch.empty((max(wsb // 4, 1),), dtype=torch.float32, device=x.device)
    with _lib.on_device(x.device):
        _lib.check(L.ttsc_colsum(C.c_void_p(x.data_ptr()), R, Cn, x.stride(0), C.c_void_p(out.data_ptr()), int(accumulate),
                                 C.c_void_p(ws.data_ptr()), wsb, _lib.current_stream()), 'ttsc_colsum')
    return out


class LSTMHip:
    """Device-side handle for a (stacked, optionally bidirectional) torch.nn.LSTM parameter set (batch_first).
    Input projections of all time steps run as one MFMA GEMM per layer; the recurrence runs in the persistent
    `lstm_seq_kernel` (one workgroup per utterance and direction)."""

    def __init__(self, lstm_module):
        self.m = lstm_module
        self.H = lstm_module.hidden_size
        self.num_layers = lstm_module.num_layers
        self.ndir = 2 if lstm_module.bidirectional else 1
        assert lstm_module.batch_first
        self._sig = None
        self._whh = []
        self._wih = []
        self._bias = []

    def _free(self):
        for p in self._whh:
            _lib.lib().ttsc_device_free(p)
        self._whh = []

    def __del__(self):
        try:
            self._free()
        except Exception:
            pass

    def _sync(self):
        pl = getattr(self, '_plist', None)
        if pl is None:      # (an nn.LSTM's Parameter objects are fixed: walk the module once)
            pl = self._plist = list(self.m.parameters())
        sig = tuple((p.data_ptr(), p._version) for p in pl)
        if sig == self._sig:
            return
        _lib.require_gpu()
        self._free()
        self._wih, self._bias = [], []
        dev = next(self.m.parameters()).device
        if dev.type == 'cpu':
            raise _lib.TTSCError('LSTMHip: parameters live on the CPU; move the module to a HIP device (no CPU path)')
        sfx = ['', '_reverse'][:self.ndir]
        for l in range(self.num_layers):
            g = lambda n: getattr(self.m, n).detach().float()
            wih = torch.cat([g('weight_ih_l%d%s' % (l, s)) for s in sfx], dim=0).contiguous()             # [ndir*4H, in]
            if wih.shape[1] % 4:      # the split-precision GEMM takes K % 4 == 0 only (Languasito2._cond_rnn reads 641 features: the exact kernel ran its
                wih = torch.nn.functional.pad(wih, (0, 4 - wih.shape[1] % 4)).contiguous()   # projection in 152 us against ~40): zero columns, zero-padded input
            bias = torch.cat([g('bias_ih_l%d%s' % (l, s)) + g('bias_hh_l%d%s' % (l, s)) for s in sfx], dim=0).contiguous()
            whh = torch.stack([g('weight_hh_l%d%s' % (l, s)) for s in sfx], dim=0).cpu().contiguous()      # [ndir,4H,H]
            ptr = C.c_void_p()
            _lib.check(_lib.lib().ttsc_lstm_pack_whh(C.c_void_p(whh.data_ptr()), self.ndir, self.H, C.byref(ptr)),
                       'ttsc_lstm_pack_whh')
            self._whh.append(ptr)
            self._wih.append(wih.to(dev))
            self._bias.append(bias.to(dev))
        self._sig = sig

    def __call__(self, x, lengths=None, hx=None, return_state=False):
        """x [B, T, in] -> y [B, T, ndir*H] (and (h_n, c_n) [num_layers*ndir, B, H] if return_state)."""
        self._sync()
        if not x.is_cuda:
            raise _lib.TTSCError('LSTMHip: input must live on a HIP device; no CPU path')
        B, T, _ = x.shape
        H, nd = self.H, self.ndir
        len_dev = None
        if lengths is not None:
            len_dev = _lib.lengths_dev(lengths, x.device)
        hn = torch.empty((self.num_layers * nd, B, H), dtype=torch.float32, device=x.device) if return_state else None
        cn = torch.empty_like(hn) if return_state else None
        cur = x.float().contiguous()
        for l in range(self.num_layers):
            # hoisted input projection of all steps: split-precision MFMA GEMM, padding tiles skipped (the recurrence never reads them)
            if cur.shape[-1] != self._wih[l].shape[1]:
                cur = torch.nn.functional.pad(cur, (0, self._wih[l].shape[1] - cur.shape[-1]))
            xg = linear_hip(cur, self._wih[l], self._bias[l], split=True, lengths_dev=len_dev, period=T)   # [B, T, nd*4H]
            y = torch.empty((B, T, nd * H), dtype=torch.float32, device=x.device)
            h0 = c0 = None
            if hx is not None:
                h0 = hx[0][l * nd:(l + 1) * nd].float().contiguous()
                c0 = hx[1][l * nd:(l + 1) * nd].float().contiguous()
            P = _lib.dev_ptr
            with _lib.on_device(x.device):
                _lib.check(_lib.lib().ttsc_lstm_seq_forward(
                    P(xg), self._whh[l], P(y), P(len_dev) if len_dev is not None else None, B, T, H, nd, nd * H, 0,
                    P(h0) if h0 is not None else None, P(c0) if c0 is not None else None,
                    C.c_void_p(hn[l * nd:(l + 1) * nd].data_ptr()) if hn is not None else None,
                    C.c_void_p(cn[l * nd:(l + 1) * nd].data_ptr()) if cn is not None else None,
                    _lib.current_stream()), 'ttsc_lstm_seq_forward')
            cur = y
        if return_state:
            return cur, (hn, cn)
        return cur
