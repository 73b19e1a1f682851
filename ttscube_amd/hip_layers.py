"""Thin Python handles over the C-ABI layer objects of libttscube_hip.so (no compute in Python).

``Conv1dHip`` stands where the reference uses torch.nn.Conv1d / ConvTranspose1d inside ConvNorm
(cube/networks/modules.py:37-55), PostNet (modules.py:117-145), the WaveRNN low-res convs (modules.py:416-420)
and the char CNNs (textcoder.py:44-53, modules.py:850-871)."""
import ctypes as C

import torch

from . import _lib

ACT = {None: _lib.ACT_NONE, 'none': _lib.ACT_NONE, 'tanh': _lib.ACT_TANH, 'relu': _lib.ACT_RELU,
       'sigmoid': _lib.ACT_SIGMOID}


class Conv1dHip:
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, transposed=False):
        L = _lib.lib()
        _lib.require_gpu()
        self.cfg = _lib.Conv1dCfg(in_channels, out_channels, kernel_size, stride, padding, dilation, int(transposed))
        self._h = C.c_void_p()
        _lib.check(L.ttsc_conv1d_create(C.byref(self.cfg), C.byref(self._h)), 'ttsc_conv1d_create')

    def set_weight(self, weight, bias=None):
        w = weight.detach().float().cpu().contiguous()
        exp = ((self.cfg.in_channels, self.cfg.out_channels) if self.cfg.transposed else
               (self.cfg.out_channels, self.cfg.in_channels)) + (self.cfg.kernel_size,)
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

    def out_len(self, Lin):
        return int(_lib.lib().ttsc_conv1d_out_len(self._h, Lin))

    def __call__(self, x, resid=None, out=None, in_scale=1.0, in_slope=1.0, out_scale=1.0, act=None, accumulate=False):
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
        ep = _lib.Conv1dEpilogue(in_scale, in_slope, out_scale, ACT[act], int(accumulate))
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().ttsc_conv1d_forward(self._h, _lib.dev_ptr(x), B, Lin, _lib.dev_ptr(out),
                                                      _lib.dev_ptr(resid) if resid is not None else None,
                                                      C.byref(ep), _lib.current_stream()), 'ttsc_conv1d_forward')
        return out

    def __del__(self):
        try:
            _lib.lib().ttsc_conv1d_destroy(self._h)
        except Exception:
            pass
