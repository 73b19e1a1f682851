"""``hifigan.models.Generator`` on MI355X.

Drop-in for the generator class of the reference's un-vendored ``hifigan`` submodule
(constructed at cube/networks/cubegan.py:41-43 and cube/io_utils/runtime.py:49-51; called as
``generator(mel[B,80,T]) -> [B,1,L]`` at cubegan.py:72,83,131 and runtime.py:78).  Same constructor
(``Generator(h)`` with ``h = AttrDict(json)``), same ``state_dict`` key layout (``conv_pre``, ``ups.N``,
``resblocks.N.convs{1,2}.M``, ``conv_post``; each ``weight_g``/``weight_v``/``bias``), same
``remove_weight_norm()``.  forward() runs entirely in the HIP kernels of libttscube_hip.so through
``ttsc_hifigan_forward``; there is no PyTorch/CPU compute path.
"""
import ctypes as C
import os

import torch
import torch.nn as nn

from .. import _lib

LRELU_SLOPE = 0.1


def get_padding(kernel_size, dilation=1):
    return int((kernel_size * dilation - dilation) / 2)


class WNConv(nn.Module):
    """Parameter holder with torch.nn.utils.weight_norm's key layout (weight_g, weight_v, bias).

    ``shape`` is the torch weight shape: Conv1d [Cout,Cin,K], ConvTranspose1d [Cin,Cout,K]; the norm is taken
    over every dim but 0, exactly as weight_norm(dim=0) does for both."""

    def __init__(self, shape, bias_len, init_std=None):
        super().__init__()
        v = torch.empty(shape)
        if init_std is None:
            fan_in = shape[1] * shape[2]
            bound = 1.0 / (fan_in ** 0.5)
            v.uniform_(-bound, bound)
        else:
            v.normal_(0.0, init_std)  # hifigan utils.init_weights: N(0, 0.01)
        self.weight_g = nn.Parameter(v.reshape(shape[0], -1).norm(dim=1).reshape(shape[0], 1, 1).clone())
        self.weight_v = nn.Parameter(v)
        self.bias = nn.Parameter(torch.zeros(bias_len))

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        # accept: folded 'weight' (after remove_weight_norm) and torch>=2.1 parametrization keys
        wk, gk, vk = prefix + 'weight', prefix + 'weight_g', prefix + 'weight_v'
        p0, p1 = prefix + 'parametrizations.weight.original0', prefix + 'parametrizations.weight.original1'
        if p0 in state_dict and p1 in state_dict:
            state_dict[gk] = state_dict.pop(p0)
            state_dict[vk] = state_dict.pop(p1)
        if wk in state_dict and not hasattr(self, 'weight'):
            self.remove_weight_norm()
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs)

    def folded_weight(self):
        if hasattr(self, 'weight'):
            return self.weight.detach()
        v = self.weight_v.detach().float()
        g = self.weight_g.detach().float()
        norm = v.reshape(v.shape[0], -1).norm(dim=1).reshape(-1, 1, 1)
        return g * v / norm

    def remove_weight_norm(self):
        if hasattr(self, 'weight'):
            return
        w = self.folded_weight()
        del self.weight_g
        del self.weight_v
        self.weight = nn.Parameter(w)


class ResBlock1(nn.Module):
    def __init__(self, h, channels, kernel_size=3, dilation=(1, 3, 5)):
        super().__init__()
        self.h = h
        self.convs1 = nn.ModuleList([WNConv((channels, channels, kernel_size), channels, 0.01) for _ in dilation])
        self.convs2 = nn.ModuleList([WNConv((channels, channels, kernel_size), channels, 0.01) for _ in dilation])

    def remove_weight_norm(self):
        for l in list(self.convs1) + list(self.convs2):
            l.remove_weight_norm()


class ResBlock2(nn.Module):
    def __init__(self, h, channels, kernel_size=3, dilation=(1, 3)):
        super().__init__()
        self.h = h
        self.convs = nn.ModuleList([WNConv((channels, channels, kernel_size), channels, 0.01) for _ in dilation])

    def remove_weight_norm(self):
        for l in self.convs:
            l.remove_weight_norm()


class Generator(nn.Module):
    def __init__(self, h):
        super().__init__()
        self.h = h
        self.num_kernels = len(h['resblock_kernel_sizes'])
        self.num_upsamples = len(h['upsample_rates'])
        num_mels = int(h.get('num_mels', 80))
        ch = int(h['upsample_initial_channel'])
        self.conv_pre = WNConv((ch, num_mels, 7), ch)
        rb_cls = ResBlock1 if str(h.get('resblock', '1')) == '1' else ResBlock2
        self.ups = nn.ModuleList()
        self.resblocks = nn.ModuleList()
        for i, (u, k) in enumerate(zip(h['upsample_rates'], h['upsample_kernel_sizes'])):
            cin, cout = ch // (2 ** i), ch // (2 ** (i + 1))
            self.ups.append(WNConv((cin, cout, k), cout, 0.01))
            for k_r, d_r in zip(h['resblock_kernel_sizes'], h['resblock_dilation_sizes']):
                self.resblocks.append(rb_cls(h, cout, k_r, tuple(d_r)))
        self.conv_post = WNConv((1, ch // (2 ** self.num_upsamples), 7), 1, 0.01)
        self._handle = None
        self._sig = None
        self._ws = None
        self._precision = os.environ.get('TTSC_HIFIGAN_PRECISION', 'f16x3')

    def set_precision(self, precision):
        """'f16x3' (default; split-precision fp16 MFMA, fp32 accumulate, ~2^-21 relative per product) or 'fp32'
        (exact fp32 MFMA fmaf chains, 5x slower).  Both meet the 1e-4 RMS parity gate."""
        assert precision in ('fp32', 'f16x3')
        self._precision = precision
        if self._handle is not None:
            _lib.check(_lib.lib().ttsc_hifigan_set_precision(
                self._handle, _lib.PREC_F16X3 if precision == 'f16x3' else _lib.PREC_FP32), 'ttsc_hifigan_set_precision')
        return self

    # ---- C-ABI plumbing ---------------------------------------------------------------------------------
    def _cfg(self):
        h = self.h
        c = _lib.HifiganCfg()
        c.num_mels = int(h.get('num_mels', 80))
        c.upsample_initial_channel = int(h['upsample_initial_channel'])
        c.resblock = 1 if str(h.get('resblock', '1')) == '1' else 2
        c.num_upsamples = self.num_upsamples
        for i, (u, k) in enumerate(zip(h['upsample_rates'], h['upsample_kernel_sizes'])):
            c.upsample_rates[i] = int(u)
            c.upsample_kernel_sizes[i] = int(k)
        c.num_kernels = self.num_kernels
        for j, (k_r, d_r) in enumerate(zip(h['resblock_kernel_sizes'], h['resblock_dilation_sizes'])):
            c.resblock_kernel_sizes[j] = int(k_r)
            c.num_dilations[j] = len(d_r)
            for m, d in enumerate(d_r):
                c.resblock_dilation_sizes[j][m] = int(d)
        return c

    def _named_convs(self):
        yield 'conv_pre', self.conv_pre
        for i, l in enumerate(self.ups):
            yield 'ups.%d' % i, l
        for n, rb in enumerate(self.resblocks):
            if isinstance(rb, ResBlock1):
                for m, l in enumerate(rb.convs1):
                    yield 'resblocks.%d.convs1.%d' % (n, m), l
                for m, l in enumerate(rb.convs2):
                    yield 'resblocks.%d.convs2.%d' % (n, m), l
            else:
                for m, l in enumerate(rb.convs):
                    yield 'resblocks.%d.convs.%d' % (n, m), l
        yield 'conv_post', self.conv_post

    def _sync(self):
        """(Re)upload folded weights to the HIP handle when any parameter changed."""
        L = _lib.lib()
        # (the Parameter objects of the module are fixed after construction — remove_weight_norm() drops this list — so the module tree is walked
        # once: the walk was 170 us of host time in front of every forward, the GPU idle behind the conditioning at B = 1)
        pl = self.__dict__.get('_plist')
        if pl is None:
            pl = list(self.parameters())
            object.__setattr__(self, '_plist', pl)
        sig = tuple((p.data_ptr(), p._version) for p in pl)
        if self._handle is not None and sig == self._sig:
            return
        if self._handle is None:
            _lib.require_gpu()
            hnd = C.c_void_p()
            cfg = self._cfg()
            _lib.check(L.ttsc_hifigan_create(C.byref(cfg), C.byref(hnd)), 'ttsc_hifigan_create')
            self._handle = hnd
            self._branch_dev = None
            self.set_precision(self._precision)
        for name, l in self._named_convs():
            for suffix, t in (('.weight', l.folded_weight()), ('.bias', l.bias.detach())):
                t = t.float().cpu().contiguous()
                shape = (C.c_int64 * t.dim())(*t.shape)
                _lib.check(L.ttsc_hifigan_set_weight(self._handle, (name + suffix).encode(), C.c_void_p(t.data_ptr()),
                                                     shape, t.dim()), 'ttsc_hifigan_set_weight(%s)' % (name + suffix))
        self._sig = sig
        pend, self._pending_scales = getattr(self, '_pending_scales', None), None
        if pend is not None and self._precision == 'f16x3' and abs(pend['fingerprint'] - self.weight_fingerprint()) <= 1e-9 * abs(pend['fingerprint']):
            self.load_activation_scales(pend['scales'])    # persisted beside the checkpoint these weights came from

    def weight_fingerprint(self):
        """sum |w| over the folded weights (float64): ties a persisted scale file to the weights it was calibrated for"""
        with torch.no_grad():
            return float(sum(l.folded_weight().double().abs().sum() for _, l in self._named_convs()))

    def export_scales(self):
        """JSON-able record of the split-precision calibration (None unless the f16x3 handle exists and is calibrated)"""
        if self._handle is None or self._precision != 'f16x3':
            return None
        return {'precision': 'f16x3', 'fingerprint': self.weight_fingerprint(), 'scales': self.activation_scales()}

    def import_scales(self, rec):
        """queue a record written by export_scales(); applied at the next weight upload if the fingerprint matches"""
        if rec and rec.get('precision') == 'f16x3':
            self._pending_scales = rec
            self._sig = None

    def __del__(self):
        try:
            if self._handle is not None:
                _lib.lib().ttsc_hifigan_destroy(self._handle)
        except Exception:
            pass

    def out_len(self, T):
        L = T
        for u, k in zip(self.h['upsample_rates'], self.h['upsample_kernel_sizes']):
            L = (L - 1) * u - 2 * ((k - u) // 2) + k
        return L

    def algorithmic_flops(self, B, T):
        self._sync()
        out = C.c_double()
        _lib.check(_lib.lib().ttsc_hifigan_algorithmic_flops(self._handle, B, T, C.byref(out)), 'algorithmic_flops')
        return out.value

    def forward(self, x, frames=None, check='sync'):
        """x: [B, num_mels, T] fp32 on a HIP device -> [B, 1, L].  frames (optional, list[int] per utterance): valid mel
        frames of a padded batch — every layer then masks beyond the utterance's own length, so
        y[b, 0, :out_len(frames[b])] equals that utterance run alone (ragged batching; not in the reference, B=1).
        check: the split-precision range guard — 'sync' (default): this call waits for its own result and, if a non-finite sample
        came out, re-calibrates and reruns by itself; 'deferred': no wait — the previous deferred call's verdict is collected at
        the start of this one (raises if it was bad), the last one's by `finish_range_check()`: for pipelined callers whose host
        work for the next batch overlaps the generator."""
        if not x.is_cuda:
            raise _lib.TTSCError('Generator.forward: input must live on a HIP device (got %s); no CPU path' % x.device)
        # differentiable path only when a gradient is actually wanted: the input carries one, or the module is in training mode.
        # (An eval-mode call made without no_grad() must run the same kernels and precision as inference — ADVICE r1.)
        if torch.is_grad_enabled() and (x.requires_grad or (self.training and any(p.requires_grad for p in self.parameters()))):
            from .autograd import generator_forward_with_grad
            return generator_forward_with_grad(self, x)
        return self._forward_hip(x, frames, check)

    def range_check_pending(self):
        """True while a forward run with check='deferred' has not had its verdict collected"""
        return self._handle is not None and bool(getattr(self, '_deferred_pending', False))

    def finish_range_check(self, raise_on_trip=True):
        """collect the verdict of the deferred range guard (synchronises the current stream); raises if a forward since the last
        collection emitted non-finite audio (raise_on_trip=False: returns the verdict instead — for clean-up paths that must not raise)"""
        if not self.range_check_pending():
            return False
        self._deferred_pending = False
        st = int(_lib.lib().ttsc_hifigan_range_status(self._handle, _lib.current_stream()))
        if st < 0 and raise_on_trip:
            _lib.check(st, 'ttsc_hifigan_range_status')
        if not raise_on_trip:
            return st != 0
        if st:
            raise _lib.TTSCError('Generator: a forward run with check="deferred" produced non-finite audio (an activation left the fp16 range its '
                                 'pre-scale was calibrated for, or the input was non-finite); rerun that batch with check="sync"')

    def _forward_hip(self, x, frames=None, check='sync'):
        L = _lib.lib()
        self._sync()
        assert check in ('sync', 'deferred', None)
        with _lib.on_device(x.device):
            self.finish_range_check()             # (verdict of the previous deferred call, before anything new is enqueued)
        mode = {'sync': 1, 'deferred': 2, None: 0}[check]
        if mode != getattr(self, '_range_mode', 1):
            _lib.check(L.ttsc_hifigan_set_range_check(self._handle, mode), 'ttsc_hifigan_set_range_check')
            self._range_mode = mode
        self._deferred_pending = mode == 2
        x = x.detach().float().contiguous()
        B, _, T = x.shape
        Lout = self.out_len(T)
        need = L.ttsc_hifigan_workspace_bytes(self._handle, B, T)
        if self._ws is None or self._ws.numel() * 4 < need or self._ws.device != x.device:
            self._ws = None
            self._ws = torch.empty((need + 3) // 4, dtype=torch.float32, device=x.device)
        y = torch.empty((B, 1, Lout), dtype=torch.float32, device=x.device)
        with _lib.on_device(x.device):
            if getattr(self, '_branch_dev', None) != x.device and os.environ.get('TTSC_BRANCH_STREAMS_OWN', '0') != '1':   # (1: measurement switch)
                # the branch schedule borrows two of the package's reserved streams instead of creating its own (see streams._reserve)
                from .streams import branch_stream_handles
                sa, sb = branch_stream_handles(x.device)
                _lib.check(L.ttsc_hifigan_set_branch_streams(self._handle, C.c_void_p(sa), C.c_void_p(sb)), 'ttsc_hifigan_set_branch_streams')
                self._branch_dev = x.device
            fr = None
            if frames is not None:
                assert len(frames) == B
                fr = (C.c_int32 * B)(*[int(f) for f in frames])
            _lib.check(L.ttsc_hifigan_forward_ragged(self._handle, _lib.dev_ptr(x), B, T, fr, _lib.dev_ptr(y),
                                                     _lib.dev_ptr(self._ws), self._ws.numel() * 4, _lib.current_stream()),
                       'ttsc_hifigan_forward')
        return y

    def calibrate(self, x):
        """(Re)derive the split-precision path's per-layer activation pre-scales from `x` [B, num_mels, T] (ttsc_hifigan_calibrate:
        one layer-by-layer forward with an abs-max reduction per layer).  The first forward after loading weights calibrates
        by itself on a fixed built-in probe mel (so the scales depend on the weights alone); call this to calibrate on other
        data.  Returns that forward's output."""
        L = _lib.lib()
        self._sync()
        x = x.detach().float().contiguous()
        B, _, T = x.shape
        need = L.ttsc_hifigan_workspace_bytes(self._handle, B, T)
        if self._ws is None or self._ws.numel() * 4 < need or self._ws.device != x.device:
            self._ws = None
            self._ws = torch.empty((need + 3) // 4, dtype=torch.float32, device=x.device)
        y = torch.empty((B, 1, self.out_len(T)), dtype=torch.float32, device=x.device)
        with _lib.on_device(x.device):
            _lib.check(L.ttsc_hifigan_calibrate(self._handle, _lib.dev_ptr(x), B, T, _lib.dev_ptr(y), _lib.dev_ptr(self._ws),
                                                self._ws.numel() * 4, _lib.current_stream()), 'ttsc_hifigan_calibrate')
        return y

    def activation_scale(self, layer_name):
        """power-of-two pre-scale currently applied to the input of `layer_name` (e.g. 'resblocks.3.convs1.0')"""
        out = C.c_float()
        _lib.check(_lib.lib().ttsc_hifigan_get_activation_scale(self._handle, layer_name.encode(), C.byref(out)), 'get_activation_scale')
        return out.value

    def activation_scales(self):
        """{layer name: power-of-two pre-scale} of the split-precision path — what `Cubegan.save` keeps beside a checkpoint"""
        self._sync()
        return {name: self.activation_scale(name) for name, _ in self._named_convs()}

    def load_activation_scales(self, scales):
        """restore persisted scales (all layers); the handle then skips its own calibration until the weights change"""
        self._sync()
        names = [n for n, _ in self._named_convs()]
        arr_n = (C.c_char_p * len(names))(*[n.encode() for n in names])
        arr_s = (C.c_float * len(names))(*[float(scales[n]) for n in names])
        _lib.check(_lib.lib().ttsc_hifigan_set_activation_scales(self._handle, arr_n, arr_s, len(names)), 'ttsc_hifigan_set_activation_scales')

    @property
    def recalibrations(self):
        """forwards that tripped the range guard (a non-finite sample left conv_post) and were rerun after re-calibration"""
        return 0 if self._handle is None else int(_lib.lib().ttsc_hifigan_recalibrations(self._handle))

    def remove_weight_norm(self):
        object.__setattr__(self, '_plist', None)     # (the Parameter objects change: _sync walks the module tree again)
        self.conv_pre.remove_weight_norm()
        for l in self.ups:
            l.remove_weight_norm()
        for rb in self.resblocks:
            rb.remove_weight_norm()
        self.conv_post.remove_weight_norm()
        self._sig = None
