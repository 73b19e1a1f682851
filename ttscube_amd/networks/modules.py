"""Mirror of cube/networks/modules.py for the hot path.  Classes keep the reference constructor signatures and
``state_dict`` key layouts (captured in SURVEY.md §8b); torch ``nn.GRU/nn.LSTM/nn.Conv1d/nn.Linear`` objects are used
ONLY as parameter containers (identical keys, shapes and default init) — their forward() is never called: all compute
goes through the C ABI of libttscube_hip.so."""
import ctypes as C

import numpy as np
import torch
import torch.nn as nn

from .. import _lib
from .loss import MULAWOutput, RAWOutput


class LinearNorm(nn.Module):
    """cube/networks/modules.py:24-34 (parameter container)."""

    def __init__(self, in_dim, out_dim, bias=True, w_init_gain='linear'):
        super().__init__()
        self.linear_layer = nn.Linear(in_dim, out_dim, bias=bias)
        nn.init.xavier_normal_(self.linear_layer.weight, gain=nn.init.calculate_gain(w_init_gain))


class ConvNorm(nn.Module):
    """cube/networks/modules.py:37-55 (parameter container)."""

    def __init__(self, in_channels, out_channels, kernel_size=1, stride=1, padding=None, dilation=1, bias=True,
                 w_init_gain='linear'):
        super().__init__()
        if padding is None:
            assert (kernel_size % 2 == 1)
            padding = int(dilation * (kernel_size - 1) / 2)
        self.conv = nn.Conv1d(in_channels, out_channels, kernel_size=kernel_size, stride=stride, padding=padding,
                              dilation=dilation, bias=bias)
        nn.init.xavier_normal_(self.conv.weight, gain=nn.init.calculate_gain(w_init_gain))


def _param_signature(module):
    return tuple((p.data_ptr(), p._version) for p in module.parameters())


class WaveRNN(nn.Module):
    """cube/networks/modules.py:392-589.  ``forward({'mel', ['x_low']})`` decodes with the persistent HIP kernel
    (``ttsc_wavernn_decode``) and returns a numpy array [B, L, 1] of decoded samples, like the reference."""

    def __init__(self, num_layers: int = 2, layer_size: int = 512, upsample=100, upsample_low=10, use_lowres=True,
                 learning_rate=1e-4, output='mol'):
        super().__init__()
        self._learning_rate = learning_rate
        self._use_lowres = use_lowres
        self._upsample = upsample
        self._upsample_low = upsample_low
        self._num_layers = num_layers
        self._layer_size = layer_size
        if self._use_lowres:
            self._lowres_conv = nn.ModuleList()
            ic = 1
            for ii in range(3):
                self._lowres_conv.append(ConvNorm(ic, 20, kernel_size=7, padding=3))
                ic = 20
        ic = 80 + 1
        if use_lowres:
            ic += 21
        self._skip = LinearNorm(ic, layer_size, w_init_gain='tanh')  # dead in the reference too (modules.py:424)
        rnn_list = []
        for ii in range(num_layers):
            rnn_list.append(nn.GRU(input_size=ic, hidden_size=layer_size, num_layers=1, batch_first=True))
            ic = layer_size
        self._rnns = nn.ModuleList(rnn_list)
        self._preoutput = LinearNorm(layer_size, 256)
        if output == 'mulaw':
            self._output_functions = MULAWOutput()
        elif output == 'raw':
            self._output_functions = RAWOutput()
        else:
            raise NotImplementedError("output='%s': the HIP sampler implements the discrete outputs 'mulaw' and 'raw' "
                                      "(cube/networks/loss.py:218-307)" % output)
        self._output_name = output
        self._output = LinearNorm(256, self._output_functions.sample_size, w_init_gain='linear')
        self._val_loss = 9999
        self._handle = None
        self._sig = None
        self._ws = None

    # ---- C-ABI plumbing ---------------------------------------------------------------------------------
    def _sync(self):
        L = _lib.lib()
        sig = _param_signature(self)
        if self._handle is not None and sig == self._sig:
            return
        if self._handle is None:
            _lib.require_gpu()
            cfg = _lib.WavernnCfg(self._layer_size, self._num_layers, int(self._use_lowres), self._upsample,
                                  self._upsample_low, self._output_functions.sample_size, 80,
                                  _lib.WR_OUT_MULAW if self._output_name == 'mulaw' else _lib.WR_OUT_RAW)
            hnd = C.c_void_p()
            _lib.check(L.ttsc_wavernn_create(C.byref(cfg), C.byref(hnd)), 'ttsc_wavernn_create')
            self._handle = hnd
        for name, p in self.state_dict().items():
            t = p.detach().float().cpu().contiguous()
            shape = (C.c_int64 * t.dim())(*t.shape)
            _lib.check(L.ttsc_wavernn_set_weight(self._handle, name.encode(), C.c_void_p(t.data_ptr()), shape, t.dim()),
                       'ttsc_wavernn_set_weight(%s)' % name)
        self._sig = sig

    def __del__(self):
        try:
            if self._handle is not None:
                _lib.lib().ttsc_wavernn_destroy(self._handle)
        except Exception:
            pass

    def decode(self, X, mode='philox', noise=None, seed=None, forced_x=None, want_logits=False):
        """Device-side decode.  Returns (idx uint8 [B,L], wav fp32 [B,L], logits fp32 [B,L,S] | None) as device tensors.

        mode: 'philox' (in-kernel counter RNG, seeded from torch's generator unless `seed` is given), 'noise'
        (injected Gumbel noise [B,L,S], used by the parity tests) or 'argmax'."""
        L = _lib.lib()
        self._sync()
        dev = self._get_device()
        mel = X['mel'].to(dev).float().contiguous()
        B, T, _ = mel.shape
        x_low, Tl = None, 0
        if self._use_lowres:
            x_low = X['x_low'].to(dev).float().contiguous()
            Tl = x_low.shape[1]
        Lout = int(L.ttsc_wavernn_out_len(self._handle, T, Tl))
        S = self._output_functions.sample_size
        idx = torch.empty((B, Lout), dtype=torch.uint8, device=dev)
        wav = torch.empty((B, Lout), dtype=torch.float32, device=dev)
        logits = torch.empty((B, Lout, S), dtype=torch.float32, device=dev) if want_logits else None
        m = {'argmax': _lib.WR_MODE_ARGMAX, 'noise': _lib.WR_MODE_NOISE, 'philox': _lib.WR_MODE_PHILOX}[mode]
        nz = None
        if mode == 'noise':
            if noise is None:
                raise _lib.TTSCError("WaveRNN.decode: mode='noise' needs a noise tensor [B, L, S]")
            nz = torch.as_tensor(noise).to(dev).float().contiguous()
            assert tuple(nz.shape) == (B, Lout, S), (tuple(nz.shape), (B, Lout, S))
        if seed is None:
            seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if mode == 'philox' else 0
        fx = None
        if forced_x is not None:
            fx = torch.as_tensor(forced_x).to(dev).float()[:, :Lout].contiguous()
            assert fx.shape[1] == Lout
        need = L.ttsc_wavernn_workspace_bytes(self._handle, B, T, Tl)
        if self._ws is None or self._ws.numel() * 4 < need or self._ws.device != mel.device:
            self._ws = torch.empty((need + 3) // 4, dtype=torch.float32, device=dev)
        P = _lib.dev_ptr
        with torch.cuda.device(mel.device):
            _lib.check(L.ttsc_wavernn_decode(self._handle, P(mel), P(x_low) if x_low is not None else None, B, T, Tl, m,
                                             P(nz) if nz is not None else None, C.c_uint64(seed),
                                             P(fx) if fx is not None else None, P(idx), P(wav),
                                             P(logits) if logits is not None else None, P(self._ws), self._ws.numel() * 4,
                                             _lib.current_stream()), 'ttsc_wavernn_decode')
        return idx, wav, logits

    def forward(self, X):
        if 'x' in X:
            return self._train_forward(X)
        return self._inference(X)

    def _inference(self, X, **kw):
        with torch.no_grad():
            _, wav, _ = self.decode(X, **kw)
        return wav.unsqueeze(2).detach().cpu().numpy()  # [B, L, 1] like modules.py:499-503

    def _train_forward(self, X):
        """Teacher-forced logits [B, L, S] (modules.py:505-539): the decode kernel with the feedback forced to X['x']
        shifted as the caller prepared it.  No autograd here — see networks/training.py for the training step."""
        x = X['x']
        # reference input at step t is X['x'][:, t]; the kernel feeds forced_x[t] back at step t+1
        fx = torch.cat([x[:, 1:], x[:, -1:]], dim=1)
        if not torch.equal(x[:, :1], torch.zeros_like(x[:, :1])):
            raise _lib.TTSCError('WaveRNN._train_forward: X["x"] must be the target shifted right by one with a leading 0 '
                                 '(modules.py:555-558)')
        _, _, logits = self.decode({k: v for k, v in X.items() if k != 'x'}, mode='argmax', forced_x=fx, want_logits=True)
        return logits

    @torch.jit.ignore
    def _get_device(self):
        p = self._output.linear_layer.weight
        if p.device.type == 'cpu':
            raise _lib.TTSCError('WaveRNN: parameters live on the CPU; move the module to a HIP device (no CPU path)')
        return p.device

    @torch.jit.ignore
    def save(self, path):
        torch.save(self.state_dict(), path)

    @torch.jit.ignore
    def load(self, path):
        self.load_state_dict(torch.load(path, map_location='cpu'))
