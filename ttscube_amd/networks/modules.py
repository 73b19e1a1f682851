"""Mirror of cube/networks/modules.py for the hot path.  Classes keep the reference constructor signatures and
``state_dict`` key layouts (captured in SURVEY.md §8b); torch ``nn.GRU/nn.LSTM/nn.Conv1d/nn.Linear`` objects are used
ONLY as parameter containers (identical keys, shapes and default init) — their forward() is never called: all compute
goes through the C ABI of libttscube_hip.so."""
import ctypes as C

import numpy as np
import os

import torch
import torch.nn as nn

from .. import _lib
from .loss import BetaOutput, GaussianOutput, MOLOutput, MULAWOutput, RAWOutput


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
        outs = {'mol': MOLOutput, 'gm': GaussianOutput, 'beta': BetaOutput, 'mulaw': MULAWOutput, 'raw': RAWOutput}
        if output not in outs:
            raise ValueError("output must be one of %s (cube/networks/modules.py:429-438), got %r" % (sorted(outs), output))
        self._output_functions = outs[output]()
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
                                  self._upsample_low, self._output_functions.sample_size, 80, self._output_functions.kind)
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

        mode: 'philox' (in-kernel counter RNG, seeded from torch's generator unless `seed` is given), 'noise' (injected noise
        [B, L, output_functions.noise_width]: Gumbel terms for mulaw/raw, the sampler's random terms for mol/gm/beta — used by the
        parity tests) or 'argmax' (no noise: the arg-max class / the mode of the selected component).  idx holds the class index
        (mulaw/raw), the mixture index (mol) or zeros."""
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
                raise _lib.TTSCError("WaveRNN.decode: mode='noise' needs a noise tensor [B, L, %d]" % self._output_functions.noise_width)
            nz = torch.as_tensor(noise).to(dev).float().contiguous()
            W = self._output_functions.noise_width
            assert tuple(nz.shape) == (B, Lout, W), (tuple(nz.shape), (B, Lout, W))
        if seed is None:
            seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if mode == 'philox' else 0
        fx = None
        if forced_x is not None:
            fx = torch.as_tensor(forced_x).to(dev).float()[:, :Lout]
            if fx.shape[1] < Lout:   # shorter target than conditioning: the net is causal, pad the tail (caller slices)
                fx = torch.nn.functional.pad(fx, (0, Lout - fx.shape[1]))
            fx = fx.contiguous()
        need = L.ttsc_wavernn_workspace_bytes(self._handle, B, T, Tl)
        if self._ws is None or self._ws.numel() * 4 < need or self._ws.device != mel.device:
            self._ws = torch.empty((need + 3) // 4, dtype=torch.float32, device=dev)
        P = _lib.dev_ptr
        with _lib.on_device(mel.device):
            _lib.check(L.ttsc_wavernn_decode(self._handle, P(mel), P(x_low) if x_low is not None else None, B, T, Tl, m,
                                             P(nz) if nz is not None else None, C.c_uint64(seed),
                                             P(fx) if fx is not None else None, P(idx), P(wav),
                                             P(logits) if logits is not None else None, P(self._ws), self._ws.numel() * 4,
                                             _lib.current_stream()), 'ttsc_wavernn_decode')
            st = L.ttsc_wavernn_last_status(self._handle, _lib.current_stream())
            if st == 1:
                raise _lib.TTSCError('WaveRNN multi-workgroup kernel aborted on a hand-off timeout: ' + L.ttsc_last_error().decode())
            self.last_kernel = {2: 'tile'}.get(st, 'stream')
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
        return logits[:, :x.shape[1]]   # msize = min(conditioning, target) as modules.py:519-522

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


# =====================================================================================================================
# Mel decoders
# =====================================================================================================================
from ..hip_layers import Conv1dHip, LSTMHip, linear_hip  # noqa: E402


class _ConvStack:
    """Lazily synced Conv1dHip handles for a list of (conv_module, bn_module|None); BatchNorm1d (eval) is folded into
    the conv weights at sync time: w' = w * g / sqrt(var + eps), b' = (b - mean) * g / sqrt(var + eps) + beta."""

    def __init__(self, items):
        self.items = items
        self._sig = None
        self.h = None

    def sync(self):
        ts = getattr(self, '_tensors', None)
        if ts is None:      # (the modules' parameter / buffer objects are fixed: walk them once)
            mods = [m for it in self.items for m in it if m is not None]
            ts = self._tensors = [t for m in mods for t in list(m.parameters()) + list(m.buffers())]
        sig = tuple((t.data_ptr(), t._version) for t in ts)
        if sig == self._sig:
            return self.h
        self.h = []
        for conv, bn in self.items:
            w, b = conv.weight.detach().float().cpu(), conv.bias.detach().float().cpu()
            if bn is not None:
                s = bn.weight.detach().float().cpu() / torch.sqrt(bn.running_var.detach().float().cpu() + bn.eps)
                w = w * s[:, None, None]
                b = (b - bn.running_mean.detach().float().cpu()) * s + bn.bias.detach().float().cpu()
            c = Conv1dHip(conv.in_channels, conv.out_channels, conv.kernel_size[0], padding=conv.padding[0],
                          dilation=conv.dilation[0])
            c.set_weight(w, b)
            self.h.append(c)
        self._sig = sig
        return self.h


class PostNet(nn.Module):
    """cube/networks/modules.py:117-145 (same nn.Sequential indices => same state_dict keys).  Inference-mode
    forward on the HIP conv kernel: BatchNorm folded, tanh fused, dropout inactive (eval)."""

    def __init__(self, num_mels=80, kernel_size=5, filter_size=512, output_size=None):
        super().__init__()
        if output_size is None:
            output_size = num_mels
        layers = []
        ic = num_mels
        for i in range(4):
            layers += [ConvNorm(ic, filter_size, kernel_size, padding=kernel_size // 2, w_init_gain='tanh'),
                       nn.BatchNorm1d(512), nn.Tanh(), nn.Dropout(0.1)]
            ic = filter_size
        layers.append(ConvNorm(filter_size, output_size, kernel_size, padding=kernel_size // 2, w_init_gain='linear'))
        self.network = nn.Sequential(*layers)
        self._stack = _ConvStack([(self.network[0].conv, self.network[1]), (self.network[4].conv, self.network[5]),
                                  (self.network[8].conv, self.network[9]), (self.network[12].conv, self.network[13]),
                                  (self.network[16].conv, None)])

    def forward(self, x, add_residual=False):
        """x [B, F, 80] -> [B, F, 80]; add_residual=True returns x + postnet(x) (textcoder.py:186-187) in one epilogue."""
        if self.training:
            raise _lib.TTSCError('PostNet: the HIP path implements eval-mode BatchNorm/Dropout (call .eval())')
        hs = self._stack.sync()
        xc = x.float().permute(0, 2, 1).contiguous()
        h = xc
        for c in hs[:-1]:
            h = c(h, act='tanh')
        y = hs[-1](h, resid=xc if add_residual else None)
        return y.permute(0, 2, 1).contiguous()


class PreNet(nn.Module):
    """cube/networks/modules.py:148-164: 2 x [Linear -> ReLU -> dropout(p=0.5, ALWAYS on)]."""

    def __init__(self, num_mels=80, hidden=256, layers=2):
        super().__init__()
        mods = []
        inp = num_mels
        for _ in range(layers):
            mods.append(LinearNorm(inp, hidden, w_init_gain='linear'))
            inp = hidden
        self.layers_h = nn.ModuleList(mods)

    def forward(self, x, masks=None):
        """masks: optional list of {0,1} tensors (one per layer, broadcastable to the layer output) for parity tests;
        by default Bernoulli(0.5) masks are drawn from torch's device generator."""
        h = x
        for i, layer in enumerate(self.layers_h):
            h = linear_hip(h, layer.linear_layer.weight, layer.linear_layer.bias, act='relu')
            m = masks[i] if masks is not None else (torch.rand(h.shape, device=h.device) >= 0.5).float()
            h = h * (m.to(h.device) * 2.0)
        return h


def _cnn_forward(stack, emb, lengths):
    """3 x tanh(Conv1d k3) over [B, N, C]; positions >= length are zeroed between layers so that a padded batch
    reproduces the per-utterance (zero-padded conv) results."""
    hs = stack.sync()
    B, N, _ = emb.shape
    mask = None
    if lengths is not None and B > 1:
        mask = (torch.arange(N, device=emb.device)[None, :] < _lib.lengths_dev(lengths, emb.device)[:, None]).float()[:, None, :]
    h = emb.float().permute(0, 2, 1).contiguous()
    if mask is not None:
        h = h * mask
    for c in hs:
        h = c(h, act='tanh')
        if mask is not None:
            h = h * mask
    return h.permute(0, 2, 1).contiguous()


class Alignment:
    """Frame -> phone map built ON THE DEVICE from a duration head (csrc/align.hip): `f2p` int32 [B, Fcap] (phone index of
    every frame, valid up to `flens[b]`), `durs` int32 [B, N].  Behaves like the reference's list of per-utterance lists
    (`X['y_frame2phone']`, modules.py:946-953) when indexed / compared — the host copy is made only then."""

    def __init__(self, f2p, flen_dev, flens, durs):
        self.f2p, self.flen_dev, self.flens, self.durs = f2p, flen_dev, flens, durs
        self._lists = None

    def tolist(self):
        if self._lists is None:
            m = max(self.flens) if self.flens else 0
            host = self.f2p[:, :m].cpu().tolist() if m else [[] for _ in self.flens]
            self._lists = [row[:n] for row, n in zip(host, self.flens)]
        return self._lists

    def durations(self):
        return self.durs.cpu().numpy()

    def __len__(self):
        return len(self.flens)

    def __getitem__(self, i):
        return self.tolist()[i]

    def __iter__(self):
        return iter(self.tolist())

    def __eq__(self, other):
        return self.tolist() == (other.tolist() if isinstance(other, Alignment) else other)


def _char_lengths(X, x_char):
    """Valid phones per utterance of a padded batch.  Preferred: the collate's own count X['x_len'] — id 0 is BOTH the padding
    and an out-of-vocabulary phone (io_cubegan.py:196-199), so counting non-zeros would cut a sentence short at an OOV phone.
    Batches made by the reference's collate carry no x_len: fall back to the position of the last non-zero id."""
    B, N = x_char.shape
    if X.get('x_len') is not None:
        return [int(v) for v in torch.as_tensor(X['x_len']).reshape(-1).tolist()]
    if B == 1:
        return [N]
    pos = torch.arange(1, N + 1, device=x_char.device)[None, :]
    return ((x_char != 0).long() * pos).max(dim=1).values.tolist()


def align_durations(out_dur, lengths):
    """argmax over the duration head [B, N, D], exclusive scan and scatter of phone indices, all in one kernel; the host
    reads back B frame counts (it needs them to size the expanded tensors) and nothing else."""
    out_dur = out_dur.float().contiguous()
    B, N, D = out_dur.shape
    dev = out_dur.device
    fcap = max(1, N * (D - 1))
    durs = torch.empty((B, N), dtype=torch.int32, device=dev)
    f2p = torch.empty((B, fcap), dtype=torch.int32, device=dev)
    flen = torch.empty((B,), dtype=torch.int32, device=dev)
    len_t = _lib.lengths_dev(lengths, dev)
    with _lib.on_device(dev):
        _lib.check(_lib.lib().ttsc_align_durations(_lib.dev_ptr(out_dur), _lib.dev_ptr(len_t) if len_t is not None else None, B, N, D,
                                                   _lib.dev_ptr(durs), _lib.dev_ptr(f2p), _lib.dev_ptr(flen), fcap, _lib.current_stream()),
                   'ttsc_align_durations')
    return Alignment(f2p, flen, _lib.DevLengths(flen.cpu().tolist(), dev_tensor=flen), durs)   # (frame counts: host list + the device copy the kernel wrote)


def _expand_rows(x, alignments, stride=1):
    """Gather rows of x [B, N, C] by per-utterance frame->phone alignments (every `stride`-th frame), padding short
    utterances with their last aligned row (Languasito2._expand_i modules.py:1043-1053 / Textcoder._expand 291-302).
    `alignments`: an Alignment (device-side map: one gather kernel, no host data) or a list of lists (teacher forcing with the
    alignments a collate supplies — host data to begin with)."""
    if isinstance(alignments, Alignment):
        al = alignments
        flens = al.flens if stride == 1 else [n // stride for n in al.flens]   # (stride 1 keeps the device copy of the counts)
        m = max(flens) if flens else 0
        if m == 0:
            return x[:, :0], [0] * len(flens)
        x = x.float().contiguous()
        B, N, C_ = x.shape
        out = torch.empty((B, m, C_), dtype=torch.float32, device=x.device)
        with _lib.on_device(x.device):
            _lib.check(_lib.lib().ttsc_expand_rows(_lib.dev_ptr(x), _lib.dev_ptr(al.f2p), _lib.dev_ptr(al.flen_dev), B, N, C_, al.f2p.shape[1],
                                                   stride, m, _lib.dev_ptr(out), _lib.current_stream()), 'ttsc_expand_rows')
        return out, flens
    sel = [[a[j * stride] for j in range(len(a) // stride)] for a in alignments]
    m = max(len(s) for s in sel) if sel else 0
    if m == 0:
        return x[:, :0], [0] * len(sel)
    idx = np.zeros((len(sel), m), dtype=np.int64)
    for b, s in enumerate(sel):
        if len(s):
            idx[b, :len(s)] = s
            idx[b, len(s):] = s[-1] if stride == 1 else x.shape[1] - 1
    idx_t = torch.from_numpy(idx).to(x.device)
    return torch.gather(x, 1, idx_t[:, :, None].expand(-1, -1, x.shape[2])).contiguous(), [len(s) for s in sel]


G_STREAM = os.environ.get('TTSC_LANG_G_STREAM', '1') != '0'          # (measurement switch: 0 = the `g` stack after the pitch recurrence, on the same stream)
COND_INPUT_FUSED = os.environ.get('TTSC_COND_INPUT_FUSED', '1') != '0'   # (measurement / test switch: 0 = the elementwise formulation)
UPLOAD_RING = os.environ.get('TTSC_UPLOAD_RING', '1') != '0'             # (measurement switch: 0 = three separate uploads)
G_STREAM_MAX_B = int(os.environ.get('TTSC_LANG_G_STREAM_MAX_B', '8'))
_G_STREAMS = {}


def _g_stream(dev):
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    if key not in _G_STREAMS:
        from ..hifigan.streams import _side_streams    # (one of the package's reserved streams — no stream of its own: hifigan/streams.py::_reserve)
        _G_STREAMS[key] = _side_streams(dev, 1)[0]
    return _G_STREAMS[key]


class Languasito2(nn.Module):
    """cube/networks/modules.py:805-1094 — text -> 80-d per-frame conditioning for the HiFi-GAN generator.
    Same constructor and state_dict keys; `inference` runs on the HIP conv / GEMM / LSTM kernels and additionally
    accepts padded batches (x_char padded with 0), which the reference (B=1 only, modules.py:946-953) does not."""

    def __init__(self, num_phones, num_speakers, max_pitch, max_duration, cond_type=None, lr: float = 2e-4):
        super().__init__()
        if cond_type in ('fasttext', 'hf'):
            in_sz = 300 if cond_type == 'fasttext' else 768
            ext = 512
            self._lm_t = nn.LSTM(input_size=in_sz, num_layers=2, hidden_size=256, batch_first=True, bidirectional=True)
            self._lm_g = nn.LSTM(input_size=in_sz, num_layers=2, hidden_size=256, batch_first=True, bidirectional=True)
            self._use_cond = True
        else:
            ext = 0
            self._lm_t = nn.Linear(1, 1)
            self._lm_g = nn.Linear(1, 1)
            self._use_cond = False
        self._pframes = 1
        self._lr = lr
        self._max_pitch = max_pitch
        self._max_dur = max_duration
        self._phon_emb_t = nn.Embedding(num_phones + 1, 64, padding_idx=0)
        self._phon_emb_g = nn.Embedding(num_phones + 1, 64, padding_idx=0)
        self._speaker_emb_t = nn.Embedding(num_speakers + 1, 128, padding_idx=0)
        self._speaker_emb_g = nn.Embedding(num_speakers + 1, 128, padding_idx=0)
        cnn_t, cnn_g = [], []
        inp = 64
        for _ in range(3):
            cnn_t += [ConvNorm(inp, 256, kernel_size=3, padding=1, w_init_gain='tanh'), nn.Tanh()]
            cnn_g += [ConvNorm(inp, 256, kernel_size=3, padding=1, w_init_gain='tanh'), nn.Tanh()]
            inp = 256
        self._char_cnn_t = nn.ModuleList(cnn_t)
        self._char_cnn_g = nn.ModuleList(cnn_g)
        self._char_rnn_t = nn.LSTM(input_size=256, hidden_size=256, num_layers=2, bidirectional=True, batch_first=True)
        self._char_rnn_g = nn.LSTM(input_size=256, hidden_size=256, num_layers=2, bidirectional=True, batch_first=True)
        self._dur_rnn = nn.LSTM(input_size=512 + 128 + ext, hidden_size=256, num_layers=2, bidirectional=True, batch_first=True)
        self._dur_output = LinearNorm(512, max_duration + 1)
        self._pitch_rnn = nn.LSTM(input_size=512 + 128 + ext, hidden_size=256, num_layers=2, bidirectional=True, batch_first=True)
        self._pitch_output = LinearNorm(512, 2)
        self._cond_rnn = nn.LSTM(input_size=512 + 128 + ext + 1, hidden_size=64, num_layers=2, bidirectional=True, batch_first=True)
        self._cond_output = LinearNorm(128, 80)
        self._hip = {}

    def _lstm(self, name):
        if name not in self._hip:
            self._hip[name] = LSTMHip(getattr(self, name))
        return self._hip[name]

    def _cnn(self, name):
        if name not in self._hip:
            ml = getattr(self, name)
            self._hip[name] = _ConvStack([(ml[0].conv, None), (ml[2].conv, None), (ml[4].conv, None)])
        return self._hip[name]

    @torch.jit.ignore
    def _get_device(self):
        p = self._dur_output.linear_layer.weight
        if p.device.type == 'cpu':
            raise _lib.TTSCError('Languasito2: parameters live on the CPU; move the module to a HIP device (no CPU path)')
        return p.device

    def _text_stack(self, which, x_char, x_speaker, lengths, X, hf_cond):
        emb = getattr(self, '_phon_emb_' + which).weight[x_char]
        spk = getattr(self, '_speaker_emb_' + which).weight[x_speaker]       # [B,1,128]
        h = _cnn_forward(self._cnn('_char_cnn_' + which), emb, lengths)
        h = self._lstm('_char_rnn_' + which)(h, lengths=lengths)
        h = torch.cat([h, spk.expand(-1, h.shape[1], -1)], dim=-1)
        if self._use_cond:
            x_words = X.get('x_words')
            if X.get('x_tok_ids') is not None:
                raise NotImplementedError('conditioning=hf:<model>: supply X["x_words"] ([B,Nw,768] encoder states per word); '
                                          'pretrained encoders cannot be downloaded here (SURVEY.md §2.1)')
            cond = self._lstm('_lm_' + which)(x_words.to(h.device).float())
            p2w = X['x_phon2word'].to(h.device)
            sel = torch.gather(cond, 1, p2w[:, :, None].expand(-1, -1, cond.shape[2]))
            h = torch.cat([h, sel], dim=-1)
        return h.contiguous()

    def forward(self, X, hf_cond=None):
        """modules.py:996-999 (teacher-forced: X carries y_frame2phone / y_pitch): (output_dur [B, N, D + 1], output_pitch [B, F], output_vuv [B, F],
        conditioning [B, F, 80]).  With gradients enabled in training mode the differentiable path runs (HIP kernels behind autograd,
        networks/training.py::languasito_forward_train — what Cubegan.training_step's first line calls, cubegan.py:93); otherwise the
        inference kernels (validation, forced-alignment synthesis)."""
        from . import training as T
        if torch.is_grad_enabled() and self.training:
            return T.languasito_forward_train(self, X)
        return T.languasito_forward(self, X)

    def inference(self, X, hf_cond=None, return_aux=False, check_status=True, timers=None):
        """modules.py:1001-1009.  X: 'x_char' long [B,N] (0 = pad), 'x_speaker' long [B,1].  Returns conditioning [B,F,80]
        (zero rows beyond each utterance's own frame count); X['y_frame2phone'] / X['y_pitch'] are (re)written like
        the reference does.  `timers` (optional list): (phase name, HIP event) pairs are appended at the phase boundaries
        ('text' = phoneme-level stacks + duration head, 'alignment' = durations -> frame map + row expansion, 'frames' = frame-level
        stacks) — bench.py --mode e2e reports them."""
        def mark(name):
            if timers is not None:
                ev = torch.cuda.Event(enable_timing=True)
                ev.record()
                timers.append((name, ev))
        X.pop('y_frame2phone', None)
        dev = self._get_device()
        xc, xs = X['x_char'], X['x_speaker']
        if UPLOAD_RING and xc.shape[0] <= G_STREAM_MAX_B and xc.device.type == 'cpu' and xs.device.type == 'cpu' and xc.dtype == torch.int64 and xs.dtype == torch.int64 and dev.type == 'cuda':
            # phone ids, speaker ids and lengths in ONE page-locked upload (lengths counted on the host copy)
            # (small batches only: behind a generator that fills the chip — Cubegan.inference_pipelined — the asynchronous copy starts late and the
            # text stack of the next batch loses its overlap: 64 sentences 30.7 ms with blocking copies, 33.8-36.2 with the ring, profiles/r06_e2e_upload_ring_ab.log)
            lens = _char_lengths(X, xc)
            x_char, x_speaker, len64 = _lib.upload_ints([xc, xs, torch.tensor(lens, dtype=torch.int64)], dev)
            lengths = _lib.DevLengths(lens, dev_tensor=len64.to(torch.int32))
        else:
            x_char = xc.to(dev)
            x_speaker = xs.to(dev)
            lengths = _lib.DevLengths(_char_lengths(X, x_char), device=dev)   # one upload; every layer below takes the device copy
        B, N = x_char.shape
        with torch.no_grad():
            mark('start')
            hcs = self._text_stack('t', x_char, x_speaker, lengths, X, hf_cond)
            hd = self._lstm('_dur_rnn')(hcs, lengths=lengths)
            out_dur = linear_hip(hd, self._dur_output.linear_layer.weight, self._dur_output.linear_layer.bias)
            # The `g` phoneme stack (embedding, char CNN, two BiLSTM layers: ~0.6 ms of latency-bound launches at B = 1) shares nothing with the
            # `t` stack but the inputs and is not needed before `_cond_rnn`: for small batches it is queued NOW — behind the duration head on the
            # host, which then waits for the frame counts anyway — on a stream of its own, and runs beside the `t` / duration / pitch recurrences
            # instead of after them (each split recurrence holds 8 CUs per utterance; hand-off areas are per stream).  Same launches, same bits.
            g_side = None
            if G_STREAM and B <= G_STREAM_MAX_B and x_char.is_cuda:
                cur_s = torch.cuda.current_stream(dev)
                g_side = _g_stream(dev)
                g_side.wait_stream(cur_s)       # (inputs and length table are uploaded on the current stream)
                with torch.cuda.stream(g_side):
                    g_char = self._text_stack('g', x_char, x_speaker, lengths, X, hf_cond)
                for t_ in (x_char, x_speaker, getattr(lengths, 'dev', None)):
                    if t_ is not None:
                        t_.record_stream(g_side)
                g_char.record_stream(cur_s)
            mark('text')
            # duration head -> frame->phone map on the device (the reference goes through the host here, modules.py:946-953)
            f2p = align_durations(out_dur, lengths)
            X['y_frame2phone'] = f2p
            hexp, flens = _expand_rows(hcs, f2p)
            mark('alignment')
            F_ = hexp.shape[1]
            if F_ == 0:
                X['y_pitch'] = torch.zeros((B, 0), device=dev)
                cond = torch.zeros((B, 0, 80), device=dev)
                return (cond, f2p if return_aux == 'device' else f2p.durations(), flens) if return_aux else cond
            hp = self._lstm('_pitch_rnn')(hexp, lengths=flens)
            op = linear_hip(hp, self._pitch_output.linear_layer.weight, self._pitch_output.linear_layer.bias, act='sigmoid')
            if g_side is not None:
                torch.cuda.current_stream(dev).wait_stream(g_side)
                g = g_char
            else:
                g = self._text_stack('g', x_char, x_speaker, lengths, X, hf_cond)
            if COND_INPUT_FUSED and op.is_cuda:
                # voiced flag, pitch, row expansion, the pitch feature and the split GEMM's zero columns in ONE launch (ttsc_cond_input: nine elementwise
                # launches of ~9 us each on the critical path of a sentence otherwise); per element the operations of the lines below
                g = g.float().contiguous()
                Cg = g.shape[2]
                Cp = (Cg + 1 + 3) // 4 * 4
                pitch = torch.empty((B, F_), dtype=torch.float32, device=dev)
                gin = torch.empty((B, F_, Cp), dtype=torch.float32, device=dev)
                opc = op.float().contiguous()
                with _lib.on_device(dev):
                    _lib.check(_lib.lib().ttsc_cond_input(_lib.dev_ptr(g), _lib.dev_ptr(f2p.f2p), _lib.dev_ptr(f2p.flen_dev), _lib.dev_ptr(opc),
                                                          float(self._max_pitch), B, g.shape[1], Cg, Cp, f2p.f2p.shape[1], F_, _lib.dev_ptr(pitch),
                                                          _lib.dev_ptr(gin), _lib.current_stream()), 'ttsc_cond_input')
                X['y_pitch'] = pitch
                g = gin
            else:
                vuv = torch.round(op[:, :, 1])
                pitch = (op[:, :, 0] * self._max_pitch) * vuv
                X['y_pitch'] = pitch
                g, _ = _expand_rows(g, f2p)
                g = torch.cat([g, (pitch / self._max_pitch).unsqueeze(2)], dim=-1).contiguous()
            g = self._lstm('_cond_rnn')(g, lengths=flens)
            cond = linear_hip(g, self._cond_output.linear_layer.weight, self._cond_output.linear_layer.bias)
            if B > 1:
                fmask = (torch.arange(F_, device=dev)[None, :] < _lib.lengths_dev(flens, dev)[:, None]).float()
                cond = cond * fmask[:, :, None]
            mark('frames')
        if check_status:   # (Cubegan.inference polls once for the whole synthesis instead)
            _lib.check_split_status('Languasito2.inference')
        # return_aux='device': the Alignment itself (durations stay on the device — reading them back here made the host wait for the whole frame-level
        # stack before it could queue the generator: ~0.15 ms of GPU idle time per sentence)
        return (cond, f2p if return_aux == 'device' else f2p.durations(), flens) if return_aux else cond

    @torch.jit.ignore
    def save(self, path):
        torch.save(self.state_dict(), path)

    @torch.jit.ignore
    def load(self, path):
        self.load_state_dict(torch.load(path, map_location='cpu'))
