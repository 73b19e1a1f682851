"""Mirror of cube/networks/textcoder.py: ``CubenetTextcoder`` — phonemes -> log10-mel (two-stage path of
cube/io_utils/runtime.py:41-80).  Same constructor / state_dict keys; `inference` and the teacher-forced `forward` run
on the HIP conv / GEMM / LSTM kernels."""
import ctypes as C

import torch
import torch.nn as nn

from .. import _lib
from ..hip_layers import LSTMHip, linear_hip
from .modules import ConvNorm, LinearNorm, PostNet, PreNet, _ConvStack, _char_lengths, _cnn_forward, _expand_rows, align_durations


class CubenetTextcoder(nn.Module):
    def __init__(self, encodings, pframes: int = 3, lr: float = 2e-4):
        super().__init__()
        self._pframes = pframes
        self._lr = lr
        self._encodings = encodings
        self._phon_emb = nn.Embedding(len(encodings.phon2int) + 1, 64, padding_idx=0)
        self._speaker_emb = nn.Embedding(len(encodings.speaker2int) + 1, 128, padding_idx=0)
        cnn = []
        inp = 64
        for _ in range(3):
            cnn += [ConvNorm(inp, 256, kernel_size=3, padding=1, w_init_gain='tanh'), nn.Tanh()]
            inp = 256
        self._char_cnn = nn.ModuleList(cnn)
        self._rnn_char = nn.LSTM(input_size=256, hidden_size=256, num_layers=2, bidirectional=True, batch_first=True)
        self._rnn_overlay = nn.LSTM(input_size=640, hidden_size=512, num_layers=2, bidirectional=True, batch_first=True)
        self._dur_rnn = nn.LSTM(input_size=640, hidden_size=256, num_layers=2, bidirectional=True, batch_first=True)
        self._dur_output = LinearNorm(512, encodings.max_duration + 1)
        self._pitch_rnn = nn.LSTM(input_size=1024, hidden_size=256, num_layers=2, bidirectional=True, batch_first=True)
        self._pitch_output = LinearNorm(512, int(encodings.max_pitch) + 1)
        self._mel_rnn = nn.LSTM(input_size=1024 + 256, hidden_size=512, num_layers=2, bidirectional=False, batch_first=True)
        self._mel_output = LinearNorm(512, 80 * pframes)
        self._prenet = PreNet(80, 256, 2)
        self._postnet = PostNet(80)
        self._hip = {}

    def _lstm(self, name):
        if name not in self._hip:
            self._hip[name] = LSTMHip(getattr(self, name))
        return self._hip[name]

    def _cnn(self):
        if '_cnn' not in self._hip:
            ml = self._char_cnn
            self._hip['_cnn'] = _ConvStack([(ml[0].conv, None), (ml[2].conv, None), (ml[4].conv, None)])
        return self._hip['_cnn']

    @torch.jit.ignore
    def _get_device(self):
        p = self._mel_output.linear_layer.weight
        if p.device.type == 'cpu':
            raise _lib.TTSCError('CubenetTextcoder: parameters live on the CPU; move the module to a HIP device')
        return p.device

    def _text_stack(self, x_char, x_speaker, lengths):
        emb = self._phon_emb.weight[x_char]
        spk = self._speaker_emb.weight[x_speaker]
        h = _cnn_forward(self._cnn(), emb, lengths)
        h = self._lstm('_rnn_char')(h, lengths=lengths)
        h = torch.cat([h, spk.expand(-1, h.shape[1], -1)], dim=-1).contiguous()
        hd = self._lstm('_dur_rnn')(h, lengths=lengths)
        return h, linear_hip(hd, self._dur_output.linear_layer.weight, self._dur_output.linear_layer.bias)

    def inference(self, X, dropout_masks=None):
        """textcoder.py:140-189 (B=1 like the reference).  dropout_masks: optional [steps, 2, 1, 256] {0,1} PreNet masks
        (parity tests); otherwise drawn from torch's device generator.  Returns post-net mel [1, F, 80] (log10)."""
        dev = self._get_device()
        x_char, x_speaker = X['x_char'].to(dev), X['x_speaker'].to(dev)
        assert x_char.shape[0] == 1, 'CubenetTextcoder.inference follows the reference: one utterance per call'
        with torch.no_grad():
            h, out_dur = self._text_stack(x_char, x_speaker, None)
            # duration head -> alignment -> gathered overlay input, all on the device (textcoder.py:160-166,291-302 do it on the host)
            h, _ = _expand_rows(h, align_durations(out_dur, None), stride=self._pframes)
            if h.shape[1] == 0:
                return torch.zeros((1, 0, 80), device=dev)
            h = self._lstm('_rnn_overlay')(h)
            mel = self._ar_decode(h, dropout_masks)
            out = self._postnet(mel, add_residual=True)
            _lib.check_split_status('CubenetTextcoder.inference')
            return out

    def _melar_handle(self):
        """(re)build the persistent AR-decoder handle when a parameter changed"""
        mods = [self._mel_rnn, self._mel_output, self._prenet]
        sig = tuple((p.data_ptr(), p._version) for m in mods for p in m.parameters())
        if self._hip.get('_melar_sig') == sig:
            return self._hip['_melar']
        L = _lib.lib()
        _lib.require_gpu()
        if '_melar' in self._hip:
            L.ttsc_melar_destroy(self._hip['_melar'])
        hnd = C.c_void_p()
        _lib.check(L.ttsc_melar_create(self._mel_rnn.hidden_size, 256, 80, 80 * self._pframes, C.byref(hnd)), 'ttsc_melar_create')
        g = lambda t: t.detach().float().cpu().contiguous()
        r = self._mel_rnn
        ts = [g(r.weight_ih_l0), g(r.weight_hh_l0), g(r.weight_ih_l1), g(r.weight_hh_l1), g(r.bias_ih_l1), g(r.bias_hh_l1),
              g(self._mel_output.linear_layer.weight), g(self._mel_output.linear_layer.bias),
              g(self._prenet.layers_h[0].linear_layer.weight), g(self._prenet.layers_h[0].linear_layer.bias),
              g(self._prenet.layers_h[1].linear_layer.weight), g(self._prenet.layers_h[1].linear_layer.bias)]
        P = lambda t: C.c_void_p(t.data_ptr())
        _lib.check(L.ttsc_melar_set_weights(hnd, P(ts[0]), ts[0].shape[1], *[P(t) for t in ts[1:]]), 'ttsc_melar_set_weights')
        self._hip['_melar'] = hnd
        self._hip['_melar_sig'] = sig
        return hnd

    def _ar_decode(self, h, dropout_masks=None, steps=None, seed=None):
        """textcoder.py:174-185 as ONE persistent kernel launch (csrc/melar.hip).  h: overlay states [B, S, 1024]."""
        hnd = self._melar_handle()
        B, S, _ = h.shape
        r = self._mel_rnn
        n_ov = r.weight_ih_l0.shape[1] - 256
        xg1 = linear_hip(h, r.weight_ih_l0[:, :n_ov].contiguous(), r.bias_ih_l0 + r.bias_hh_l0)    # hoisted input projection
        y = torch.empty((B, S, 80 * self._pframes), dtype=torch.float32, device=h.device)
        m = None
        if dropout_masks is not None:
            m = torch.as_tensor(dropout_masks).to(h.device).float().reshape(S, 2, B, 256).permute(2, 0, 1, 3).contiguous()
        if seed is None:
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        st = torch.as_tensor(steps, dtype=torch.int32, device=h.device) if steps is not None else None
        with _lib.on_device(h.device):
            _lib.check(_lib.lib().ttsc_melar_decode(hnd, _lib.dev_ptr(xg1), B, S, _lib.dev_ptr(m) if m is not None else None,
                                                    C.c_uint64(seed), _lib.dev_ptr(st) if st is not None else None, _lib.dev_ptr(y),
                                                    _lib.current_stream()), 'ttsc_melar_decode')
        return y.reshape(B, S * self._pframes, 80).contiguous()

    def _ar_decode_stepwise(self, h, dropout_masks=None):
        """The same loop as one kernel launch per layer and step (kept for A/B tests of the persistent kernel)."""
        dev = h.device
        last = torch.full((1, 1, 80), -5.0, device=dev)
        hx = None
        outs = []
        rnn = self._lstm('_mel_rnn')
        for t in range(h.shape[1]):
            m = None if dropout_masks is None else [dropout_masks[t][0].to(dev), dropout_masks[t][1].to(dev)]
            pn = self._prenet(last, masks=m)
            y, hx = rnn(torch.cat([h[:, t:t + 1], pn], dim=-1).contiguous(), hx=hx, return_state=True)
            o = linear_hip(y, self._mel_output.linear_layer.weight, self._mel_output.linear_layer.bias)
            outs.append(o)
            last = o[:, :, -80:].contiguous()
        return torch.cat(outs, dim=1).reshape(1, -1, 80).contiguous()

    def forward(self, X, dropout_masks=None):
        """Teacher-forced path (textcoder.py:100-138), inference numerics (no autograd): returns
        (output_dur, output_pitch, output_mel, output_mel_post)."""
        dev = self._get_device()
        x_char, x_speaker = X['x_char'].to(dev), X['x_speaker'].to(dev)
        B = x_char.shape[0]
        lengths = _char_lengths(X, x_char) if B > 1 else None
        with torch.no_grad():
            h, out_dur = self._text_stack(x_char, x_speaker, lengths)
            h, flens = _expand_rows(h, X['y_frame2phone'], stride=self._pframes)
            h = self._lstm('_rnn_overlay')(h, lengths=flens if B > 1 else None)
            hp = self._lstm('_pitch_rnn')(h, lengths=flens if B > 1 else None)
            out_pitch = linear_hip(hp, self._pitch_output.linear_layer.weight, self._pitch_output.linear_layer.bias)
            y_mgc = X['y_mgc'].to(dev).float()
            lst = [torch.full((B, 1, 80), -5.0, device=dev)]
            for ii in range(y_mgc.shape[1] // self._pframes):
                lst.append(y_mgc[:, (ii + 1) * self._pframes - 1, :].unsqueeze(1))
            cond = self._prenet(torch.cat(lst, dim=1).contiguous(), masks=dropout_masks)
            m = min(h.shape[1], cond.shape[1])
            y = self._lstm('_mel_rnn')(torch.cat([h[:, :m], cond[:, :m]], dim=-1).contiguous())
            mel = linear_hip(y, self._mel_output.linear_layer.weight, self._mel_output.linear_layer.bias).reshape(B, -1, 80).contiguous()
            return out_dur, out_pitch, mel, self._postnet(mel, add_residual=True)

    @torch.jit.ignore
    def save(self, path):
        torch.save(self.state_dict(), path)

    @torch.jit.ignore
    def load(self, path):
        self.load_state_dict(torch.load(path, map_location='cpu'))
