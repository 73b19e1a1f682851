"""Training-side helpers of the hot path.

`languasito_forward` / `wavernn_loss` evaluate the teacher-forced paths with the inference kernels (no autograd):
they serve validation and the forced-alignment synthesis (`Cubegan.forward`, cubegan.py:65-72)."""
import torch

from ..hip_layers import linear_hip
from .modules import _expand_rows


def languasito_forward(lang, X):
    """Languasito2.forward (modules.py:996-999) with given alignments/pitch:
    returns (output_dur [B,N,D+1], output_pitch [B,F], output_vuv [B,F], conditioning [B,F,80])."""
    dev = lang._get_device()
    x_char, x_speaker = X['x_char'].to(dev), X['x_speaker'].to(dev)
    B = x_char.shape[0]
    lengths = (x_char != 0).sum(dim=1).tolist() if B > 1 else None
    f2p = X['y_frame2phone']
    with torch.no_grad():
        hcs = lang._text_stack('t', x_char, x_speaker, lengths, X, None)
        hd = lang._lstm('_dur_rnn')(hcs, lengths=lengths)
        out_dur = linear_hip(hd, lang._dur_output.linear_layer.weight, lang._dur_output.linear_layer.bias)
        hexp, flens = _expand_rows(hcs, f2p)
        fl = flens if B > 1 else None
        hp = lang._lstm('_pitch_rnn')(hexp, lengths=fl)
        op = linear_hip(hp, lang._pitch_output.linear_layer.weight, lang._pitch_output.linear_layer.bias, act='sigmoid')
        g = lang._text_stack('g', x_char, x_speaker, lengths, X, None)
        g, _ = _expand_rows(g, f2p)
        pitch = (X['y_pitch'].to(dev).float().unsqueeze(2) / lang._max_pitch)
        m = min(g.shape[1], pitch.shape[1])
        g = torch.cat([g[:, :m], pitch[:, :m]], dim=-1).contiguous()
        g = lang._lstm('_cond_rnn')(g, lengths=[min(f, m) for f in flens] if B > 1 else None)
        cond = linear_hip(g, lang._cond_output.linear_layer.weight, lang._cond_output.linear_layer.bias)
    return out_dur, op[:, :, 0], op[:, :, 1], cond


def wavernn_loss(net, X):
    """WaveRNN.training_step's loss value (modules.py:553-563): CE of teacher-forced logits against the target audio."""
    gs = X['x']
    xin = torch.nn.functional.pad(gs[:, :-1], (1, 0), mode='constant', value=0)
    Xt = dict(X)
    Xt['x'] = xin.to(net._get_device())
    logits = net._train_forward(Xt)
    L = logits.shape[1]
    return net._output_functions.loss(logits, gs[:, :L].to(logits.device))
