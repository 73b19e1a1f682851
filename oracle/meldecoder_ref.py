"""ORACLE (test infrastructure — never imported by the product path).

CPU restatement (explicit fp32 torch tensor algebra, no fused nn.LSTM) of the reference's mel-spectrogram decoders:

  Languasito2.inference / _text_forward / _cond_forward / _expand_i   cube/networks/modules.py:916-1009,1043-1053
  CubenetTextcoder.inference / forward / _expand                       cube/networks/textcoder.py:100-189,291-302
  PreNet (dropout p=0.5 ALWAYS on; masks injected)                     cube/networks/modules.py:148-164
  PostNet (Conv k5 + BatchNorm1d(eval) + tanh, x5)                     cube/networks/modules.py:117-145

Pinned against the reference itself (imported here) by tools/gen_golden_meldecoder.py / tools/gen_golden_training.py ->
tests/golden/languasito2_*.npz / textcoder_*.npz -> tests/test_oracle_meldecoder.py (inference, teacher-forced forward, text losses and their
parameter gradients, the external-conditioning branch).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import numpy as np
import torch
import torch.nn.functional as F


def fill_state_dict(named_shapes, seed):
    """Seeded synthetic weights for any module: {name: shape} -> {name: fp32 tensor}.  Fan-in scaled so that
    activations stay O(1); BatchNorm running stats get plausible non-trivial values.  Deterministic in (order, seed),
    so the same call fills the reference module (golden generation) and the build's mirror (tests)."""
    rng = np.random.RandomState(seed)
    out = {}
    for name, shape in named_shapes:
        shape = tuple(shape)
        if name.endswith('num_batches_tracked'):
            out[name] = torch.tensor(100, dtype=torch.long)
            continue
        if name.endswith('running_var'):
            v = rng.uniform(0.5, 1.5, size=shape)
        elif name.endswith('running_mean'):
            v = rng.uniform(-0.2, 0.2, size=shape)
        elif len(shape) == 1:
            v = rng.uniform(-0.1, 0.1, size=shape)
            if '.1.weight' in name or '.5.weight' in name or '.9.weight' in name or '.13.weight' in name:
                v = rng.uniform(0.8, 1.2, size=shape)  # BatchNorm gamma inside PostNet's nn.Sequential
        elif 'emb' in name:
            v = rng.randn(*shape)
        else:
            fan_in = int(np.prod(shape[1:]))
            v = rng.uniform(-1, 1, size=shape) * np.sqrt(3.0 / fan_in) * 1.3
        out[name] = torch.from_numpy(np.asarray(v, dtype=np.float32).reshape(shape))
    return out


def named_shapes(module):
    return [(k, tuple(v.shape)) for k, v in module.state_dict().items()]


# ---- primitives -------------------------------------------------------------------------------------------------
def lstm_dir(x, w_ih, w_hh, b_ih, b_hh, reverse=False, hx=None):
    """One direction of one LSTM layer (torch gate order i,f,g,o).  x [B,T,I] -> y [B,T,H], (h,c)."""
    B, T, _ = x.shape
    H = w_hh.shape[1]
    h = torch.zeros(B, H) if hx is None else hx[0]
    c = torch.zeros(B, H) if hx is None else hx[1]
    ys = [None] * T
    xg = x @ w_ih.t() + b_ih
    order = range(T - 1, -1, -1) if reverse else range(T)
    for t in order:
        g = xg[:, t] + h @ w_hh.t() + b_hh
        i, f, gg, o = g[:, :H], g[:, H:2 * H], g[:, 2 * H:3 * H], g[:, 3 * H:]
        c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
        h = torch.sigmoid(o) * torch.tanh(c)
        ys[t] = h
    return torch.stack(ys, dim=1), (h, c)


def lstm(x, sd, prefix, num_layers, bidirectional, hx=None):
    """torch.nn.LSTM(batch_first=True) from its state_dict keys.  hx = (h0,c0) [L*D,B,H] or None."""
    hs, cs = [], []
    D = 2 if bidirectional else 1
    for l in range(num_layers):
        outs = []
        for d, sfx in enumerate(['', '_reverse'][:D]):
            k = lambda n: sd['%s.%s_l%d%s' % (prefix, n, l, sfx)]
            h0 = None if hx is None else (hx[0][l * D + d], hx[1][l * D + d])
            y, (h, c) = lstm_dir(x, k('weight_ih'), k('weight_hh'), k('bias_ih'), k('bias_hh'), reverse=(d == 1), hx=h0)
            outs.append(y)
            hs.append(h)
            cs.append(c)
        x = torch.cat(outs, dim=-1)
    return x, (torch.stack(hs), torch.stack(cs))


def char_cnn(x_emb, sd, prefix):
    """3 x [Conv1d k3 p1 + tanh] over [B,N,64] (ModuleList indices 0,2,4 are the ConvNorms)."""
    h = x_emb.permute(0, 2, 1)
    for i in (0, 2, 4):
        h = torch.tanh(F.conv1d(h, sd['%s.%d.conv.weight' % (prefix, i)], sd['%s.%d.conv.bias' % (prefix, i)], padding=1))
    return h.permute(0, 2, 1)


def linear(x, sd, prefix):
    return x @ sd[prefix + '.linear_layer.weight'].t() + sd[prefix + '.linear_layer.bias']


def durations_to_frame2phone(durs):
    f2p = []
    for p, d in enumerate(durs):
        f2p.extend([p] * int(d))
    return f2p


def expand_i(x, alignments):
    """Languasito2._expand_i (modules.py:1043-1053): gather rows by frame2phone, pad with the last aligned row."""
    m = max(len(a) for a in alignments)
    rows = []
    for b, a in enumerate(alignments):
        idx = list(a) + [a[-1]] * (m - len(a)) if len(a) else [0] * m
        rows.append(x[b, idx])
    return torch.stack(rows) if m > 0 else x[:, :0]


# ---- Languasito2 ---------------------------------------------------------------------------------------------
def text_stack(sd, which, x_char, x_speaker, x_words=None, x_phon2word=None):
    """The phoneme-level stack shared by _text_forward / _cond_forward (modules.py:917-943 / 963-990): embedding -> 3 x conv+tanh -> BiLSTM,
    + speaker embedding, + (cond_type 'fasttext' / 'hf') the `_lm_<which>` BiLSTM over the word vectors gathered per phoneme
    (`_get_cond_selection`, modules.py:1079-1082: cond[b, phon2word[b, n]])."""
    # nn.Embedding(padding_idx=0) (modules.py:845-849): row 0 is read like any other row but never receives a gradient
    spk = F.embedding(x_speaker, sd['_speaker_emb_%s.weight' % which], padding_idx=0)           # [B,1,128]
    h = char_cnn(F.embedding(x_char, sd['_phon_emb_%s.weight' % which], padding_idx=0), sd, '_char_cnn_' + which)
    h, _ = lstm(h, sd, '_char_rnn_' + which, 2, True)
    h = torch.cat([h, spk.repeat(1, h.shape[1], 1)], dim=-1)        # [B,N,640]
    if x_words is not None:
        cond, _ = lstm(x_words, sd, '_lm_' + which, 2, True)        # [B,Nw,512]
        sel = torch.stack([cond[b, x_phon2word[b]] for b in range(cond.shape[0])])
        h = torch.cat([h, sel], dim=-1)
    return h


def languasito2_forward(sd, x_char, x_speaker, frame2phone, y_pitch, max_pitch, x_words=None, x_phon2word=None):
    """Languasito2.forward, teacher-forced (modules.py:996-999): X carries the alignments and the target pitch.  Padded batches run exactly like
    the reference's (no masking: the recurrences walk the padding).  -> (output_dur [B,N,D+1], output_pitch [B,F], output_vuv [B,F],
    conditioning [B,min(F,Fp),80]).  Plain differentiable torch algebra: autograd through it gives the oracle's gradients."""
    hcs = text_stack(sd, 't', x_char, x_speaker, x_words, x_phon2word)
    hd, _ = lstm(hcs, sd, '_dur_rnn', 2, True)
    out_dur = linear(hd, sd, '_dur_output')
    hp, _ = lstm(expand_i(hcs, frame2phone), sd, '_pitch_rnn', 2, True)
    op = linear(hp, sd, '_pitch_output')
    g = expand_i(text_stack(sd, 'g', x_char, x_speaker, x_words, x_phon2word), frame2phone)
    p = y_pitch.unsqueeze(2) / max_pitch
    m = min(g.shape[1], p.shape[1])
    g, _ = lstm(torch.cat([g[:, :m], p[:, :m]], dim=-1), sd, '_cond_rnn', 2, True)
    return out_dur, torch.sigmoid(op[:, :, 0]), torch.sigmoid(op[:, :, 1]), linear(g, sd, '_cond_output')


def text_losses(p_dur, p_pitch, p_vuv, y_dur, y_pitch, max_pitch, max_duration):
    """The text-side losses of Cubegan.training_step (cubegan.py:94-112): CE over durations (ignore_index = max(max_pitch, max_duration) + 1,
    modules.py:910) and masked L1 pitch + L1 voicing.  -> (loss_duration, loss_pitch)."""
    t_vuv = (y_pitch > 1).float()
    m = min(y_dur.shape[1], p_dur.shape[1])
    t_dur, p_dur = y_dur[:, :m], p_dur[:, :m, :]
    m = min(y_pitch.shape[1], p_pitch.shape[1])
    t_pitch, p_pitch, t_vuv, p_vuv = y_pitch[:, :m], p_pitch[:, :m], t_vuv[:, :m], p_vuv[:, :m]
    loss_duration = F.cross_entropy(p_dur.reshape(-1, p_dur.shape[2]), t_dur.reshape(-1), ignore_index=int(max(max_pitch, max_duration) + 1))
    loss_pitch = (torch.abs(t_pitch / max_pitch - p_pitch) * t_vuv).mean() + torch.abs(t_vuv - p_vuv).mean()
    return loss_duration, loss_pitch


def languasito2_inference(sd, x_char, x_speaker, max_pitch, x_words=None, x_phon2word=None):
    """modules.py:1001-1009 (cond_type=None; with x_words / x_phon2word: cond_type='fasttext' / 'hf' given the word vectors).
    Returns (conditioning [1,F,80], durations, pitch [1,F])."""
    assert x_char.shape[0] == 1, 'the reference inference path is B=1 (modules.py:946-953)'
    hcs = text_stack(sd, 't', x_char, x_speaker, x_words, x_phon2word)      # [1,N,640(+512)]
    hd, _ = lstm(hcs, sd, '_dur_rnn', 2, True)
    out_dur = linear(hd, sd, '_dur_output')
    durs = torch.argmax(out_dur, dim=-1).reshape(-1).tolist()
    f2p = durations_to_frame2phone(durs)
    if len(f2p) == 0:
        return torch.zeros(1, 0, 80), durs, torch.zeros(1, 0)
    hp, _ = lstm(expand_i(hcs, [f2p]), sd, '_pitch_rnn', 2, True)
    op = linear(hp, sd, '_pitch_output')
    vuv = torch.round(torch.sigmoid(op[:, :, 1]))
    pitch = (torch.sigmoid(op[:, :, 0]) * max_pitch) * vuv
    # conditioning stack (the _g copies), modules.py:962-994
    g = expand_i(text_stack(sd, 'g', x_char, x_speaker, x_words, x_phon2word), [f2p])
    p = pitch.unsqueeze(2) / max_pitch
    m = min(g.shape[1], p.shape[1])
    g = torch.cat([g[:, :m], p[:, :m]], dim=-1)
    g, _ = lstm(g, sd, '_cond_rnn', 2, True)
    return linear(g, sd, '_cond_output'), durs, pitch


# ---- Textcoder ------------------------------------------------------------------------------------------------
def prenet(x, sd, masks):
    """modules.py:159-164: relu(Linear) then dropout(p=0.5, train=True) == * mask * 2, for both layers."""
    h = x
    for i in range(2):
        h = torch.relu(h @ sd['_prenet.layers_h.%d.linear_layer.weight' % i].t() + sd['_prenet.layers_h.%d.linear_layer.bias' % i])
        h = h * masks[i] * 2.0
    return h


def postnet(x, sd, eps=1e-5):
    """modules.py:117-145 in eval mode: conv k5 p2 -> BatchNorm1d(running stats) -> tanh (x4) -> conv k5 p2."""
    h = x.permute(0, 2, 1)
    for n, (ci, bi) in enumerate([(0, 1), (4, 5), (8, 9), (12, 13), (16, None)]):
        h = F.conv1d(h, sd['_postnet.network.%d.conv.weight' % ci], sd['_postnet.network.%d.conv.bias' % ci], padding=2)
        if bi is not None:
            p = '_postnet.network.%d.' % bi
            h = (h - sd[p + 'running_mean'][None, :, None]) / torch.sqrt(sd[p + 'running_var'][None, :, None] + eps)
            h = h * sd[p + 'weight'][None, :, None] + sd[p + 'bias'][None, :, None]
            h = torch.tanh(h)
    return h.permute(0, 2, 1)


def expand_pframes(x, alignments, pframes):
    """CubenetTextcoder._expand (textcoder.py:291-302): one row per `pframes` frames."""
    m = max(len(a) // pframes for a in alignments)
    rows = []
    for b, a in enumerate(alignments):
        n = len(a) // pframes
        idx = [a[j * pframes] for j in range(n)]
        r = x[b, idx] if n else x[b, :0]
        if m - n:
            r = torch.cat([r, x[b, -1:].repeat(m - n, 1)], dim=0)
        rows.append(r)
    return torch.stack(rows)


def textcoder_text_stack(sd, x_char, x_speaker):
    spk = sd['_speaker_emb.weight'][x_speaker]
    h = char_cnn(sd['_phon_emb.weight'][x_char], sd, '_char_cnn')
    h, _ = lstm(h, sd, '_rnn_char', 2, True)
    h = torch.cat([h, spk.repeat(1, h.shape[1], 1)], dim=-1)
    hd, _ = lstm(h, sd, '_dur_rnn', 2, True)
    return h, linear(hd, sd, '_dur_output')


def textcoder_inference(sd, x_char, x_speaker, masks, pframes=3):
    """textcoder.py:140-189.  masks: float {0,1} [steps, 2, 1, 256] PreNet dropout masks (one pair per AR step)."""
    h, out_dur = textcoder_text_stack(sd, x_char, x_speaker)
    durs = torch.argmax(out_dur, dim=-1).reshape(-1).tolist()
    f2p = durations_to_frame2phone(durs)
    h = expand_pframes(h, [f2p], pframes)
    h, _ = lstm(h, sd, '_rnn_overlay', 2, True)
    last = torch.ones(1, 1, 80) * -5
    hx = None
    outs = []
    for t in range(h.shape[1]):
        pn = prenet(last, sd, masks[t])
        y, hx = lstm(torch.cat([h[:, t:t + 1], pn], dim=-1), sd, '_mel_rnn', 2, False, hx=hx)
        o = linear(y, sd, '_mel_output')
        outs.append(o)
        last = o[:, :, -80:]
    if not outs:
        return torch.zeros(1, 0, 80), durs
    mel = torch.cat(outs, dim=1).reshape(1, -1, 80)
    return mel + postnet(mel, sd), durs


def textcoder_forward(sd, x_char, x_speaker, frame2phone, y_mgc, masks, pframes=3):
    """Teacher-forced path (textcoder.py:100-138): returns (output_dur, output_mel, output_mel_post)."""
    h, out_dur = textcoder_text_stack(sd, x_char, x_speaker)
    h = expand_pframes(h, frame2phone, pframes)
    h, _ = lstm(h, sd, '_rnn_overlay', 2, True)
    B = y_mgc.shape[0]
    lst = [torch.ones(B, 1, 80) * -5] + [y_mgc[:, (i + 1) * pframes - 1].unsqueeze(1) for i in range(y_mgc.shape[1] // pframes)]
    cond = prenet(torch.cat(lst, dim=1), sd, masks)
    m = min(h.shape[1], cond.shape[1])
    y, _ = lstm(torch.cat([h[:, :m], cond[:, :m]], dim=-1), sd, '_mel_rnn', 2, False)
    mel = linear(y, sd, '_mel_output').reshape(B, -1, 80)
    return out_dur, mel, mel + postnet(mel, sd)
