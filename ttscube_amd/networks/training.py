"""Training-side helpers of the hot path.

`languasito_forward` / `wavernn_loss` evaluate the teacher-forced paths with the inference kernels (no autograd):
they serve validation and the forced-alignment synthesis (`Cubegan.forward`, cubegan.py:65-72)."""
import torch

from .. import _lib
from ..hip_layers import linear_hip
from .modules import _char_lengths, _expand_rows


def _h2d(X, key, dev):
    """batch tensor `key` on the device: from page-locked memory and without blocking the host (io_utils.loader.pin_batch; a batch that arrives
    pageable is pinned here once and the pinned copy kept in the batch dict, so a batch that is stepped on repeatedly is pinned once)"""
    t = X[key]
    if not torch.is_tensor(t) or t.device.type != 'cpu' or dev.type != 'cuda':
        return t.to(dev)
    if not t.is_pinned() and t.numel() > 0:
        cache = X.setdefault('_pinned', {})
        hit = cache.get(key)
        if hit is None or hit[0] is not t:
            try:
                hit = (t, t.pin_memory())
            except RuntimeError:
                return t.to(dev)
            cache[key] = hit
        t = hit[1]
    return t.to(dev, non_blocking=True)


def languasito_forward(lang, X):
    """Languasito2.forward (modules.py:996-999) with given alignments/pitch:
    returns (output_dur [B,N,D+1], output_pitch [B,F], output_vuv [B,F], conditioning [B,F,80])."""
    dev = lang._get_device()
    x_char, x_speaker = X['x_char'].to(dev), X['x_speaker'].to(dev)
    B = x_char.shape[0]
    lengths = _char_lengths(X, x_char) if B > 1 else None
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
    _lib.check_split_status('languasito_forward')   # a timed-out split recurrence must not return garbage silently
    return out_dur, op[:, :, 0], op[:, :, 1], cond


def wavernn_loss(net, X):
    """WaveRNN.training_step's loss value (modules.py:553-563): CE of teacher-forced logits against the target audio."""
    gs = X['x']
    xin = torch.nn.functional.pad(gs[:, :-1], (1, 0), mode='constant', value=0)
    Xt = dict(X)
    Xt['x'] = xin.to(net._get_device())
    logits = net._train_forward(Xt)
    L = logits.shape[1]
    _lib.check_split_status('wavernn_loss')
    return net._output_functions.loss(logits, gs[:, :L].to(logits.device))


# =====================================================================================================================
# Training steps (autograd graph of hand-written HIP kernels + explicit RCCL gradient exchange).
#
# Every parametrised layer of the two steps runs forward AND backward on `ttsc::` kernels behind `torch.autograd.Function`s:
# generator / discriminator convolutions (hifigan/autograd.py, hifigan/disc_hip.py), LSTM / GRU recurrences with their weight and
# input gradient GEMMs (lstm_autograd.py, gru_autograd.py), embeddings / char-CNN / Linears (text_autograd.py), the STFT-mel loss
# (io_utils/melspec.py), the GAN losses (hifigan/losses_hip.py) and AdamW over flat arenas (optim.py).  There is no CPU path and
# no torch-op formulation in this package: the all-torch formulation of the same steps that the native steps are held to lives
# under tests/ (tests/torch_reference.py) and is swapped in through the module-level hooks below (`_text_ops`, `_make_adamw`,
# `_gan_loss_fns`, `_discriminator_fns`, `_lowres_features`, `_output_linears`, `lstm_forward_train`, `gru_forward_train`,
# `generator_forward_with_grad`).  The explicit flat-arena RCCL exchange (ttscube_amd/distributed.py) replaces Lightning's
# implicit DDP.
# =====================================================================================================================
import collections.abc
import contextlib
import itertools
import os
import random

import torch.nn.functional as F

from ..hifigan.autograd import generator_forward_with_grad
from .lstm_autograd import lstm_forward_train
from .gru_autograd import gru_forward_train


PHASE_HOOK = None     # measurement only (tools/probes/train_phase_timeline.py): called with a phase name at the boundaries of cubegan_training_step
TEXT_STREAM = os.environ.get('TTSC_TEXT_STREAM', '1') != '0'
# where in the step the HOST queues the text side (it runs on its own stream either way): 0 = first (round 5), 1 = after the discriminator step's forward
# pass, 2 = after its backward pass, 3 = after the generator step's forward pass, 4 = after its backward pass
TEXT_AT = int(os.environ.get('TTSC_TEXT_AT', '2'))
# 1 (default): the step never makes the host wait for the GPU — the status of its split recurrences guards the AdamW launches on the device (with a gradient
# exchange: its maximum over the ranks, so that no rank applies sums a failed rank has sent garbage into) and travels back with the losses, which are read
# when first looked at (StepLosses).  0 = the host checks before each exchange / update and reads the losses back at the end of the step (round 5)
STEP_LAZY = os.environ.get('TTSC_STEP_LAZY', '1') != '0'
FMAP_RAW = os.environ.get('TTSC_FMAP_RAW', '1') != '0'      # (measurement switch: 0 = activated feature maps through ATen, round 5's path)
_TEXT_STREAMS = {}


def _text_stream(dev):
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    if key not in _TEXT_STREAMS:
        prio = int(os.environ.get('TTSC_TEXT_STREAM_PRIORITY', '0'))
        from ..hifigan.streams import text_stream     # (reserved together with the side streams, in a fixed order: hifigan/streams.py::_reserve)
        _TEXT_STREAMS[key] = text_stream(dev) if prio == 0 else torch.cuda.Stream(device=dev, priority=prio)
        # (normal priority on purpose: a high-priority stream gets a hardware queue of its own, and with MORE than the runtime's default four
        # queues really running side by side this step gets slower, not faster — 77 ms vs 101 ms at b = 16, tools/probes/train_host_bound.py)
    return _TEXT_STREAMS[key]


def _require_device(t, what):
    if not t.is_cuda:
        raise _lib.TTSCError('%s: tensors must live on a HIP device (got %s); the training steps have no CPU path' % (what, t.device))


# ---- hooks: the native building blocks of the steps (tests substitute torch formulations here, never the product) ---------
def _text_ops(lang):
    """(embedding, linear, char_cnn(name, h)) of the text stacks on the HIP kernels (networks/text_autograd.py)"""
    from .text_autograd import char_cnn_train, hip_embedding, hip_linear
    return hip_embedding, hip_linear, (lambda name, h: char_cnn_train(lang, name, h))


def _make_adamw(params, lr):
    """one HIP kernel per group and step over flat parameter / gradient / moment arenas (ttscube_amd/optim.py); the gradient arena
    doubles as the RCCL exchange buffer (distributed.ArenaReducer).  state_dict layout == torch.optim.AdamW's."""
    from ..optim import FlatAdamW
    return FlatAdamW(params, lr, betas=(0.8, 0.99))


def _gan_loss_fns():
    """(discriminator_loss, feature_loss, generator_loss): value + gradient of a whole list of tensors in one launch"""
    from ..hifigan.losses_hip import discriminator_loss, feature_loss, generator_loss
    return discriminator_loss, feature_loss, generator_loss


def _discriminator_fns(model):
    """(mpd(y, y_hat, want_fmap), msd(...)): every convolution of MPD / MSD on the HIP kernels (hifigan/disc_hip.py)"""
    from ..hifigan.disc_hip import mpd_forward, msd_forward
    mpd = lambda a_, b_, fm=True: mpd_forward(model._mpd, a_, b_, want_fmap=fm)
    msd = lambda a_, b_, fm=True: msd_forward(model._msd, a_, b_, want_fmap=fm)
    mpd.accepts_raw = msd.accepts_raw = True     # fm='raw': feature maps as (convolution output, slope) pairs (losses_hip.RawFmap)
    return mpd, msd


def _lowres_features(net, hidden):
    """WaveRNN's three k = 7 ConvNorm + tanh layers over the low-resolution signal (modules.py:416-420,459-461) on the HIP
    convolution / weight-gradient kernels"""
    from ..hifigan.autograd import TrainConv, hip_conv
    cache = net.__dict__.setdefault('_train_lowres', {})
    for i, conv in enumerate(net._lowres_conv):
        c = conv.conv
        tc = cache.get(i)
        if tc is None:
            tc = cache[i] = TrainConv(c.in_channels, c.out_channels, c.kernel_size[0], padding=c.padding[0], dilation=c.dilation[0])
        hidden = torch.tanh(hip_conv(tc, hidden.contiguous(), c.weight, c.bias))
    return hidden


def _output_linears(net, hidden):
    """tanh(Linear H -> 256) -> Linear 256 -> S over all B x L rows (modules.py:535-537) on the MFMA GEMM, forward and backward"""
    from .text_autograd import hip_linear
    pre = torch.tanh(hip_linear(hidden, net._preoutput.linear_layer.weight, net._preoutput.linear_layer.bias))
    return hip_linear(pre, net._output.linear_layer.weight, net._output.linear_layer.bias)


def _languasito_branches(lang, X):
    """The two INDEPENDENT halves of the differentiable Languasito2.forward (modules.py:996-999) as closures:
         text()  -> (output_dur, output_pitch, output_vuv)   phoneme stack `t`, duration and pitch recurrences — the parameters of opt_t
         cond()  -> conditioning [B, F, 80]                  phoneme stack `g`, conditioning recurrence (target pitch as input) — part of opt_g
    They share inputs only (no parameter, no activation), so a caller may run them on different streams."""
    if getattr(lang, '_use_cond', False):
        raise NotImplementedError("training with external conditioning ('fasttext:..' / 'hf:..') is not built: the encoders cannot be "
                                  "downloaded here and the conditioning branch (modules.py:963-990) is inference-only; use conditioning=None")
    dev = lang._get_device()
    x_char, x_speaker = _h2d(X, 'x_char', dev), _h2d(X, 'x_speaker', dev)

    _require_device(x_char, 'languasito_forward_train')
    embed, linear, cnn = _text_ops(lang)

    def stack(which):
        h = embed(getattr(lang, '_phon_emb_' + which), x_char).permute(0, 2, 1)
        h = cnn('_char_cnn_' + which, h)
        h = lstm_forward_train(getattr(lang, '_char_rnn_' + which), h.permute(0, 2, 1))
        spk = embed(getattr(lang, '_speaker_emb_' + which), x_speaker)
        return torch.cat([h, spk.repeat(1, h.shape[1], 1)], dim=-1)

    alignments = X['y_frame2phone']
    m_ = max(len(a) for a in alignments)
    idx = torch.zeros((len(alignments), m_), dtype=torch.long, pin_memory=(dev.type == 'cuda'))
    for b, a in enumerate(alignments):
        idx[b, :len(a)] = torch.as_tensor(a)
        idx[b, len(a):] = a[-1]
    # frame -> phoneme maps are monotone; with the utterance offsets added the flat index list is non-decreasing, which lets the backward pass
    # sum each phoneme's frames as one contiguous run (checked here, on the host copy: anything else takes the general scatter kernel)
    idx_sorted = bool((idx[:, 1:] >= idx[:, :-1]).all()) if idx.shape[1] > 1 else True
    idx_dev = idx.to(dev, non_blocking=True)
    pitch_in = _h2d(X, 'y_pitch', dev).float().unsqueeze(2) / lang._max_pitch

    def expand(x):
        """phoneme rows -> frame rows (modules.py:1043-1053).  A row gather whose backward adds the frames of a phoneme in a FIXED order
        (text_autograd.HipEmbeddingFn: ttsc_rows_gather / ttsc_rows_scatter_add) — torch.gather's backward is a scatter of float atomics, and its
        run-to-run rounding noise, amplified by AdamW's normalisation, made the whole step irreproducible (round 4: ~440 of 900 parameter
        tensors differed between two identical 5-step runs)."""
        from .text_autograd import HipEmbeddingFn
        B_, N_, C_ = x.shape
        flat = idx_dev + torch.arange(B_, dtype=torch.long, device=dev)[:, None] * N_
        return HipEmbeddingFn.apply(x.reshape(B_ * N_, C_), flat, None, idx_sorted and bool(int(idx.max()) < N_))

    def text():
        hcs = stack('t')
        hd = lstm_forward_train(lang._dur_rnn, hcs)
        out_dur = linear(hd, lang._dur_output.linear_layer.weight, lang._dur_output.linear_layer.bias)
        hp = lstm_forward_train(lang._pitch_rnn, expand(hcs))
        op = linear(hp, lang._pitch_output.linear_layer.weight, lang._pitch_output.linear_layer.bias)
        return out_dur, torch.sigmoid(op[:, :, 0]), torch.sigmoid(op[:, :, 1])

    def cond():
        g = expand(stack('g'))
        m = min(g.shape[1], pitch_in.shape[1])
        g = lstm_forward_train(lang._cond_rnn, torch.cat([g[:, :m], pitch_in[:, :m]], dim=-1))
        return linear(g, lang._cond_output.linear_layer.weight, lang._cond_output.linear_layer.bias)

    text.shared_inputs = cond.shared_inputs = [x_char, x_speaker, idx_dev, pitch_in]   # allocated here, read by both closures
    return text, cond


def languasito_forward_train(lang, X):
    """Differentiable Languasito2.forward (modules.py:996-999): (output_dur, output_pitch, output_vuv, conditioning)."""
    text, cond = _languasito_branches(lang, X)
    out_dur, p_pitch, p_vuv = text()
    return out_dur, p_pitch, p_vuv, cond()


def text_losses(p_dur, p_pitch, p_vuv, t_dur, t_pitch, max_pitch, ignore_index):
    """The text-side losses of Cubegan.training_step (cubegan.py:94-112): duration cross-entropy (padding carries `ignore_index` =
    max(max_pitch, max_duration) + 1, modules.py:910) and the voiced-masked L1 pitch + L1 voicing losses.  -> (loss_duration, loss_pitch);
    held to values and gradients made by the reference itself (tests/golden/languasito2_train_*.npz)."""
    t_vuv = (t_pitch > 1).float()
    m = min(t_dur.shape[1], p_dur.shape[1])
    t_dur, p_dur = t_dur[:, :m], p_dur[:, :m, :]
    m = min(t_pitch.shape[1], p_pitch.shape[1])
    t_pitch, p_pitch, t_vuv, p_vuv = t_pitch[:, :m], p_pitch[:, :m], t_vuv[:, :m], p_vuv[:, :m]
    loss_duration = F.cross_entropy(p_dur.reshape(-1, p_dur.shape[2]), t_dur.reshape(-1), ignore_index=ignore_index)
    loss_pitch = (torch.abs(t_pitch / max_pitch - p_pitch) * t_vuv).mean() + torch.abs(t_vuv - p_vuv).mean()
    return loss_duration, loss_pitch


def cubegan_param_groups(model):
    """The three parameter groups of cubegan.py:275-298 (generator side, discriminators, text side)."""
    l = model._languasito
    g = list(itertools.chain(model._generator.parameters(), l._phon_emb_g.parameters(), l._speaker_emb_g.parameters(),
                             l._char_cnn_g.parameters(), l._char_rnn_g.parameters(), l._lm_g.parameters(),
                             l._cond_rnn.parameters(), l._cond_output.parameters()))
    d = list(itertools.chain(model._msd.parameters(), model._mpd.parameters()))
    t = list(itertools.chain(l._phon_emb_t.parameters(), l._speaker_emb_t.parameters(), l._char_cnn_t.parameters(),
                             l._char_rnn_t.parameters(), l._lm_t.parameters(), l._dur_rnn.parameters(),
                             l._dur_output.parameters(), l._pitch_rnn.parameters(), l._pitch_output.parameters()))
    return g, d, t


def cubegan_configure_optimizers(model):
    """cubegan.py:275-311: AdamW(0.8,0.99) x3 + Adam(1e-6) on the dummy; restores `.opt.last` states when present
    (the reference sets `_loaded_optimizer_state` but reads `_loaded_optimizer_states`, so its resume silently skips this)."""
    g, d, t = cubegan_param_groups(model)
    for p_ in itertools.chain(g, d, t):
        _require_device(p_, 'cubegan_configure_optimizers')
    mk = lambda ps: _make_adamw(ps, model._current_lr)
    opt_g, opt_d, opt_t = mk(g), mk(d), mk(t)
    opt_b = torch.optim.Adam(model._dummy.parameters(), lr=1e-6)
    if model._loaded_optimizer_states is not None:
        for k, opt in zip(['0', '1', '2', '3'], [opt_g, opt_d, opt_t, opt_b]):
            if k in model._loaded_optimizer_states:
                opt.load_state_dict(model._loaded_optimizer_states[k])
        model._loaded_optimizer_states = None
    return opt_g, opt_d, opt_t, opt_b


def cubegan_reducers(model, optimizers, force=False, overlap=True, bucket_mb=64):
    """The three gradient exchanges of a Cubegan step, in the order `cubegan_training_step` expects (generator side, discriminators,
    text side).  Over a FlatAdamW group the exchange runs on the optimizer's own gradient arena and its reduce_scatters leave from
    bucket-ready gradient hooks while backward() is still running (distributed.ArenaReducer); over a torch optimizer it is the
    copy-in FlatBucketReducer."""
    from ..distributed import ArenaReducer, FlatBucketReducer
    from ..optim import FlatAdamW
    out = []
    for opt, ps in zip(optimizers[:3], cubegan_param_groups(model)):
        if isinstance(opt, FlatAdamW):
            out.append(ArenaReducer(opt, bucket_mb=bucket_mb, force=force, overlap=overlap))
        else:
            out.append(FlatBucketReducer(ps, bucket_mb=bucket_mb, force=force))
    return tuple(out)


class StepLosses(collections.abc.MutableMapping):
    """The dict a training step returns (cubegan.py:181-187: loss_g, loss_t, loss_d, ..., lr), with the values read back from the device when somebody first
    LOOKS at them.  The reference hands its Lightning logger device tensors; this step used to end with four blocking `float()` read-backs and a device-wide
    status check — the GPU then idled while the host queued the head of the next step (inputs, conditioning recurrence, generator forward: ~5 ms at b = 16).
    Now the losses and the step's status words travel to page-locked memory in one asynchronous copy; `wait()` (any read access) waits for THAT copy only,
    raises if a split recurrence of the step gave up on a hand-off (its AdamW launches have skipped themselves on the device: optim.FlatAdamW.step(guard=...))
    and fills the dict.  A loop that logs the previous step's losses after queueing the next one never stalls the GPU."""

    def __init__(self, fetch):
        self._fetch = fetch
        self._d = None
        self._derive = []

    def also(self, fn):
        """fn(dict) -> dict of derived entries, evaluated when the values arrive"""
        if self._d is not None:
            self._d.update(fn(self._d))
        else:
            self._derive.append(fn)
        return self

    def wait(self):
        if self._d is None:
            fetch, self._fetch = self._fetch, None
            d = fetch()
            for fn in self._derive:
                d.update(fn(d))
            self._d = d
        return self._d

    @property
    def pending(self):
        return self._d is None

    def __getitem__(self, k):
        return self.wait()[k]

    def __setitem__(self, k, v):
        self.wait()[k] = v

    def __delitem__(self, k):
        del self.wait()[k]

    def __iter__(self):
        return iter(self.wait())

    def __len__(self):
        return len(self.wait())

    def __repr__(self):
        return repr(self.wait())


class _StepReadback:
    """per device: the step's status words (device) and a small ring of page-locked result slots"""
    _of = {}
    SLOTS = 4

    def __init__(self, dev):
        self.words = torch.zeros(4, dtype=torch.int32, device=dev)       # [generator side, text side, end of step, -]
        self.host = [torch.zeros(8, dtype=torch.float32).pin_memory() for _ in range(self.SLOTS)]
        self.owner = [None] * self.SLOTS
        self.n = 0

    @classmethod
    def of(cls, dev):
        key = dev.index if dev.index is not None else torch.cuda.current_device()
        st = cls._of.get(key)
        if st is None:
            st = cls._of[key] = cls(torch.device('cuda', key))
        return st

    def collect(self, slot, with_gemm=False):
        """status of everything launched on the CURRENT stream so far -> words[slot], by one launch on that stream"""
        if _lib.lib().ttsc_split_status_collect(_lib.current_stream(), self.words[slot:slot + 1].data_ptr(), int(with_gemm)) < 0:
            raise _lib.TTSCError('ttsc_split_status_collect: %s' % _lib.lib().ttsc_last_error().decode())

    def send(self, losses, extra):
        """queue the copy of [losses..., status words] to a page-locked slot on the current stream -> StepLosses"""
        i = self.n % self.SLOTS
        self.n += 1
        prev = self.owner[i]
        if prev is not None and prev.pending:
            prev.wait()      # a result nobody looked at for SLOTS steps: its copy finished long ago; a tripped guard must not get lost
        host = self.host[i]
        host.copy_(torch.cat([torch.stack([l.detach().float().reshape(()) for l in losses]), self.words.float()]), non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        names = list(extra.pop('_names'))

        def fetch():
            ev.synchronize()
            v = host.tolist()
            bad = [int(x) for x in v[len(names):len(names) + 3]]
            if any(bad):
                m = bad[0] | bad[1] | bad[2]
                kinds = '/'.join(n for b, n in ((1, 'LSTM'), (2, 'GRU'), (4, 'mel-AR'), (8, 'other')) if m & b)
                side = ', '.join(n for b, n in zip(bad, ('generator side', 'text side', 'end of step')) if b)
                if m & 15:
                    raise _lib.TTSCError('cubegan_training_step (%s): split %s recurrence aborted on a hand-off timeout — the AdamW update of that side was '
                                         'skipped on the device (are other kernels occupying the CUs? TTSC_LSTM_SPLIT=1 / TTSC_GRU_SPLIT=1 select the '
                                         'single-workgroup kernels)' % (side, kinds))
                raise _lib.TTSCError('cubegan_training_step: an operand of a split-precision GEMM lay beyond the fp16 range (|v| > 65504) or was not finite — '
                                     'its results are invalid; TTSC_GEMM_SPLIT=0 keeps these projections on the exact fp32 kernel')
            d = dict(zip(names, v))
            d.update(extra)
            return d
        res = StepLosses(fetch)
        self.owner[i] = res
        return res


def cubegan_training_step(model, batch, optimizers, reducers=None, rng=None):
    """Cubegan.training_step (cubegan.py:85-189): discriminator step, generator step (adv + feature + 45 x mel-L1),
    text step (duration CE + pitch/vuv L1); one gradient exchange per backward pass (reducers = (g, d, t))."""
    discriminator_loss, feature_loss, generator_loss = _gan_loss_fns()
    from ..io_utils import melspec as _melspec   # DFT / mel GEMMs + element-wise kernels on HIP, forward and backward
    mel_spectrogram = _melspec.mel_spectrogram
    arm = lambda i: reducers and hasattr(reducers[i], 'arm') and reducers[i].arm()   # bucket-ready hooks restart with every backward pass
    mpd, msd = _discriminator_fns(model)
    opt_g, opt_d, opt_t, opt_b = optimizers
    rng = rng or random
    dev = model.get_device()
    lang = model._languasito
    from ..optim import FlatAdamW
    lazy = STEP_LAZY and dev.type == 'cuda' and isinstance(opt_g, FlatAdamW) and isinstance(opt_t, FlatAdamW)

    def agreed(slot, reducer):
        # with a gradient exchange every rank must take the same decision about an update: the status word becomes the MAXIMUM over the ranks (one
        # 4-byte all-reduce, queued like the gradient collectives — nothing waits on the host).  It is issued AFTER the reducer's reduce(): all of
        # this reducer's chunks have been launched on every rank by then, hooked or not (distributed.ArenaReducer: one agreed launch sequence per
        # communicator), and no other backward pass — no other hook — runs between reduce() and here.  A rank whose recurrence gave up has sent
        # garbage into the sums; no rank applies them.
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(getattr(reducer, 'group', None)) > 1:
            dist.all_reduce(rb.words[slot:slot + 1], op=dist.ReduceOp.MAX, group=getattr(reducer, 'group', None))
        return rb.words[slot:slot + 1]
    rb = None
    if dev.type == 'cuda':
        from ..hifigan.wbank import AmaxPool
        AmaxPool.of(dev).reset()       # the convolution launches' range words of this step: one zeroing launch for all of them
        if lazy:
            rb = _StepReadback.of(dev)
            rb.words.zero_()
    # The text side (phoneme stack `t`, duration / pitch recurrences, their losses, backward pass and optimizer: opt_t) shares nothing but
    # inputs with the rest of the step, and it is a chain of latency-bound recurrences on a few dozen CUs.  It runs whole — forward, backward,
    # exchange, AdamW — on a stream of its own, under the discriminator / generator work (reference order: last, cubegan.py:172-180; the
    # parameter sets are disjoint, so the updates are the same).  TTSC_TEXT_STREAM=0 keeps it on the current stream.
    ph = PHASE_HOOK or (lambda name: None)
    ph('start')
    text_fn, cond_fn = _languasito_branches(lang, batch)
    cur = torch.cuda.current_stream(dev)
    ph('inputs')
    s_t = _text_stream(dev) if (TEXT_STREAM and dev.type == 'cuda') else None
    ev_inputs = cur.record_event() if s_t is not None else None     # the shared inputs are on the device from here on
    loss_text = None

    def text_side():
        # forward, losses and backward pass of the text side, queued on its stream.  WHEN the host queues it matters more than where it runs: queued
        # first (round 5: the reference's order has it last, cubegan.py:172-180) its recurrences ran alone on the chip for ~10 ms while the host had
        # not reached the generator yet, and the step's critical path — conditioning recurrence -> generator -> discriminators -> back — started
        # behind them.  Queued at `TEXT_AT` (default: after the discriminator step's backward pass has been queued) the main stream has a backlog
        # of chip-filling launches by then, the text recurrences run beside them on a few dozen CUs, and the host's ~10 ms of text-side queueing
        # comes out of the slack it has over the GPU in the discriminator / generator part (35 ms of queueing for 56 ms of kernels at b = 16).
        nonlocal loss_text
        if s_t is not None:
            s_t.wait_event(ev_inputs)          # (not wait_stream: the text side must not wait for the backlog it is meant to run beside)
            for t_ in text_fn.shared_inputs:      # allocated on the current stream, read (forward and backward) on the text stream
                if t_.is_cuda:
                    t_.record_stream(s_t)
        with (torch.cuda.stream(s_t) if s_t is not None else contextlib.nullcontext()):
            p_dur, p_pitch, p_vuv = text_fn()
            loss_duration, loss_pitch = text_losses(p_dur, p_pitch, p_vuv, _h2d(batch, 'y_dur', dev), _h2d(batch, 'y_pitch', dev), lang._max_pitch,
                                                    int(max(model._encodings.max_pitch, model._encodings.max_duration) + 1))
            loss_text = loss_pitch + loss_duration
            opt_t.zero_grad()
            arm(2)
            loss_text.backward()
        ph('text_queued')

    def text_update():
        # Exchange + AdamW of the text side, still on its stream — but only after THIS stream's split recurrences have reported that every
        # hand-off completed (they share the chip with the chip-filling discriminator / generator launches; a timed-out recurrence has
        # produced garbage gradients, which must neither reach the other ranks' sums nor AdamW: ADVICE r4).  The check waits for the text
        # stream alone, and it is made when the text backward has long finished: the host does not stall on it and nothing on the other
        # streams is drained.
        with (torch.cuda.stream(s_t) if s_t is not None else contextlib.nullcontext()):
            if lazy:
                rb.collect(1)                      # the same question asked and answered on the device: the update skips itself
                if reducers:
                    reducers[2].reduce()
                opt_t.step(guard=agreed(1, reducers[2]) if reducers else rb.words[1:2])
                return
            if dev.type == 'cuda':
                _lib.check_split_status('cubegan_training_step (text side, before its update)', stream=_lib.current_stream().value or 0)
            if reducers:
                reducers[2].reduce()
            opt_t.step()
    text_at = TEXT_AT if s_t is not None else 0
    if text_at == 0:
        text_side()
    conditioning = cond_fn()
    ph('cond_fwd')
    y = _h2d(batch, 'y_audio', dev)
    if y.shape[1] > 12000 - 240:   # random 50-frame / 12000-sample crop per item (cubegan.py:116-128)
        ys, cs = [], []
        for ii in range(y.shape[0]):
            max_frame = len(batch['y_frame2phone'][ii])
            r = rng.randint(0, max_frame - 50 - 1) if max_frame > 51 else 0
            cs.append(conditioning[ii, r:r + 50, :].unsqueeze(0))
            ys.append(y[ii, r * 240:r * 240 + 12000].unsqueeze(0))
        y = torch.cat(ys, dim=0)
        conditioning = torch.cat(cs, dim=0)
    y = y.unsqueeze(1)
    y_g_hat = generator_forward_with_grad(model._generator, conditioning.permute(0, 2, 1).contiguous())
    ph('gen_fwd')
    m = min(y.shape[2], y_g_hat.shape[2])
    y, y_g_hat = y[:, :, :m], y_g_hat[:, :, :m]
    y_mel = mel_spectrogram(y.squeeze(1), 1024, 80, 24000, 240, 1024, 0, 12000)
    y_g_hat_mel = mel_spectrogram(y_g_hat.squeeze(1), 1024, 80, 24000, 240, 1024, 0, 12000)
    ph('mels')
    opt_b.zero_grad()
    opt_d.zero_grad()
    arm(1)
    y_df_hat_r, y_df_hat_g, _, _ = mpd(y, y_g_hat.detach(), False)   # (the discriminator step reads no feature maps)
    loss_disc_f, _, _ = discriminator_loss(y_df_hat_r, y_df_hat_g)
    y_ds_hat_r, y_ds_hat_g, _, _ = msd(y, y_g_hat.detach(), False)
    loss_disc_s, _, _ = discriminator_loss(y_ds_hat_r, y_ds_hat_g)
    loss_disc_all = loss_disc_s + loss_disc_f
    ph('d_fwd')
    if text_at == 1:
        text_side()
    loss_disc_all.backward()
    ph('d_bwd')
    if text_at == 2:
        text_side()
    if reducers:
        reducers[1].reduce()
    opt_d.step()
    ph('d_opt')
    if text_at == 0:
        text_update()
    opt_g.zero_grad()
    arm(0)
    loss_mel = F.l1_loss(y_mel, y_g_hat_mel) * 45
    # the generator step only needs the discriminators' DATA gradients: the reference also accumulates their weight
    # gradients here and throws them away at the next opt_d.zero_grad() (cubegan.py:137,157-170); freezing the
    # discriminator parameters for this pass skips a third of its backward work with identical updates
    d_params = [p for p in itertools.chain(model._mpd.parameters(), model._msd.parameters()) if p.requires_grad]
    for p in d_params:
        p.requires_grad_(False)
    try:
        # the feature maps are only ever read by the feature-matching loss: the native pair hands them over un-activated and the loss launch applies
        # the leaky-relu (no materialised copies, no activation backward launches: ~130 launches of a step); substituted formulations get tensors
        fm = 'raw' if (FMAP_RAW and all(getattr(f, 'accepts_raw', False) for f in (mpd, msd, feature_loss))) else True
        y_df_hat_r, y_df_hat_g, fmap_f_r, fmap_f_g = mpd(y, y_g_hat, fm)
        y_ds_hat_r, y_ds_hat_g, fmap_s_r, fmap_s_g = msd(y, y_g_hat, fm)
        loss_gen_all = (generator_loss(y_ds_hat_g)[0] + generator_loss(y_df_hat_g)[0] + feature_loss(fmap_s_r, fmap_s_g)
                        + feature_loss(fmap_f_r, fmap_f_g) + loss_mel)
        ph('g_fwd')
        if text_at == 3:
            text_side()
        loss_gen_all.backward()          # (the reference retains the graph for the text loss; here that graph is a separate one)
        ph('g_bwd')
        if text_at == 4:
            text_side()
    finally:
        for p in d_params:
            p.requires_grad_(True)
    if lazy:
        rb.collect(0)
        if reducers:
            reducers[0].reduce()
        opt_g.step(guard=agreed(0, reducers[0]) if reducers else rb.words[0:1])
    else:
        if dev.type == 'cuda':
            # the generator side holds the `g` phoneme stack's split recurrences: the same question before ITS exchange and update (the host waits
            # for the current stream here)
            _lib.check_split_status('cubegan_training_step (generator side, before its update)', stream=_lib.current_stream().value or 0)
        if reducers:
            reducers[0].reduce()
        opt_g.step()
    opt_b.step()
    ph('g_opt')
    if text_at != 0:
        text_update()
    if s_t is not None:
        cur.wait_stream(s_t)          # the step ends when both sides have
    ph('joined')
    if not lazy:
        _lib.check_split_status('cubegan_training_step')
    model._global_step += 1
    model._current_lr = model._compute_lr(model._learning_rate, 1e-5, model._global_step)
    for o in (opt_d, opt_g, opt_t):
        o.param_groups[0]['lr'] = model._current_lr
    if lazy:
        rb.collect(2, with_gemm=True)     # whatever ran after the two guarded updates, and the split-precision GEMMs' range word
        res = rb.send([loss_gen_all, loss_text, loss_disc_all, loss_mel], {'_names': ('loss_g', 'loss_t', 'loss_d', 'loss_mel'), 'lr': model._current_lr})
        res.also(lambda d: {'loss_mel': d['loss_mel'] / 45})
        ph('end')
        return res
    ph('end')
    vals = torch.stack([loss_gen_all.detach(), loss_text.detach(), loss_disc_all.detach(), loss_mel.detach()]).tolist()   # ONE read-back
    return {'loss_g': vals[0], 'loss_t': vals[1], 'loss_d': vals[2], 'loss_mel': vals[3] / 45, 'lr': model._current_lr}


def cubegan_validation_step(model, batch, rng=None):
    """Cubegan.validation_step (cubegan.py:191-273) on the inference kernels (no autograd): teacher-forced Languasito2 with the
    dev item's own alignments and pitch, a 200-frame / 48 000-sample crop per item, generator forward, mel-L1.  Returns
    loss_mel — the quantity `validation_epoch_end` averages into `_val_loss`, i.e. what `.best` is selected on
    (train_cubegan.py:38-76) — plus the duration / pitch losses.  (The adversarial terms of the reference's validation dict are
    logged, never selected on; they are not evaluated here.)"""
    from ..io_utils.melspec import mel_spectrogram
    rng = rng or random
    dev = model.get_device()
    lang = model._languasito
    with torch.no_grad():
        p_dur, p_pitch, p_vuv, conditioning = languasito_forward(lang, batch)
        t_dur = batch['y_dur'].to(dev)
        t_pitch = batch['y_pitch'].to(dev).float()
        t_vuv = (t_pitch > 1).float()
        m = min(t_dur.shape[1], p_dur.shape[1])
        t_dur, p_dur = t_dur[:, :m], p_dur[:, :m, :]
        m = min(t_pitch.shape[1], p_pitch.shape[1])
        t_pitch, p_pitch, t_vuv, p_vuv = t_pitch[:, :m], p_pitch[:, :m], t_vuv[:, :m], p_vuv[:, :m]
        ignore = int(max(model._encodings.max_pitch, model._encodings.max_duration) + 1)
        loss_duration = F.cross_entropy(p_dur.reshape(-1, p_dur.shape[2]), t_dur.reshape(-1), ignore_index=ignore)
        loss_pitch = (torch.abs(t_pitch / lang._max_pitch - p_pitch) * t_vuv).mean() + torch.abs(t_vuv - p_vuv).mean()
        y = batch['y_audio'].to(dev)
        if y.shape[1] > 48000 - 240:
            ys, cs = [], []
            for ii in range(y.shape[0]):
                max_frame = len(batch['y_frame2phone'][ii])
                r = rng.randint(0, max_frame - 200 - 1) if max_frame > 201 else 0
                cs.append(conditioning[ii, r:r + 200, :].unsqueeze(0))
                ys.append(y[ii, r * 240:r * 240 + 48000].unsqueeze(0))
            m = min(c.shape[1] for c in cs)
            conditioning = torch.cat([c[:, :m] for c in cs], dim=0)
            y = torch.cat([v[:, :m * 240] for v in ys], dim=0)
        y_g_hat = model._generator(conditioning.permute(0, 2, 1).contiguous())
        m = min(y.shape[1], y_g_hat.shape[2])
        y_mel = mel_spectrogram(y[:, :m], 1024, 80, 24000, 240, 1024, 0, 12000)
        y_g_hat_mel = mel_spectrogram(y_g_hat[:, 0, :m], 1024, 80, 24000, 240, 1024, 0, 12000)
        loss_mel = F.l1_loss(y_mel, y_g_hat_mel)
    return {'loss_mel': float(loss_mel), 'loss_t': float(loss_pitch + loss_duration)}


def wavernn_logits_train(net, X):
    """Differentiable WaveRNN._train_forward (modules.py:505-539): the GRU(s) run on the persistent HIP forward / backward
    kernels with their weight / input gradients on the MFMA GEMMs (gru_autograd.py), the low-resolution conditioning convolutions on
    the HIP convolution kernels, the two output Linears on the MFMA GEMM (text_autograd.hip_linear); repeat / interpolate / concat
    are data movement."""
    mel, gs_x = X['mel'], X['x']
    _require_device(mel, 'wavernn_logits_train')
    up = mel.repeat_interleave(net._upsample, dim=1)
    if net._use_lowres:
        low_x = X['x_low']
        interp = F.interpolate(low_x.unsqueeze(1), net._upsample_low * low_x.shape[1], mode='linear').squeeze(1)
        hidden = low_x.unsqueeze(1)
        hidden = _lowres_features(net, hidden)
        ux = hidden.repeat_interleave(net._upsample_low, dim=2).permute(0, 2, 1)
        m = min(up.shape[1], gs_x.shape[1], ux.shape[1], interp.shape[1])
        hidden = torch.cat([up[:, :m], ux[:, :m], interp[:, :m].unsqueeze(2), gs_x[:, :m].unsqueeze(2)], dim=-1)
    else:
        m = min(up.shape[1], gs_x.shape[1])
        hidden = torch.cat([up[:, :m], gs_x[:, :m].unsqueeze(2)], dim=-1)
    for rnn in net._rnns:
        hidden = gru_forward_train(rnn, hidden)
    return _output_linears(net, hidden)


def wavernn_train_loss(net, batch):
    """WaveRNN.training_step (modules.py:553-563): target shifted right by one with a 0, CE on the encoded target."""
    gs = batch['x']
    b = dict(batch)
    b['x'] = F.pad(gs[:, :-1], (1, 0), mode='constant', value=0)
    out = wavernn_logits_train(net, b)
    return net._output_functions.loss(out, gs[:, :out.shape[1]])


def vocoder_training_step(voc, batch, optimizers, reducers=None):
    """CubenetVocoder.training_step (vocoder.py:136-156): two independent networks, clip_grad_norm 5, Adam x2,
    lr = lr0 / (1 + 5e-5 * step)."""
    opt_lr, opt_hr = optimizers
    dev = voc._wavernn_hr._get_device()
    batch = {k: v.to(dev) for k, v in batch.items()}
    opt_lr.zero_grad()
    opt_hr.zero_grad()
    loss_hr = wavernn_train_loss(voc._wavernn_hr, {'x': batch['x'], 'x_low': batch['x_low'], 'mel': batch['mel']})
    loss_lr = wavernn_train_loss(voc._wavernn_lr, {'x': batch['x_low'], 'mel': batch['mel']})
    loss_lr.backward()
    loss_hr.backward()
    # a split recurrence that timed out on a hand-off has produced garbage gradients: find out BEFORE they are exchanged with the other ranks and
    # applied — the step already synchronises here for the gradient norms, so the check costs nothing extra
    _lib.check_split_status('vocoder_training_step (before the update)')
    if reducers:
        reducers[0].reduce()
        reducers[1].reduce()
    torch.nn.utils.clip_grad_norm_(voc._wavernn_lr.parameters(), 5)
    torch.nn.utils.clip_grad_norm_(voc._wavernn_hr.parameters(), 5)
    opt_lr.step()
    opt_hr.step()
    _lib.check_split_status('vocoder_training_step')
    voc._global_step += 1
    alpha = voc._compute_lr(voc._learning_rate, 5e-5, voc._global_step)
    opt_lr.param_groups[0]['lr'] = alpha
    opt_hr.param_groups[0]['lr'] = alpha
    return {'lr': float(loss_lr.detach()), 'hr': float(loss_hr.detach()), 'loss': float((loss_hr + loss_lr).detach()) / 2, 'alpha': alpha}
