"""Per-module weight preparation of the training step in three launches (csrc/conv_train.hip: ttsc_wbank_*).

The reference gets the effective weights of its weight-normed convolutions from torch.nn.utils.weight_norm's pre-forward hooks, one small
element-wise launch chain per layer and call ([EXTERNAL hifigan/models.py]; call sites cube/networks/cubegan.py:131,144-149,160-167).  The
first native version of this step did the same per convolution LAUNCH: weight norm, (strided layers) tap de-interleave, a range reduction and
a fragment packing for the forward call, another packing for the data gradient — 388 + 190 + 120 launches per step plus half of 392 range
reductions (profiles/r05_train_last300ms_kernel_stats.csv), each a few microseconds of work.  A `WeightBank` owns persistent buffers for
every layer of a module (generator; MultiPeriodDiscriminator; MultiScaleDiscriminator) and refills all of them with `prepare()`:
one memset, one launch for the norms / effective weights / ranges, one launch for both fragment orders of every layer.

`weight(i)` is the autograd handle of layer i's effective weight: its backward runs the (de-interleave adjoint and) weight-norm backward
kernels for that layer; the convolution functions (autograd.HipConvFn) take the prepared fragments from its `_ttsc_pack` attribute."""
import ctypes as C

import torch

from .. import _lib


REUSE = __import__('os').environ.get('TTSC_WBANK_REUSE', '1') != '0'   # (measurement switch: 0 = prepare on every call)


class _Entry:
    __slots__ = ('layer', 'Cin', 'Cout', 'K', 'groups', 'stride', 'J', 'w', 'norm', 'amax', 'pack_fwd', 'pack_dgrad', 'conv_shape', 'placeholder')


class BankWeightFn(torch.autograd.Function):
    """layer i's effective weight as the convolution sees it ([Cout, stride * Cin / groups, ceil(K / stride)]); the values live in the bank's
    fragment buffers — for a strided layer the returned tensor is a zero-memory placeholder of that shape that nothing reads"""

    @staticmethod
    def forward(ctx, v, g, bank, i):
        e = bank.entries[i]
        ctx.bank, ctx.i = bank, i
        ctx.save_for_backward(v, g)
        ctx.norm = e.norm                      # (refilled by the next prepare(): backward passes of a step run before it)
        return e.w.view(e.conv_shape) if e.stride == 1 else e.placeholder.expand(e.conv_shape)

    @staticmethod
    def backward(ctx, dwp):
        v, g = ctx.saved_tensors
        e = ctx.bank.entries[ctx.i]
        L = _lib.lib()
        dwp = dwp.contiguous()
        Cg = e.Cin // e.groups
        with _lib.on_device(v.device):
            if e.stride > 1:                   # adjoint of the tap de-interleave: dw[co, ci, k] = dw'[co, (k % s, ci), k // s]
                dw = torch.empty((e.Cout, Cg, e.K), dtype=torch.float32, device=v.device)
                _lib.check(L.ttsc_deinterleave_w(_lib.dev_ptr(dwp), _lib.dev_ptr(dw), e.Cout, Cg, e.K, e.stride, 1, _lib.current_stream()),
                           'ttsc_deinterleave_w')
            else:
                dw = dwp
            dv = torch.empty_like(v)
            dg = torch.empty_like(g)
            _lib.check(L.ttsc_weight_norm_backward(_lib.dev_ptr(dw), _lib.dev_ptr(v), _lib.dev_ptr(g), _lib.dev_ptr(ctx.norm), _lib.dev_ptr(dv),
                                                   _lib.dev_ptr(dg), e.Cout, Cg * e.K, _lib.current_stream()), 'ttsc_weight_norm_backward')
        return dv, dg, None, None


class WeightBank:
    def __init__(self, specs):
        """specs: list of (layer, Cin, Cout, K, groups, stride) for weight-normed layers (`layer.weight_v` [Cout, Cin / groups, K(, 1)],
        `layer.weight_g`); every (stride * Cin, Cout, ceil(K / stride), groups) must be a shape ttsc_conv_train takes in both directions"""
        self.entries = []
        for layer, Cin, Cout, K, groups, stride in specs:
            e = _Entry()
            e.layer, e.Cin, e.Cout, e.K, e.groups, e.stride = layer, Cin, Cout, K, groups, stride
            e.J = -(-K // stride)
            e.conv_shape = (Cout, stride * (Cin // groups), e.J)
            self.entries.append(e)
        self._handle = None
        self._ptrs = None
        self._bufs = None
        self._sig = None          # (parameter versions, stream) of the last preparation

    def __del__(self):
        try:
            if self._handle is not None:
                _lib.lib().ttsc_wbank_destroy(self._handle)
        except Exception:
            pass

    def _build(self, dev):
        L = _lib.lib()
        n = len(self.entries)
        if self._bufs is None or self._bufs[0].device != dev:
            # one allocation per kind: [range words | norms | effective weights], fragments of both orders
            nw = sum(e.Cout * (e.Cin // e.groups) * e.K for e in self.entries)
            nn_ = sum(e.Cout for e in self.entries)
            fb = [int(L.ttsc_conv_train_workspace_bytes(e.stride * e.Cin, e.Cout, e.J, e.groups)) - 256 for e in self.entries]
            db = [int(L.ttsc_conv_train_workspace_bytes(e.Cout, e.stride * e.Cin, e.J, e.groups)) - 256 for e in self.entries]
            words = torch.zeros(((n + 63) // 64) * 64, dtype=torch.float32, device=dev)
            norms = torch.empty(nn_, dtype=torch.float32, device=dev)
            wbuf = torch.empty(nw, dtype=torch.float32, device=dev)
            up = lambda b: (b + 255) // 256 * 256
            frag = torch.empty(sum(up(b) for b in fb) + sum(up(b) for b in db), dtype=torch.uint8, device=dev)
            self._bufs = (words, norms, wbuf, frag)
            ow = on = of = 0
            for i, e in enumerate(self.entries):
                e.amax = words[i:i + 1]
                e.norm = norms[on:on + e.Cout]
                cnt = e.Cout * (e.Cin // e.groups) * e.K
                e.w = wbuf[ow:ow + cnt]
                e.pack_fwd = frag[of:of + fb[i]]
                of += up(fb[i])
                e.pack_dgrad = frag[of:of + db[i]]
                of += up(db[i])
                on += e.Cout
                ow += cnt
                e.placeholder = torch.empty((1, 1, 1), dtype=torch.float32, device=dev)
        tab = (_lib.WBankEntry * n)()
        for i, e in enumerate(self.entries):
            t = tab[i]
            t.v, t.g = e.layer.weight_v.data_ptr(), e.layer.weight_g.data_ptr()
            t.w, t.norm, t.amax = e.w.data_ptr(), e.norm.data_ptr(), e.amax.data_ptr()
            t.pack_fwd, t.pack_dgrad = e.pack_fwd.data_ptr(), e.pack_dgrad.data_ptr()
            t.Cin, t.Cout, t.K, t.groups, t.stride = e.Cin, e.Cout, e.K, e.groups, e.stride
        if self._handle is not None:
            L.ttsc_wbank_destroy(self._handle)
            self._handle = None
        h = C.c_void_p()
        with _lib.on_device(dev):
            _lib.check(L.ttsc_wbank_create(tab, n, C.byref(h)), 'ttsc_wbank_create')
        self._handle = h

    def prepare(self):
        """refill every buffer from the live parameters, on the current stream (three launches).  The parameter addresses are re-read every
        time: an optimizer that moves the parameters into a flat arena (optim.FlatAdamW) changes them once."""
        ps = tuple((e.layer.weight_v.data_ptr(), e.layer.weight_g.data_ptr()) for e in self.entries)
        dev = self.entries[0].layer.weight_v.device
        if dev.type != 'cuda':
            raise _lib.TTSCError('WeightBank: parameters live on the CPU; move the module to a HIP device (no CPU path)')
        if ps != self._ptrs or self._bufs is None or self._bufs[0].device != dev:
            for e in self.entries:
                if not (e.layer.weight_v.is_contiguous() and e.layer.weight_g.is_contiguous() and e.layer.weight_v.dtype == torch.float32):
                    raise _lib.TTSCError('WeightBank: parameters must be contiguous fp32 tensors')
            self._build(dev)
            self._ptrs = ps
            self._sig = None
        # Unchanged parameters -> the buffers still hold what this call would write: the discriminators' banks are prepared for the generator step
        # AFTER opt_d.step() and again for the next step's discriminator pass, with nothing written in between (two of the four large
        # preparations of a Cubegan step; FlatAdamW.step and every in-place torch op move the version counters).  The skip needs the last
        # preparation to be ordered before this call's consumers: same stream (side streams fork behind the current one).
        sig = (tuple(e.layer.weight_v._version for e in self.entries), tuple(e.layer.weight_g._version for e in self.entries),
               _lib.current_stream().value)
        if REUSE and sig == self._sig:
            return
        with _lib.on_device(dev):
            _lib.check(_lib.lib().ttsc_wbank_prepare(self._handle, _lib.current_stream()), 'ttsc_wbank_prepare')
        self._sig = sig

    def invalidate(self):
        """after a write to the parameters that bypassed their version counters (through `.data`, or a raw-pointer kernel that did not call
        torch.autograd.graph.increment_version): the next prepare() refills the buffers"""
        self._sig = None

    def weight(self, i):
        e = self.entries[i]
        v, g = e.layer.weight_v, e.layer.weight_g
        if (v.requires_grad or g.requires_grad) and torch.is_grad_enabled():
            w = BankWeightFn.apply(v, g, self, i)
        else:   # frozen layer (the discriminators during the generator step): no autograd node, just the handle that carries the fragments
            w = e.w.view(e.conv_shape) if e.stride == 1 else e.placeholder.expand(e.conv_shape)
        w._ttsc_pack = (e.pack_fwd, e.pack_dgrad, e.amax)
        return w


class AmaxPool:
    """Range words for the convolution launches of a step, zeroed with ONE launch: `take()` hands out four fresh words of one convolution call —
    [max |x| (when measured by this call), max |y| (left by the forward launch's epilogue), max |dy| (when measured), max |dx| (left by the
    data-gradient launch)]; `reset()` (start of a step, on the main stream before any side stream forks) zeroes the pool and starts over.  A pool
    that runs dry, or that nobody resets, falls back to a fresh zeroed tensor per call.

    A tensor whose producing launch left its maximum in a word carries it as `_ttsc_amax` (`tag_range`); `range_of` hands the word to the next
    launch that reads the tensor — unless the tensor was written since (autograd accumulates a second gradient INTO the first one in place: the
    version counter moved) or the pool was reset (the word belongs to a later step now)."""
    _pools = {}

    def __init__(self, dev, slots=4096):
        self.buf = torch.zeros(slots * 4, dtype=torch.float32, device=dev)
        self.next = 0
        self.slots = slots
        self.epoch = 0

    @classmethod
    def of(cls, dev):
        key = dev.index if dev.index is not None else torch.cuda.current_device()
        p = cls._pools.get(key)
        if p is None:
            p = cls._pools[key] = cls(torch.device('cuda', key))
        return p

    def reset(self):
        if self.next:
            self.buf[:self.next * 4].zero_()
        self.next = 0
        self.epoch += 1

    def take(self):
        if self.next >= self.slots:
            return torch.zeros(4, dtype=torch.float32, device=self.buf.device)
        i = self.next
        self.next += 1
        return self.buf[4 * i:4 * i + 4]


PROPAGATE = __import__('os').environ.get('TTSC_TRAIN_AMAX_PROPAGATE', '1') != '0'   # (measurement switch: 0 = every launch reduces its input itself)


def tag_range(t, word, pool):
    """`t` was just produced by a launch that left max |t| in `word` (a one-element view of `pool`'s buffer, or of a fresh tensor)"""
    if PROPAGATE:
        t._ttsc_amax = (word, t._version, pool.epoch, pool)
    return t


def range_of(t):
    """the word holding max |t| (an upper bound is enough), or None when nobody left one or it can no longer be trusted"""
    a = getattr(t, '_ttsc_amax', None)
    if a is None:
        return None
    word, version, epoch, pool = a
    if t._version != version or pool.epoch != epoch:
        return None
    return word


def pass_range(src, dst):
    """dst holds a permutation / crop / zero-padding of src's elements (de-interleave, its adjoint): the same bound serves"""
    a = getattr(src, '_ttsc_amax', None)
    if a is not None and src._version == a[1] and a[3].epoch == a[2]:
        dst._ttsc_amax = (a[0], dst._version, a[2], a[3])
    return dst
