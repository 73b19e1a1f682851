"""GAN loss terms of the Cubegan training step on ONE HIP launch per list of tensors (csrc/train_ops.hip::gan_loss_kernel):
`feature_loss`, `generator_loss`, `discriminator_loss` of hifigan.models [EXTERNAL; call sites cube/networks/cubegan.py:144-149,
160-167].  The torch-op formulations in `discriminators.py` walk ~50 tensors with a mean / sub / abs-or-square / add each, forward
and backward (~400 launches per step); here value AND gradient of a whole list come out of one kernel, the backward pass only
scales the stored gradient by the incoming scalar.  Same public names and return shapes as the torch-op versions (the per-term
loss lists, which the reference only logs, are returned empty)."""
import ctypes as C

import torch

from .. import _lib


class _ListLoss(torch.autograd.Function):
    """kind 0: sum_k w_k mean|a_k - b_k| over pairs (a = first half of `tensors`, b = second half);
    kind 1: sum_k w_k mean (a_k - target)^2."""

    @staticmethod
    def forward(ctx, kind, target, weight, *tensors):
        slopes = None
        if isinstance(weight, tuple):      # (weight, per-pair leaky-relu slopes): kind 0 over pre-activations (ttsc_gan_loss_lrelu)
            weight, slopes = weight
        n = len(tensors) if kind == 1 else len(tensors) // 2
        a = [t.contiguous() for t in tensors[:n]]
        b = [t.contiguous() for t in tensors[n:]] if kind == 0 else []
        dev = a[0].device
        if dev.type != 'cuda':
            raise _lib.TTSCError('GAN losses: tensors must live on a HIP device (got %s); no CPU path' % dev)
        need_a = [ctx.needs_input_grad[3 + i] for i in range(n)]
        need_b = [ctx.needs_input_grad[3 + n + i] for i in range(n)] if kind == 0 else []
        # one flat gradient buffer, carved into per-tensor views
        sizes = [t.numel() for t in a]
        tot = sum(s for s, k in zip(sizes, need_a) if k) + sum(s for s, k in zip(sizes, need_b) if k)
        gbuf = torch.empty(max(tot, 1), dtype=torch.float32, device=dev)
        ga, gb, off = [None] * n, [None] * n, 0
        for i in range(n):
            if need_a[i]:
                ga[i] = gbuf[off:off + sizes[i]].view_as(a[i])
                off += sizes[i]
        for i in range(n if kind == 0 else 0):
            if need_b[i]:
                gb[i] = gbuf[off:off + sizes[i]].view_as(b[i])
                off += sizes[i]
        out = torch.empty((), dtype=torch.float32, device=dev)
        L = _lib.lib()
        ws = torch.empty(int(L.ttsc_gan_loss_workspace_bytes(n)), dtype=torch.uint8, device=dev)
        P = C.c_void_p * n
        ptrs = lambda ts: P(*[(t.data_ptr() if t is not None else None) for t in ts])
        with _lib.on_device(dev):
            _lib.check(L.ttsc_gan_loss_lrelu(kind, n, ptrs(a), ptrs(b) if kind == 0 else None, ptrs(ga), ptrs(gb) if kind == 0 else None,
                                             (C.c_int64 * n)(*sizes), (C.c_float * n)(*([float(weight)] * n)),
                                             (C.c_float * n)(*[float(v) for v in slopes]) if slopes is not None else None, float(target),
                                             _lib.dev_ptr(out), _lib.dev_ptr(ws), ws.numel(), _lib.current_stream()), 'ttsc_gan_loss')
        ctx.grads = ga + (gb if kind == 0 else [])
        ctx.gbuf = gbuf
        return out

    @staticmethod
    def backward(ctx, go):
        # ONE launch scales every stored gradient by the incoming scalar — in place, so the node can be differentiated ONCE: a second pass
        # through a retained graph (the generator step keeps its graph for the text step) would compound the factor silently
        if getattr(ctx, 'consumed', False):
            raise RuntimeError('_ListLoss: second backward pass through the same loss node (its stored gradients were scaled in place)')
        ctx.consumed = True
        ctx.gbuf.mul_(go)
        return (None, None, None) + tuple(ctx.grads)


def _flat(list_of_lists):
    return [t for sub in list_of_lists for t in sub]


class RawFmap:
    """a discriminator feature map as the convolution left it: the feature map proper is leaky_relu(x, slope) (slope 1: x itself).  What
    disc_hip.mpd_forward / msd_forward hand out with want_fmap='raw'; `feature_loss` applies the activation inside its one launch."""
    __slots__ = ('x', 'slope')

    def __init__(self, x, slope):
        self.x, self.slope = x, slope

    def materialize(self):
        return torch.nn.functional.leaky_relu(self.x, self.slope) if self.slope != 1.0 else self.x


def feature_loss(fmap_r, fmap_g):
    r, g = _flat(fmap_r), _flat(fmap_g)
    if r and isinstance(r[0], RawFmap):
        if any(a.slope != b.slope for a, b in zip(r, g)):
            raise _lib.TTSCError('feature_loss: real and generated feature maps of a layer carry different activations')
        return _ListLoss.apply(0, 0.0, (2.0, tuple(a.slope for a in r)), *([a.x for a in r] + [b.x for b in g]))
    return _ListLoss.apply(0, 0.0, 2.0, *(r + g))


feature_loss.accepts_raw = True


def generator_loss(disc_outputs):
    return _ListLoss.apply(1, 1.0, 1.0, *disc_outputs), []


def discriminator_loss(disc_real_outputs, disc_generated_outputs):
    loss = _ListLoss.apply(1, 1.0, 1.0, *disc_real_outputs) + _ListLoss.apply(1, 0.0, 1.0, *disc_generated_outputs)
    return loss, [], []
