"""Which layers does the Cubegan training step (b = 16) still run on the exact-fp32 convolution / weight-gradient kernels, and what do they
cost?  Wraps the python entry points with device-event timers (synchronising: the totals are per-call kernel times, not step time).
    python tools/probes/prof_train_convs.py"""
import collections
import os
import random
import sys

import torch

sys.path.insert(0, os.getcwd())
from ttscube_amd import hip_layers  # noqa: E402
from ttscube_amd.hifigan import autograd as AG  # noqa: E402
from ttscube_amd.io_utils.io_cubegan import CubeganCollate  # noqa: E402
from ttscube_amd.io_utils.synthetic import synthetic_encodings, synthetic_examples  # noqa: E402
from ttscube_amd.networks import training as T  # noqa: E402
from ttscube_amd.networks.cubegan import Cubegan  # noqa: E402

acc = collections.defaultdict(lambda: [0, 0.0])
ON = [False]


def timed(key, fn, *a, **kw):
    if not ON[0]:
        return fn(*a, **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = fn(*a, **kw)
    e1.record()
    torch.cuda.synchronize()
    acc[key][0] += 1
    acc[key][1] += e0.elapsed_time(e1)
    return r


_call = hip_layers.Conv1dHip.__call__
hip_layers.Conv1dHip.__call__ = lambda self, x, *a, **kw: timed(
    ('fp32 conv', self.cfg.in_channels, self.cfg.out_channels, self.cfg.kernel_size, self.cfg.stride, self.cfg.dilation, self.groups,
     int(self.cfg.transposed), tuple(x.shape)), _call, self, x, *a, **kw)
_split = AG._conv_split
AG._conv_split = lambda x, w, b, resid, gate, Cin, Cout, K, padding, dilation, flip, **kw: timed(
    ('split conv', Cin, Cout, K, 1, dilation, kw.get('groups', 1), flip, tuple(x.shape)), _split, x, w, b, resid, gate, Cin, Cout, K, padding, dilation, flip, **kw)
_wg = AG._wgrad
AG._wgrad = lambda P, Q, A, Bc, J, base, step, qs, ql, groups=1, **kw: timed(
    ('wgrad' + (' split' if AG.SPLIT_TRAIN and groups == 1 and AG._lib.lib().ttsc_conv_wgrad_split_supported(A, Bc, J, step) else ' fp32'), A, Bc, J,
     step, groups, tuple(P.shape)), _wg, P, Q, A, Bc, J, base, step, qs, ql, groups, **kw)

dev = torch.device('cuda', 0)
enc = synthetic_encodings()
torch.manual_seed(1234)
model = Cubegan(enc, conditioning=None, train=True).to(dev)
model.train()
opts = T.cubegan_configure_optimizers(model)
batch = CubeganCollate(enc).collate_fn(list(synthetic_examples(16, 777, min_ph=30, max_ph=50)))
crop = random.Random(99)
for _ in range(2):
    T.cubegan_training_step(model, batch, opts, None, rng=crop)
ON[0] = True
STEPS = 2
for _ in range(STEPS):
    T.cubegan_training_step(model, batch, opts, None, rng=crop)
tot = collections.defaultdict(float)
for k, (n, ms) in acc.items():
    tot[k[0]] += ms / STEPS
print('per step: ' + ', '.join('%s %.2f ms' % kv for kv in sorted(tot.items())))
for k, (n, ms) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:45]:
    print('%7.3f ms/step %4d calls/step  %s' % (ms / STEPS, n // STEPS, k))
