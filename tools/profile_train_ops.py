#!/usr/bin/env python
"""Where do the launches of one Cubegan training step come from?  torch.profiler over two warm steps, aggregated by operator
and by the innermost ttscube_amd source line (launch-count view: the step is host-bound when launches x ~15 us > GPU time).

    python tools/profile_train_ops.py [--batch 16] > ops.txt"""
import argparse
import collections
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=16)
    a = ap.parse_args()
    import torch
    from torch.profiler import ProfilerActivity, profile
    from ttscube_amd.io_utils.io_cubegan import CubeganCollate
    from ttscube_amd.io_utils.synthetic import synthetic_encodings, synthetic_examples
    from ttscube_amd.networks import training as T
    from ttscube_amd.networks.cubegan import Cubegan
    dev = torch.device('cuda', 0)
    enc = synthetic_encodings()
    torch.manual_seed(1234)
    model = Cubegan(enc, conditioning=None, train=True).to(dev)
    model.train()
    opts = T.cubegan_configure_optimizers(model)
    batch = CubeganCollate(enc).collate_fn(list(synthetic_examples(a.batch, 777, min_ph=30, max_ph=50)))
    crop = random.Random(99)
    for _ in range(3):
        T.cubegan_training_step(model, batch, opts, None, rng=crop)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        for _ in range(2):
            T.cubegan_training_step(model, batch, opts, None, rng=crop)
        torch.cuda.synchronize()
    ev = prof.events()
    by_op = collections.Counter()
    by_line = collections.Counter()
    kern_by_line = collections.Counter()
    for e in ev:
        if e.device_type != torch.autograd.DeviceType.CPU or not e.name.startswith('aten::'):
            continue
        nk = len(e.kernels)
        if nk == 0:
            continue
        # only leaf operators launch kernels themselves; parents list the same kernels again -> keep ops without aten children launching
        if any(c.name.startswith('aten::') and len(c.kernels) for c in e.cpu_children):
            continue
        by_op[e.name] += nk
        src = '?'
        for fr in (e.stack or []):
            if 'ttscube_amd' in fr and 'site-packages' not in fr:
                src = fr.split('ttscube_amd/')[-1]
                break
        by_line[src] += nk
        kern_by_line[(src, e.name)] += nk
    print('kernel launches per step by leaf aten op:')
    for k, v in by_op.most_common(25):
        print('  %6d  %s' % (v // 2, k))
    print('kernel launches per step by innermost ttscube_amd frame:')
    for k, v in by_line.most_common(45):
        tops = sorted(((n, c) for (s, n), c in kern_by_line.items() if s == k), key=lambda t: -t[1])[:3]
        print('  %6d  %-60s %s' % (v // 2, k[:60], ', '.join('%s x%d' % (n.replace('aten::', ''), c // 2) for n, c in tops)))


if __name__ == '__main__':
    main()
