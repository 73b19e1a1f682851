"""Host and GPU time of the phases of one Cubegan training step, untraced: at every phase boundary of cubegan_training_step (training.PHASE_HOOK) the
host clock is read and an event is recorded on the stream that is current there; per phase the table shows when the HOST finished queueing it and
when the GPU finished executing what was queued up to there (both relative to the step's start, mean over the steps).  A phase whose GPU time
trails the host time by a few hundred microseconds is host-bound (the GPU waits for launches); a growing gap is a backlog (the GPU is the bound).
    python tools/probes/train_phase_timeline.py [--batch 16] [--iters 8]"""
import argparse
import os
import random
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=16)
    ap.add_argument('--iters', type=int, default=8)
    a = ap.parse_args()
    from bench_cubegan_step import make_batch
    from ttscube_amd.networks.cubegan import Cubegan
    from ttscube_amd.networks import training as T
    rng = np.random.RandomState(0)
    batch, enc = make_batch(a.batch, 40, rng)
    torch.manual_seed(0)
    model = Cubegan(enc, conditioning=None, train=True).cuda()
    model.train()
    opts = T.cubegan_configure_optimizers(model)
    r = random.Random(1)
    for _ in range(3):
        T.cubegan_training_step(model, batch, opts, rng=r)
    torch.cuda.synchronize()
    rec = []

    def hook(name):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(torch.cuda.current_stream())
        rec.append((name, time.perf_counter(), ev))
    T.PHASE_HOOK = hook
    t_all = time.perf_counter()
    marks = []
    for _ in range(a.iters):
        marks.append(len(rec))
        T.cubegan_training_step(model, batch, opts, rng=r)      # (no synchronisation between steps: with TTSC_STEP_LAZY=1 the host runs ahead)
        rec.append(('returned', time.perf_counter(), None))
    marks.append(len(rec))
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t_all) / a.iters * 1e3
    # reference points: host clock and GPU clock are tied together at the first event of the first step (the device was idle there)
    t00, e00 = rec[0][1], rec[0][2]
    steps = []
    for k in range(1, a.iters):          # (step 0 starts on an idle device; the steady state is what the others show)
        seg = rec[marks[k]:marks[k + 1]]
        t0 = seg[0][1]
        g0 = e00.elapsed_time(seg[0][2])
        steps.append([(n, (t - t0) * 1e3, (e00.elapsed_time(e) - g0) if e is not None else float('nan'), ((e00.elapsed_time(e) - (t - t00) * 1e3) if e is not None else float('nan')))
                      for n, t, e in seg])
    print('b = %d, TTSC_TEXT_AT = %d, TTSC_STEP_LAZY = %d: %.1f ms per step (with the probe\'s events)' % (a.batch, T.TEXT_AT, int(T.STEP_LAZY), wall))
    print('%-12s %10s %10s %12s' % ('phase end', 'host ms', 'GPU ms', 'GPU behind host'))
    names = [n for n, _, _, _ in steps[0]]
    for i, n in enumerate(names):
        h = np.mean([s[i][1] for s in steps])
        g = np.mean([s[i][2] for s in steps])
        lag = np.mean([s[i][3] for s in steps])
        print('%-12s %10.2f %10.2f %12.2f' % (n, h, g, lag))
    return
    print('%-12s %10s %10s %10s' % ('phase end', 'host ms', 'GPU ms', 'GPU - host'))
    names = [n for n, _, _ in steps[0]]
    for i, n in enumerate(names):
        h = np.mean([s[i][1] for s in steps])
        g = np.mean([s[i][2] for s in steps])
        print('%-12s %10.2f %10.2f %10.2f' % (n, h, g, g - h))


if __name__ == '__main__':
    main()
