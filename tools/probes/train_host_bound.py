"""Is the b = 16 Cubegan step bound by the host (Python + launch calls) or by the GPU?  Per step: the wall time until the host has
enqueued everything (it reaches the split-status poll at the end of the step), and how long it then waits for the GPU to drain."""
import os
import random
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.test_distributed_gpu import _cubegan_setup
from ttscube_amd import _lib
from ttscube_amd.networks import training as T


def main():
    nitems = int(os.environ.get('PROBE_B', '16'))
    model, batch, _ = _cubegan_setup(777, nitems=nitems)
    opts = T.cubegan_configure_optimizers(model)
    rng = random.Random(3)
    marks = {}
    orig = _lib.check_split_status

    def poll(where, stream=None):
        if stream is not None:      # (the per-side polls before the exchanges: pass through)
            return orig(where, stream=stream)
        marks['enq'] = time.perf_counter()
        torch.cuda.synchronize()
        marks['drained'] = time.perf_counter()
        return orig(where)

    T._lib.check_split_status = poll
    for _ in range(3):
        T.cubegan_training_step(model, batch, opts, rng=rng)
    rows = []
    for _ in range(8):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        T.cubegan_training_step(model, batch, opts, rng=rng)
        t1 = time.perf_counter()
        rows.append(((marks['enq'] - t0) * 1e3, (marks['drained'] - marks['enq']) * 1e3, (t1 - t0) * 1e3))
    for r in rows:
        print('host enqueue %.1f ms   then waits %.1f ms for the GPU   step %.1f ms' % r, flush=True)
    h, w, s = (sum(r[i] for r in rows) / len(rows) for i in range(3))
    print('b = %d: mean host enqueue %.1f ms, GPU drain wait %.1f ms, step %.1f ms -> %s-bound' % (nitems, h, w, s, 'host' if w < 0.15 * s else 'GPU'))


if __name__ == '__main__':
    main()
