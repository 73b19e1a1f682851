"""Where do the milliseconds between `bench.py`'s training leg (no process group, no reducers) and `bench.py --mode train` go?
Times the b = 16 Cubegan step (a) plain, (b) with the ArenaReducers built (gradient hooks registered) but not used, (c) with the exchange."""
import os
import random
import socket
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.test_distributed_gpu import _cubegan_setup
from ttscube_amd.networks import training as T


def timed(model, batch, opts, reds, n=6):
    rng = random.Random(3)
    for _ in range(3):
        T.cubegan_training_step(model, batch, opts, reducers=reds, rng=rng)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        T.cubegan_training_step(model, batch, opts, reducers=reds, rng=rng)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    model, batch, _ = _cubegan_setup(777, nitems=16)
    opts = T.cubegan_configure_optimizers(model)
    print('plain step (no process group, no hooks):          %.1f ms' % timed(model, batch, opts, None), flush=True)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
    print('process group initialised, still no hooks:        %.1f ms' % timed(model, batch, opts, None), flush=True)
    overlap = os.environ.get('PROBE_OVERLAP', '1') != '0'
    reds = T.cubegan_reducers(model, opts, force=True, overlap=overlap)
    print('reducers built (overlap=%s), not used:             %.1f ms' % (overlap, timed(model, batch, opts, None)), flush=True)
    print('with the exchange:                                %.1f ms   early chunks %s' % (timed(model, batch, opts, reds), [r.launched_early for r in reds]), flush=True)
    print('exchange no longer passed (same process):         %.1f ms' % timed(model, batch, opts, None), flush=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        for r in reds:
            r.reduce()
    torch.cuda.synchronize()
    print('the three exchanges alone:                        %.2f ms' % ((time.perf_counter() - t0) / 5 * 1e3), flush=True)
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
