"""Is the Cubegan step bit-reproducible, and does the hooked RCCL exchange (world 1) leave it unchanged?  Repeats 5-step runs from identical
weights / data without and with the exchange and prints the largest relative parameter difference against the first run."""
import os
import socket
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.test_distributed_gpu import _cubegan_setup, _run_steps


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
    ref = None
    for ex in (False, True):
        for r in range(reps):
            model, batch, random = _cubegan_setup(777)
            got, early = _run_steps(model, batch, random, 5, with_exchange=ex)
            if ref is None:
                ref = got
            worst = max(float((a - b).abs().max() / (b.abs().max() + 1e-12)) for a, b in zip(got, ref))
            nbad = sum(1 for a, b in zip(got, ref) if not torch.equal(a, b))
            print('exchange=%s run %d: tensors differing from run 0: %d, worst relative difference %.3e, early chunks %s' % (ex, r, nbad, worst, early[-1] if early else None), flush=True)
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
