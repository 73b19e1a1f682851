"""Reproducibility of the Cubegan step when fresh allocations hold GARBAGE instead of the zeros a young process gets from the driver: the caching
allocator is filled with NaN / large-value blocks of many sizes and emptied back into its cache before every run, so a kernel that reads memory it
(or its producer) never wrote shows up as a run-to-run difference.  Prints which parameters differ between two identical 5-step runs."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.test_distributed_gpu import _cubegan_setup, _run_steps


def poison(seed):
    g = torch.Generator(device='cuda').manual_seed(seed)
    keep = []
    for n in [1 << k for k in range(6, 27)] * 3:
        t = torch.empty(n, device='cuda')
        t.uniform_(-1e3, 1e3, generator=g)
        if n % 3 == 0 or seed % 2:
            t[::7] = float('nan')
        keep.append(t)
    torch.cuda.synchronize()
    del keep


def main():
    nsteps = int(os.environ.get('PROBE_STEPS', '5'))
    held = []
    if os.environ.get('PROBE_PRIO', '0') != '0':   # a high-priority stream gets a hardware queue of its own: the other streams then map differently
        for _ in range(int(os.environ.get('PROBE_PRIO'))):
            st = torch.cuda.Stream(priority=-1)
            with torch.cuda.stream(st):
                held.append(torch.zeros(1024, device='cuda') + 1)
            held.append(st)
        torch.cuda.synchronize()
    ref = None
    for r in range(3):
        model, batch, random = _cubegan_setup(777)
        if os.environ.get('PROBE_POISON', '1') != '0':
            poison(r + 1)
        got, _ = _run_steps(model, batch, random, nsteps, with_exchange=False)
        names = [n for n, _ in model.named_parameters()]
        if ref is None:
            ref = got
            continue
        bad = [(n, float((a - b).abs().max() / (b.abs().max() + 1e-12))) for n, a, b in zip(names, got, ref) if not torch.equal(a, b)]
        print('run %d vs run 0: %d of %d parameter tensors differ' % (r, len(bad), len(names)), flush=True)
        groups = {}
        for n, d in bad:
            k = '.'.join(n.split('.')[:3])
            groups.setdefault(k, [0, 0.0])
            groups[k][0] += 1
            groups[k][1] = max(groups[k][1], d)
        for k, (c, d) in sorted(groups.items())[:40]:
            print('   %-60s %3d tensors, worst %.2e' % (k, c, d))


if __name__ == '__main__':
    main()
