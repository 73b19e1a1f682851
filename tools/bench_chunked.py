"""Round-5 experiment: do the chain launches of a narrow generator stage run faster when the batch is walked in chunks whose working set
(x chunk + y chunk) stays in the 256 MiB Infinity Cache — K = 3, 7, 11 back to back per chunk, so that the second and third block's x reads and
the y read-modify-writes never reach HBM — with the chunks dealt round-robin over S streams so that one chunk's tail rounds overlap another's?

    python tools/bench_chunked.py [--C 32 --L 192064] [--B 64] [--iters 5]
"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ttscube_amd import _lib
from ttscube_amd.hip_layers import Conv1dHip


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--B', type=int, default=64)
    ap.add_argument('--L', type=int, default=192064)
    ap.add_argument('--C', type=int, default=32)
    ap.add_argument('--iters', type=int, default=5)
    ap.add_argument('--chunks', default='64,16,8,4,2')
    ap.add_argument('--streams', default='1,2,3')
    a = ap.parse_args()
    L_ = _lib.lib()
    Cc, L, B = a.C, a.L, a.B
    torch.manual_seed(0)
    x = torch.randn(B, Cc, L, device='cuda')
    y = torch.empty_like(x)
    blocks = []
    for k in (3, 7, 11):
        c1s, c2s = [], []
        for d in (1, 3, 5):
            c1 = Conv1dHip(Cc, Cc, k, padding=d * (k - 1) // 2, dilation=d).set_precision('f16x3')
            c2 = Conv1dHip(Cc, Cc, k, padding=(k - 1) // 2).set_precision('f16x3')
            c1.set_weight(torch.randn(Cc, Cc, k) / (Cc * k) ** 0.5 * 0.5, torch.randn(Cc) * 0.1)
            c2.set_weight(torch.randn(Cc, Cc, k) / (Cc * k) ** 0.5 * 0.5, torch.randn(Cc) * 0.1)
            c1s.append(c1)
            c2s.append(c2)
        blocks.append((k, c1s, c2s))
    flops = sum(2.0 * B * L * Cc * Cc * k * 6 for k in (3, 7, 11))
    ceiling = 2500.0 / 3

    def arr(cs):
        return (C.c_void_p * len(cs))(*[c._h for c in cs])

    arrs = [(arr(c1s), arr(c2s)) for _, c1s, c2s in blocks]

    def chain(j, b0, nb, stream):
        xs, ys = x[b0:b0 + nb], y[b0:b0 + nb]
        _lib.check(L_.ttsc_rbchain_forward(arrs[j][0], arrs[j][1], 3, _lib.dev_ptr(xs), nb, L, _lib.dev_ptr(ys), 1 if j else 0, None, -1,
                                           C.c_void_p(stream)), 'rbchain')

    main_s = torch.cuda.current_stream()
    pool = [torch.cuda.Stream() for _ in range(4)]

    def run(chunk, ns):
        if ns == 1 and chunk >= B:
            for j in range(3):
                chain(j, 0, B, main_s.cuda_stream)
            return
        sts = pool[:ns]
        for s in sts:
            s.wait_stream(main_s)
        for i, b0 in enumerate(range(0, B, chunk)):
            s = sts[i % ns]
            for j in range(3):
                chain(j, b0, min(chunk, B - b0), s.cuda_stream)
        for s in sts:
            main_s.wait_stream(s)

    def timed(fn):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / a.iters

    run(B, 1)
    torch.cuda.synchronize()
    ref = y.clone()
    for chunk in [int(v) for v in a.chunks.split(',')]:
        for ns in [int(v) for v in a.streams.split(',')]:
            if chunk >= B and ns > 1:
                continue
            ms = timed(lambda: run(chunk, ns))
            y.zero_()
            run(chunk, ns)
            torch.cuda.synchronize()
            print('C=%d L=%d B=%d  chunk %2d utterances (%5.0f MB of x + y)  %d stream(s): %.3f ms  %.0f TF/s (%.3f)  same bits: %s' % (
                Cc, L, B, chunk, 2 * chunk * Cc * L * 4 / 1e6, ns, ms, flops / ms / 1e9, flops / ms / 1e9 / ceiling, bool(torch.equal(ref, y))), flush=True)


if __name__ == '__main__':
    main()
