"""Per-stage micro-benchmark of the generator's ResBlock1 work at BASELINE config[1] sizes (B=64 x 8 s), through the C ABI.

    python tools/bench_layers.py [--stages 4,3] [--B 64] [--iters 5]

For every stage (channels C, length L) and kernel size K it times one whole ResBlock1 (three pairs) as
  pairs   : the per-pair path (respair32 for C=32, two conv launches per pair otherwise)
  chain0/1: the fused chain kernel (resblock.hip), small / large tile
and prints ms, algorithmic TFLOP/s and the fraction of the split-precision MFMA ceiling (2500/3 TF/s)."""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ttscube_amd import _lib
from ttscube_amd.hip_layers import Conv1dHip

STAGES = {1: (256, 4001), 2: (128, 12004), 3: (64, 48016), 4: (32, 192064)}


def timed(fn, iters):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--stages', default='4,3')
    ap.add_argument('--B', type=int, default=64)
    ap.add_argument('--iters', type=int, default=5)
    ap.add_argument('--ks', default='3,7,11')
    ap.add_argument('--only-chain', action='store_true')
    a = ap.parse_args()
    L_ = _lib.lib()
    for st in [int(v) for v in a.stages.split(',')]:
        Cc, L = STAGES[st]
        x = torch.randn(a.B, Cc, L, device='cuda')
        y = torch.empty_like(x)
        t1 = torch.empty_like(x)
        t2 = torch.empty_like(x)
        for k in [int(v) for v in a.ks.split(',')]:
            c1s, c2s = [], []
            for d in (1, 3, 5):
                c1 = Conv1dHip(Cc, Cc, k, padding=d * (k - 1) // 2, dilation=d).set_precision('f16x3')
                c2 = Conv1dHip(Cc, Cc, k, padding=(k - 1) // 2).set_precision('f16x3')
                c1.set_weight(torch.randn(Cc, Cc, k) / (Cc * k) ** 0.5, torch.randn(Cc) * 0.1)
                c2.set_weight(torch.randn(Cc, Cc, k) / (Cc * k) ** 0.5, torch.randn(Cc) * 0.1)
                c1s.append(c1)
                c2s.append(c2)
            flops = 2.0 * a.B * L * Cc * Cc * k * 6
            a1 = (C.c_void_p * 3)(*[c._h for c in c1s])
            a2 = (C.c_void_p * 3)(*[c._h for c in c2s])

            def pairs():
                src = x
                for m in range(3):
                    dst = y if m == 2 else (t1 if src is not t1 else t2)
                    if Cc == 32:
                        _lib.check(L_.ttsc_respair_forward(c1s[m]._h, c2s[m]._h, _lib.dev_ptr(src), a.B, L, _lib.dev_ptr(dst), 0, None,
                                                           _lib.current_stream()), 'respair')
                    else:
                        xt = t2 if dst is not t2 else t1
                        if xt is src:
                            xt = y
                        c1s[m](src, out=xt, in_slope=0.1)
                        c2s[m](xt, out=dst, resid=src, in_slope=0.1)
                    src = dst

            def chain(shape):
                _lib.check(L_.ttsc_rbchain_forward(a1, a2, 3, _lib.dev_ptr(x), a.B, L, _lib.dev_ptr(y), 0, None, shape,
                                                   _lib.current_stream()), 'rbchain')

            if Cc >= 128:
                os.environ['TTSC_CONV_WIDE'] = '0'
                res = [('old', timed(pairs, a.iters))]
                os.environ['TTSC_CONV_WIDE'] = '1'
                res.append(('wide', timed(pairs, a.iters)))
            else:
                res = [('pairs', timed(pairs, a.iters))]
            if a.only_chain:
                res = []
            if L_.ttsc_rbchain_supported(a1, a2, 3):
                res.append(('chain0', timed(lambda: chain(0), a.iters)))
                res.append(('chain1', timed(lambda: chain(1), a.iters)))
                for sh in [int(v) for v in os.environ.get('BENCH_CHAIN_SHAPES', '').split(',') if v]:
                    res.append(('chain%d' % sh, timed(lambda: chain(sh), a.iters)))
            print('stage %d C=%3d L=%6d K=%2d  ' % (st, Cc, L, k) + '  '.join(
                '%s %.3f ms %.0f TF/s (%.2f)' % (n, ms, flops / ms / 1e9, flops / ms / 1e9 / (2500.0 / 3)) for n, ms in res), flush=True)


if __name__ == '__main__':
    main()
