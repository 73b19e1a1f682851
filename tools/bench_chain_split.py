"""One ResBlock1 chain (three pairs) as ONE fused launch vs split into two or three launches with smaller halos (fewer recomputed columns, one more
tile fill / store each), at BASELINE config[1] sizes, through the C ABI.  Results of every split are compared bit for bit with the single launch.

    python tools/bench_chain_split.py [--stages 3,4] [--ks 7,11] [--B 64] [--iters 5]"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ttscube_amd import _lib
from ttscube_amd.hip_layers import Conv1dHip
from tools.bench_layers import STAGES, timed


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--stages', default='3,4')
    ap.add_argument('--B', type=int, default=64)
    ap.add_argument('--iters', type=int, default=5)
    ap.add_argument('--ks', default='7,11')
    a = ap.parse_args()
    L_ = _lib.lib()
    for st in [int(v) for v in a.stages.split(',')]:
        Cc, L = STAGES[st]
        torch.manual_seed(st)
        x = torch.randn(a.B, Cc, L, device='cuda')
        y0 = torch.randn(a.B, Cc, L, device='cuda')
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

            def run(split, shapes):
                """split: pairs per launch, e.g. (3,), (2, 1), (1, 2), (1, 1, 1); the last launch accumulates into y like the product's K = 7 / 11 blocks"""
                src, p0 = x, 0
                for i, (n, sh) in enumerate(zip(split, shapes)):
                    last = i == len(split) - 1
                    dst = y if last else (t1 if src is not t1 else t2)
                    a1 = (C.c_void_p * n)(*[c._h for c in c1s[p0:p0 + n]])
                    a2 = (C.c_void_p * n)(*[c._h for c in c2s[p0:p0 + n]])
                    _lib.check(L_.ttsc_rbchain_forward(a1, a2, n, _lib.dev_ptr(src), a.B, L, _lib.dev_ptr(dst), 1 if last else 0, None, sh,
                                                       _lib.current_stream()), 'rbchain')
                    src, p0 = dst, p0 + n

            def result(split, shapes):
                y.copy_(y0)
                run(split, shapes)
                torch.cuda.synchronize()
                return y.clone()

            ref = result((3,), (-1,))
            out = []
            cases = [((3,), (-1,)), ((2, 1), (-1, -1)), ((1, 2), (-1, -1)), ((1, 1, 1), (-1, -1, -1))]
            big = 12 if Cc == 32 else 11
            cases += [((2, 1), (big, big)), ((2, 1), (big, 10)), ((1, 2), (big, big)), ((1, 1, 1), (big, big, big)), ((1, 1, 1), (10, 10, 10))]
            for split, shapes in cases:
                same = bool(torch.equal(result(split, shapes), ref))
                ms = timed(lambda: run(split, shapes), a.iters)
                out.append('%s%s %.3f ms%s' % ('+'.join(map(str, split)), '' if shapes[0] < 0 else '@' + '/'.join(map(str, shapes)), ms, '' if same else ' DIFFERS'))
            print('stage %d C=%3d L=%6d K=%2d  ' % (st, Cc, L, k) + '  '.join(out) + '   [3 = %.0f TF/s]' % (flops / timed(lambda: run((3,), (-1,)), a.iters) / 1e9), flush=True)


if __name__ == '__main__':
    main()
