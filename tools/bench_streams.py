"""Do the three ResBlocks of a generator stage overlap usefully when they run on three streams?  (They are independent given the
stage input; within a launch all workgroups reach their memory-bound epilogue at about the same time, and every launch has a
tail.)   python tools/bench_streams.py [--stages 1,2,3,4] [--B 64]"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ttscube_amd import _lib
from ttscube_amd.hip_layers import Conv1dHip

STAGES = {1: (256, 4001), 2: (128, 12004), 3: (64, 48016), 4: (32, 192064)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--stages', default='1,2,3,4')
    ap.add_argument('--B', type=int, default=64)
    ap.add_argument('--iters', type=int, default=5)
    a = ap.parse_args()
    L_ = _lib.lib()
    for st in [int(v) for v in a.stages.split(',')]:
        Cc, L = STAGES[st]
        x = torch.randn(a.B, Cc, L, device='cuda')
        blocks = []
        for k in (3, 7, 11):
            c1s, c2s = [], []
            for d in (1, 3, 5):
                c1 = Conv1dHip(Cc, Cc, k, padding=d * (k - 1) // 2, dilation=d).set_precision('f16x3')
                c2 = Conv1dHip(Cc, Cc, k, padding=(k - 1) // 2).set_precision('f16x3')
                c1.set_weight(torch.randn(Cc, Cc, k) / (Cc * k) ** 0.5, torch.randn(Cc) * 0.1)
                c2.set_weight(torch.randn(Cc, Cc, k) / (Cc * k) ** 0.5, torch.randn(Cc) * 0.1)
                c1s.append(c1)
                c2s.append(c2)
            bufs = [torch.empty_like(x) for _ in range(3)]
            blocks.append((c1s, c2s, bufs, (C.c_void_p * 3)(*[c._h for c in c1s]), (C.c_void_p * 3)(*[c._h for c in c2s])))

        def run_block(j):
            c1s, c2s, (y, t1, t2), a1, a2 = blocks[j]
            if Cc <= 64:
                _lib.check(L_.ttsc_rbchain_forward(a1, a2, 3, _lib.dev_ptr(x), a.B, L, _lib.dev_ptr(y), 0, None, -1, _lib.current_stream()), 'rbchain')
                return
            src = x
            for m in range(3):
                dst = y if m == 2 else t1
                c1s[m](src, out=t2, in_slope=0.1)
                c2s[m](t2, out=dst, resid=src, in_slope=0.1)
                src = dst

        streams = [torch.cuda.Stream() for _ in range(3)]

        def serial():
            for j in range(3):
                run_block(j)

        def parallel():
            main_s = torch.cuda.current_stream()
            ev = torch.cuda.Event()
            ev.record(main_s)
            for j in range(3):
                streams[j].wait_event(ev)
                with torch.cuda.stream(streams[j]):
                    run_block(j)
            for j in range(3):
                main_s.wait_stream(streams[j])

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

        ts, tp = timed(serial), timed(parallel)
        print('stage %d C=%3d L=%6d: one stream %.3f ms, three streams %.3f ms (%.1f %%)' % (st, Cc, L, ts, tp, 100.0 * (tp / ts - 1)), flush=True)


if __name__ == '__main__':
    main()
