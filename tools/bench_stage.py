"""Micro-benchmark of the 32-channel generator stage at BASELINE config[1] size (B=64 x 192 064 samples), through the C ABI:

  chains   : three ttsc_rbchain_forward launches (K = 3, 7, 11; default tile shapes) + conv_post        (round-3 path)
  chain sN : every chain with tile shape N (0: 4 waves x 512 columns, 1: 8 x 1024, 2: 8 x 1024 with 6-step weight groups,
             3 / 4: 8 waves x 768 columns with half- / whole-convolution weight groups), per K

    python tools/bench_stage.py [--B 64] [--L 192064] [--iters 5] [--shapes 0,1,2,3,4]
"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ttscube_amd import _lib
from ttscube_amd.hip_layers import Conv1dHip


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
    ap.add_argument('--B', type=int, default=64)
    ap.add_argument('--L', type=int, default=192064)
    ap.add_argument('--C', type=int, default=32)
    ap.add_argument('--iters', type=int, default=5)
    ap.add_argument('--shapes', default='0,1,2,3,4')
    ap.add_argument('--acc', type=int, default=0, help='1: time the chain launches as y += chain(x) (how the second and third block of a stage run)')
    ap.add_argument('--data', choices=('random', 'zeros'), default='random', help='zeros: all-zero input, weights and biases (how much of the time is the power budget)')
    a = ap.parse_args()
    L_ = _lib.lib()
    Cc, L, B = a.C, a.L, a.B
    torch.manual_seed(0)
    x = torch.randn(B, Cc, L, device='cuda') if a.data == 'random' else torch.zeros(B, Cc, L, device='cuda')
    wscale = 1.0 if a.data == 'random' else 0.0
    y = torch.empty_like(x)
    wav = torch.empty(B, 1, L, device='cuda')
    wav2 = torch.empty(B, 1, L, device='cuda')
    blocks = []
    for k in (3, 7, 11):
        c1s, c2s = [], []
        for d in (1, 3, 5):
            c1 = Conv1dHip(Cc, Cc, k, padding=d * (k - 1) // 2, dilation=d).set_precision('f16x3')
            c2 = Conv1dHip(Cc, Cc, k, padding=(k - 1) // 2).set_precision('f16x3')
            c1.set_weight(torch.randn(Cc, Cc, k) / (Cc * k) ** 0.5 * 0.5 * wscale, torch.randn(Cc) * 0.1 * wscale)
            c2.set_weight(torch.randn(Cc, Cc, k) / (Cc * k) ** 0.5 * 0.5 * wscale, torch.randn(Cc) * 0.1 * wscale)
            c1s.append(c1)
            c2s.append(c2)
        blocks.append((k, c1s, c2s))
    post = Conv1dHip(Cc, 1, 7, padding=3)
    post.set_weight(torch.randn(1, Cc, 7) / (Cc * 7) ** 0.5, torch.randn(1) * 0.1)
    flops_k = {k: 2.0 * B * L * Cc * Cc * k * 6 for k in (3, 7, 11)}
    flops = sum(flops_k.values()) + 2.0 * B * L * Cc * 7
    ceiling = 2500.0 / 3

    def arr(cs):
        return (C.c_void_p * len(cs))(*[c._h for c in cs])

    def chain(j, shape, acc):
        k, c1s, c2s = blocks[j]
        _lib.check(L_.ttsc_rbchain_forward(arr(c1s), arr(c2s), 3, _lib.dev_ptr(x), B, L, _lib.dev_ptr(y), acc, None, shape,
                                           _lib.current_stream()), 'rbchain')

    def chains_default(out):
        for j in range(3):
            chain(j, -1, 1 if j else 0)
        post(y, out=out, in_scale=1.0 / 3, in_slope=0.01, act='tanh')

    ms = timed(lambda: chains_default(wav), a.iters)
    print('chains + conv_post (round-3 path)   %.3f ms  %.0f TF/s (%.3f)' % (ms, flops / ms / 1e9, flops / ms / 1e9 / ceiling), flush=True)
    for sh in [int(v) for v in a.shapes.split(',')]:
        row = []
        for j, (k, _, _) in enumerate(blocks):
            try:
                ms = timed(lambda: chain(j, sh, a.acc), a.iters)
                row.append('K=%2d %.3f ms (%.3f)' % (k, ms, flops_k[k] / ms / 1e9 / ceiling))
            except _lib.TTSCError as e:
                row.append('K=%2d n/a' % k)
        print('chain shape %d: ' % sh + '   '.join(row), flush=True)



if __name__ == '__main__':
    main()
