"""Per-launch time and memory rate of the polyphase de-interleave (ttsc_deinterleave_x) on the shapes of the Cubegan step's discriminators at b = 16
(32 sequences: real + generated as one batch).     python tools/probes/deint_time.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ttscube_amd import _lib

SHAPES = [  # N, C, L (rows), G, s, P, pad, K
    (32, 1, 4000, 1, 3, 3, 2, 5), (32, 32, 1334, 1, 3, 3, 2, 5), (32, 128, 446, 1, 3, 3, 2, 5), (32, 512, 150, 1, 3, 3, 2, 5),
    (32, 32, 1091, 1, 3, 11, 2, 5), (32, 128, 364, 1, 3, 11, 2, 5),
    (32, 128, 12000, 4, 2, 1, 20, 41), (32, 128, 6000, 16, 2, 1, 20, 41), (32, 256, 3000, 16, 4, 1, 20, 41), (32, 512, 750, 16, 4, 1, 20, 41),
]
L_ = _lib.lib()
for N, C, L, G, s, P, pad, K in SHAPES:
    J = -(-K // s)
    Lout = (L + 2 * pad - K) // s + 1
    M = Lout + J - 1
    x = torch.randn(N, C, L * P, device='cuda')
    out = torch.empty(N, s * C, M * P, device='cuda')
    for bwd in (0, 1):
        src, dst = (x, out) if not bwd else (out, x)
        call = lambda: _lib.check(L_.ttsc_deinterleave_x(_lib.dev_ptr(src), _lib.dev_ptr(dst), N, C, L, G, s, P, pad, M, bwd, _lib.current_stream()), 'deint')
        for _ in range(3):
            call()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            call()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 50 * 1e3
        by = (x.numel() + out.numel()) * 4
        print('N=%d C=%d L=%d G=%d s=%d P=%d %s: %.1f us  %.2f TB/s' % (N, C, L, G, s, P, 'bwd' if bwd else 'fwd', us, by / us / 1e6))
