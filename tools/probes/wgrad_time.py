"""Per-launch time of the split-precision weight gradient on the training step's shapes (b = 16: real + generated = 32 sequences).
    python tools/probes/wgrad_time.py [--lib path/to/libttscube_hip_variant.so]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ttscube_amd import _lib
if '--lib' in sys.argv:
    _lib.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index('--lib') + 1])
from ttscube_amd.hifigan import autograd as AG

SHAPES = [  # N, A, Bc, LP, J, step, groups
    (32, 1024, 1024, 100, 5, 1, 1),     # MPD deep layer, period-folded (dilation = period is `step` in the real step; 1 here)
    (32, 1024, 512 * 3, 150, 2, 7, 1),  # strided MPD layer after the de-interleave: 2 taps, period 7
    (32, 512, 128 * 3, 450, 2, 3, 1),
    (16, 256, 256, 4032, 7, 1, 1),      # generator stage 1, K = 7
    (16, 128, 128, 12032, 11, 1, 1),    # generator stage 2, K = 11
    (16, 1024, 128 * 16, 188, 11, 1, 16),   # MSD 512 -> 1024, k41 s4 g16 (64 x 128 per group)
    (16, 1024, 64 * 16, 188, 41, 1, 16),    # MSD 1024 -> 1024, k41 s1 g16
]
for N, A, Bc, LP, J, step, G in SHAPES:
    P = torch.randn(N, A, LP, device='cuda')
    Q = torch.randn(N, Bc, LP + (J - 1) * step, device='cuda')
    for _ in range(3):
        AG._wgrad(P, Q, A, Bc, J, 0, step, 1.0, 0.1, groups=G)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        AG._wgrad(P, Q, A, Bc, J, 0, step, 1.0, 0.1, groups=G)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    fl = 2.0 * A * (Bc // G) * J * N * LP
    print('N=%d A=%d Bc=%d LP=%d J=%d step=%d groups=%d: %.1f us per call (range words + launches + reduction)  %.0f algorithmic TFLOP/s' % (N, A, Bc, LP, J, step, G, ms * 1e3, fl / ms / 1e9))
