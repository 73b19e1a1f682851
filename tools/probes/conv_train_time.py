"""Per-launch time of the split-precision TRAINING convolution (ttsc_conv_train: range reduction + weight packing + convolution) on shapes of the
Cubegan step at b = 16, optionally on the measurement build with ablation bits (TTSC_CONV_DBG: 1 = no staging conversion / LDS commit, 2 = no MFMA
loop, 4 = no epilogue, 8 = no global loads of the next chunk).     python tools/probes/conv_train_time.py [--lib path/to/lib.so]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ttscube_amd import _lib
if '--lib' in sys.argv:
    _lib.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index('--lib') + 1])
from ttscube_amd.hifigan.autograd import _conv_split

SHAPES = [  # B, Cin, Cout, K, L, padding, dilation   (MPD period 3 unless noted; strided layers after the de-interleave: 2 taps over 3 x Cin channels)
    (32, 3 * 32, 128, 2, 1334 * 3, 0, 3),
    (32, 3 * 128, 512, 2, 446 * 3, 0, 3),
    (32, 3 * 512, 1024, 2, 150 * 3, 0, 3),
    (32, 1024, 1024, 5, 150 * 3, 6, 3),
    (16, 256, 256, 3, 1000, 1, 1),       # generator stage 1 (50-frame crop: 50 x 5 x ... = 1 000 positions at 256 channels)
    (16, 256, 256, 7, 1000, 3, 1),
    (16, 128, 128, 11, 3000, 5, 1),
    (16, 64, 64, 7, 12000, 3, 1),
]
for B, Cin, Cout, K, L, pad, d in SHAPES:
    x = torch.randn(B, Cin, L, device='cuda')
    w = torch.randn(Cout, Cin, K, device='cuda') / (Cin * K) ** 0.5
    b = torch.randn(Cout, device='cuda')
    for _ in range(3):
        _conv_split(x, w, b, None, None, Cin, Cout, K, pad, d, 0, in_slope=0.1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        _conv_split(x, w, b, None, None, Cin, Cout, K, pad, d, 0, in_slope=0.1)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    Lout = L + 2 * pad - d * (K - 1)
    fl = 2.0 * B * Cin * Cout * K * Lout
    print('B=%d Cin=%d Cout=%d K=%d L=%d d=%d: %.1f us per call  %.0f algorithmic TFLOP/s' % (B, Cin, Cout, K, L, d, ms * 1e3, fl / ms / 1e9))
