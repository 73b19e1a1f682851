"""Where does a kernel's time go?  Times one ResBlock1 of a generator stage with parts of the kernel switched off
(measurement build `make -C ttscube_amd/csrc ablate` -> libttscube_hip_ablate.so, -DTTSC_ABLATE; results are wrong by design).

    python tools/ablate.py --stage 1 --k 11            # wide conv kernel (stages 1, 2): TTSC_CONV_DBG bits
    python tools/ablate.py --stage 4 --k 11 --shape 1  # fused chain kernel (stages 3, 4): TTSC_CHAIN_DBG bits
"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ttscube_amd import _lib

_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), 'libttscube_hip_ablate.so')
from ttscube_amd.hip_layers import Conv1dHip  # noqa: E402

STAGES = {1: (256, 4001), 2: (128, 12004), 3: (64, 48016), 4: (32, 192064)}
WIDE_BITS = [(0, 'full'), (4, 'no epilogue'), (1, 'no staging in loop'), (2, 'no chunk barrier'), (8, 'no weight prefetch'),
             (1 | 2, 'no staging, no barrier'), (1 | 2 | 4 | 8, 'MFMA + LDS reads only'),
             (32 | (6 << 8), 'skew 2nd workgroup ~24 us'), (32 | 64 | (6 << 8) | (1 << 16), 'skew 24 us + CU spread 0..28 us'), (32 | 64 | (6 << 8) | (2 << 16), 'skew 24 us + CU spread 0..56 us'), (64 | (2 << 16), 'CU spread 0..56 us only')]
CHAIN_BITS = [(0, 'full'), (1, 'no image conversions'), (2, 'no barriers'), (4, 'no final store'), (8, 'no x load'), (16, 'no weight loads in loop'),
              (4 | 8, 'no HBM traffic'), (1 | 2 | 4 | 8 | 16, 'MFMA + LDS reads only')]


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


GEN_BITS = [(0, 'full'), (4, 'no epilogue'), (1, 'no staging commit'), (8, 'no global loads in loop'), (1 | 8, 'no staging at all'), (2, 'no MFMA loop'),
            (1 | 4 | 8, 'MFMA + LDS reads only')]


def ups_ablation(a):
    """conv_pre / upsampler i at config[1] sizes (B x T = 64 x 800) on conv_f16x3_kernel with parts switched off (TTSC_CONV_DBG)"""
    cfg = {-2: (80, 512, 7, 1, 800), 0: (512, 256, 16, 5, 800), 1: (256, 128, 16, 3, 4001), 2: (128, 64, 4, 4, 12004), 3: (64, 32, 4, 4, 48016)}[a.ups]
    Cin, Cout, k, u, L = cfg
    if a.ups == -2:
        c = Conv1dHip(Cin, Cout, k, padding=3).set_precision('f16x3')
        w = torch.randn(Cout, Cin, k) / (Cin * k) ** 0.5
    else:
        c = Conv1dHip(Cin, Cout, k, stride=u, padding=(k - u) // 2, transposed=True).set_precision('f16x3')
        w = torch.randn(Cin, Cout, k) / (Cin * k / u) ** 0.5
    c.set_weight(w, torch.randn(Cout) * 0.1)
    x = torch.randn(a.B, Cin, L, device='cuda')
    y = torch.empty((a.B, Cout, c.out_len(L)), device='cuda')
    flops = 2.0 * a.B * L * Cin * Cout * k
    base = None
    for bit, name in GEN_BITS:
        os.environ['TTSC_CONV_DBG'] = str(bit)
        ms = timed(lambda: c(x, out=y, in_slope=0.1), a.iters)
        base = base or ms
        print('%s full=%s %-28s %7.3f ms  %5.1f%% of full  %6.1f TF/s  (in %.0f MB, out %.0f MB)' % (
            'conv_pre' if a.ups == -2 else 'ups.%d' % a.ups, 'y' if bit == 0 else 'n', name, ms, 100 * ms / base, flops / ms / 1e9, x.numel() * 4 / 1e6, y.numel() * 4 / 1e6), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--stage', type=int, default=1)
    ap.add_argument('--k', type=int, default=11)
    ap.add_argument('--B', type=int, default=64)
    ap.add_argument('--shape', type=int, default=-1)
    ap.add_argument('--iters', type=int, default=5)
    ap.add_argument('--ups', type=int, default=-1, help='ablate upsampler i of config_v1 (-2: conv_pre) on the general split kernel instead of a ResBlock')
    a = ap.parse_args()
    L_ = _lib.lib()
    if a.ups != -1:
        return ups_ablation(a)
    Cc, L = STAGES[a.stage]
    k = a.k
    x = torch.randn(a.B, Cc, L, device='cuda')
    y = torch.empty_like(x)
    t1 = torch.empty_like(x)
    c1s, c2s = [], []
    for d in (1, 3, 5):
        c1 = Conv1dHip(Cc, Cc, k, padding=d * (k - 1) // 2, dilation=d).set_precision('f16x3')
        c2 = Conv1dHip(Cc, Cc, k, padding=(k - 1) // 2).set_precision('f16x3')
        c1.set_weight(torch.randn(Cc, Cc, k) / (Cc * k) ** 0.5, torch.randn(Cc) * 0.1)
        c2.set_weight(torch.randn(Cc, Cc, k) / (Cc * k) ** 0.5, torch.randn(Cc) * 0.1)
        c1s.append(c1)
        c2s.append(c2)
    flops = 2.0 * a.B * L * Cc * Cc * k * 6
    if Cc >= 128:
        def run():
            src = x
            for m in range(3):
                c1s[m](src, out=t1, in_slope=0.1)
                c2s[m](t1, out=y, resid=src, in_slope=0.1)
        bits, env = WIDE_BITS, 'TTSC_CONV_DBG'
    else:
        a1 = (C.c_void_p * 3)(*[c._h for c in c1s])
        a2 = (C.c_void_p * 3)(*[c._h for c in c2s])

        def run():
            _lib.check(L_.ttsc_rbchain_forward(a1, a2, 3, _lib.dev_ptr(x), a.B, L, _lib.dev_ptr(y), 0, None, a.shape, _lib.current_stream()), 'rbchain')
        bits, env = CHAIN_BITS, 'TTSC_CHAIN_DBG'
    base = None
    for bit, name in bits:
        os.environ[env] = str(bit)
        ms = timed(run, a.iters)
        base = base or ms
        print('stage %d C=%d K=%d %-28s %7.3f ms  %5.1f%% of full   %.0f TF/s' % (a.stage, Cc, k, name, ms, 100 * ms / base, flops / ms / 1e9), flush=True)
    os.environ[env] = '0'


if __name__ == '__main__':
    main()
