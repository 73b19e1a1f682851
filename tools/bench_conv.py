"""Micro-benchmark of one conv layer through the C ABI (ablation switches via env TTSC_CONV_DBG)."""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ttscube_amd.hip_layers import Conv1dHip

ap = argparse.ArgumentParser()
ap.add_argument('--c', type=int, default=256); ap.add_argument('--k', type=int, default=11); ap.add_argument('--d', type=int, default=1)
ap.add_argument('--L', type=int, default=4001); ap.add_argument('--B', type=int, default=64); ap.add_argument('--prec', default='f16x3')
ap.add_argument('--resid', type=int, default=0)
a = ap.parse_args()
conv = Conv1dHip(a.c, a.c, a.k, padding=a.d * (a.k - 1) // 2, dilation=a.d).set_precision(a.prec)
conv.set_weight(torch.randn(a.c, a.c, a.k) / (a.c * a.k) ** 0.5, torch.randn(a.c) * 0.1)
x = torch.randn(a.B, a.c, a.L, device='cuda'); y = torch.empty_like(x); r = torch.randn_like(x) if a.resid else None
for _ in range(3): conv(x, out=y, resid=r, in_slope=0.1)
torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): conv(x, out=y, resid=r, in_slope=0.1)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
fl = 2.0 * a.B * a.L * a.c * a.c * a.k
print('DBG=%s C=%d k=%d d=%d L=%d B=%d %s: %.3f ms  %.1f TF/s algorithmic (x3 = %.0f MFMA TF/s)' % (os.environ.get('TTSC_CONV_DBG', '0'), a.c, a.k, a.d, a.L, a.B, a.prec, ms, fl / ms / 1e9, 3 * fl / ms / 1e9))
