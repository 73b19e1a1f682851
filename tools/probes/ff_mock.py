"""Round-6 probe: what could a 2-parallel fast-FIR buy the wide kernel?  MOCK: one dilation-1 layer of K taps over L positions against
THREE launches of the same kernel with (K + 1) / 2 taps over L / 2 positions (the three half-rate sub-filters of the transposed fast-FIR form,
each paying a full prologue / epilogue: pessimistic by half an epilogue).  Needs the probe build (-DTTSC_PROBE_EVENK: even K in the wide kernel)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ttscube_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, 'ttscube_amd', 'csrc', 'libttscube_hip_probe.so')
from ttscube_amd.hip_layers import Conv1dHip


def run(c, k, L, B, resid):
    conv = Conv1dHip(c, c, k, padding=(k - 1) // 2, dilation=1).set_precision('f16x3')
    conv.set_weight(torch.randn(c, c, k) / (c * k) ** 0.5, torch.randn(c) * 0.1)
    x = torch.randn(B, c, L, device='cuda')
    Lo = L + 2 * ((k - 1) // 2) - (k - 1)
    y = torch.empty(B, c, Lo, device='cuda')
    r = torch.randn_like(y) if resid else None
    for _ in range(3):
        conv(x, out=y, resid=r, in_slope=0.1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        conv(x, out=y, resid=r, in_slope=0.1)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 10


for c, L in ((256, 4032), (128, 12032)):
    for k in (7, 11):
        full = run(c, k, L, 64, 1)
        kh = (k + 1) // 2
        h0 = run(c, kh, L // 2, 64, 0)
        h1 = run(c, kh, L // 2, 64, 1)
        print('C=%d K=%d L=%d: full %.3f ms | half-rate K=%d: %.3f (no resid) %.3f (resid) -> mock fast-FIR %.3f ms = %.1f %% of full (MFMA ratio %.1f %%)'
              % (c, k, L, full, kh, h0, h1, h0 + 2 * h1, 100 * (h0 + 2 * h1) / full, 100 * 3 * kh / (2 * k)))
