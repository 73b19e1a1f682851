"""Workgroup phase timeline of the wide convolution kernel and of the chain kernel (measurement build: `make -C ttscube_amd/csrc ablate`).

Every workgroup's thread 0 stamps the 100 MHz wall clock at its phase boundaries (TTSC_STAMP in csrc/conv_kernels.hpp); this tool runs ONE
launch of a generator layer at BASELINE config[1] size, reads the records back and prints: mean / p10 / p90 duration of each phase, the
share of the launch's wall time each CU spends with 0 / 1 / 2 resident workgroups in their MFMA main loop, and the chip-wide count of
workgroups inside their prologue / main loop / epilogue over time (a coarse histogram), i.e. whether the memory phases of the workgroups
coincide.

    python tools/wg_timeline.py --stage 1 --k 3 [--resid 1]      # conv_f16x3_wide_kernel
    python tools/wg_timeline.py --stage 4 --k 3 [--acc 1]        # rbchain_f16x3_kernel
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ttscube_amd import _lib

_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), 'libttscube_hip_ablate.so')
from ttscube_amd.hip_layers import Conv1dHip  # noqa: E402

STAGES = {1: (256, 4001), 2: (128, 12004), 3: (64, 48016), 4: (32, 192064)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--stage', type=int, default=1)
    ap.add_argument('--k', type=int, default=3)
    ap.add_argument('--d', type=int, default=1)
    ap.add_argument('--B', type=int, default=64)
    ap.add_argument('--L', type=int, default=0)
    ap.add_argument('--resid', type=int, default=1)
    ap.add_argument('--acc', type=int, default=1)
    ap.add_argument('--shape', type=int, default=-1)
    ap.add_argument('--chain', type=int, default=-1, help='1: force the chain kernel, 0: a single wide convolution (default: chain for stages 3, 4)')
    a = ap.parse_args()
    Cc, L = STAGES[a.stage]
    L = a.L or L
    B, k = a.B, a.k
    L_ = _lib.lib()
    chain = a.chain if a.chain >= 0 else int(a.stage >= 3)
    torch.manual_seed(0)
    x = torch.randn(B, Cc, L, device='cuda')
    y = torch.zeros_like(x)
    NSLOT = 24
    prof = torch.zeros(1 << 16, NSLOT, dtype=torch.int64, device='cuda')
    if chain:
        c1s, c2s = [], []
        for d in (1, 3, 5):
            c1 = Conv1dHip(Cc, Cc, k, padding=d * (k - 1) // 2, dilation=d).set_precision('f16x3')
            c2 = Conv1dHip(Cc, Cc, k, padding=(k - 1) // 2).set_precision('f16x3')
            c1.set_weight(torch.randn(Cc, Cc, k) / (Cc * k) ** 0.5 * 0.5, torch.randn(Cc) * 0.1)
            c2.set_weight(torch.randn(Cc, Cc, k) / (Cc * k) ** 0.5 * 0.5, torch.randn(Cc) * 0.1)
            c1s.append(c1)
            c2s.append(c2)
        arr = lambda cs: (C.c_void_p * len(cs))(*[c._h for c in cs])
        a1, a2 = arr(c1s), arr(c2s)

        def run():
            _lib.check(L_.ttsc_rbchain_forward(a1, a2, 3, _lib.dev_ptr(x), B, L, _lib.dev_ptr(y), a.acc, None, a.shape, _lib.current_stream()), 'rbchain')
        names = ['prologue (x load, image)'] + [n for p in range(3) for n in ('conv1.%d' % p, 'epilogue1.%d' % p, 'conv2.%d' % p, 'epilogue2.%d + image' % p)] + ['final store']
        bounds = list(range(0, 14)) + [14]
        mfma_phases = {1 + 4 * p + q for p in range(3) for q in (0, 2)}   # indices into `names`
    else:
        conv = Conv1dHip(Cc, Cc, k, padding=a.d * (k - 1) // 2, dilation=a.d).set_precision('f16x3')
        conv.set_weight(torch.randn(Cc, Cc, k) / (Cc * k) ** 0.5, torch.randn(Cc) * 0.1)
        r = torch.randn_like(x) if a.resid else None

        def run():
            conv(x, out=y, resid=r, in_slope=0.1)
        names = ['prologue', 'main loop', 'epilogue']
        bounds = [0, 1, 2, 3]
        mfma_phases = {1}
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms_plain = e0.elapsed_time(e1) / 5
    os.environ['TTSC_PROF_PTR'] = str(prof.data_ptr())
    run()
    torch.cuda.synchronize()
    prof.zero_()
    e0.record()
    run()
    e1.record()
    torch.cuda.synchronize()
    ms_prof = e0.elapsed_time(e1)
    del os.environ['TTSC_PROF_PTR']
    P = prof.cpu().numpy()
    live = P[:, 0] > 0
    P = P[live]
    n = P.shape[0]
    t0 = P[:, 0].min()
    T = (P[:, bounds] - t0) * 0.01   # us
    end = T[:, -1].max()
    print('%s stage %d C=%d L=%d K=%d: %.3f ms per launch (%.3f ms with the stamps); %d workgroups, launch wall time %.1f us' % (
        'chain' if chain else 'wide conv', a.stage, Cc, L, k, ms_plain, ms_prof, n, end))
    dur = np.diff(T, axis=1)
    tot = T[:, -1] - T[:, 0]
    print('  workgroup lifetime: mean %.1f us  p10 %.1f  p90 %.1f' % (tot.mean(), np.percentile(tot, 10), np.percentile(tot, 90)))
    for i, nm in enumerate(names):
        d = dur[:, i]
        print('  %-26s mean %7.2f us  p10 %7.2f  p90 %7.2f   %5.1f%% of the lifetime' % (nm, d.mean(), np.percentile(d, 10), np.percentile(d, 90), 100 * d.mean() / tot.mean()))
    if chain and P[:, 16].max() > 0:
        d = (P[:, [16, 17, 18, 1]] - P[:, [0, 16, 17, 18]]) * 0.01
        for nm, col in zip(('start -> x loads issued (margins zeroed, first weights sent)', 'x loads issued -> landed', 'landed -> image written (wave 0)', 'image written -> barrier passed'), d.T):
            print('    prologue detail: %-62s mean %6.2f us  p10 %6.2f  p90 %6.2f' % (nm, col.mean(), np.percentile(col, 10), np.percentile(col, 90)))
    mf = sum(dur[:, i] for i in mfma_phases)
    print('  MFMA phases together: %.1f us = %.1f%% of the lifetime' % (mf.mean(), 100 * mf.mean() / tot.mean()))
    # chip-wide phase census over time
    nb = 60
    edges = np.linspace(0, end, nb + 1)
    mid = 0.5 * (edges[1:] + edges[:-1])
    in_mfma = np.zeros(nb)
    in_other = np.zeros(nb)
    for i in range(len(names)):
        s, e = T[:, i], T[:, i + 1]
        cnt = ((s[:, None] <= mid[None, :]) & (mid[None, :] < e[:, None])).sum(0)
        if i in mfma_phases:
            in_mfma += cnt
        else:
            in_other += cnt
    print('  census over time (workgroups in MFMA phases / in other phases), %d bins of %.1f us:' % (nb, end / nb))
    print('   mfma : ' + ' '.join('%3d' % v for v in in_mfma))
    print('   other: ' + ' '.join('%3d' % v for v in in_other))
    # per CU: time with 0 / 1 / 2+ workgroups in MFMA phases
    hw = P[:, 15]
    cu_key = ((hw >> 32) & 0xf) * 4096 + ((hw >> 13) & 0x7) * 256 + ((hw >> 12) & 0x1) * 64 + ((hw >> 8) & 0xf)   # xcc, se, sh, cu
    keys = np.unique(cu_key)
    fine = np.linspace(0, end, 2001)
    fm = 0.5 * (fine[1:] + fine[:-1])
    occ = np.zeros((3,))
    res = np.zeros((4,))
    for kcu in keys:
        sel = cu_key == kcu
        c_m = np.zeros(fm.shape[0])
        c_all = np.zeros(fm.shape[0])
        for i in range(len(names)):
            s, e = T[sel, i], T[sel, i + 1]
            cnt = ((s[:, None] <= fm[None, :]) & (fm[None, :] < e[:, None])).sum(0)
            c_all += cnt
            if i in mfma_phases:
                c_m += cnt
        for v in range(3):
            occ[v] += ((c_m == v) if v < 2 else (c_m >= 2)).mean()
        for v in range(4):
            res[v] += ((c_all == v) if v < 3 else (c_all >= 3)).mean()
    occ /= len(keys)
    res /= len(keys)
    print('  %d distinct CUs seen; share of the launch a CU has 0 / 1 / 2+ workgroups in an MFMA phase: %.2f / %.2f / %.2f; resident workgroups 0 / 1 / 2 / 3+: %.2f / %.2f / %.2f / %.2f' % (
        len(keys), occ[0], occ[1], occ[2], res[0], res[1], res[2], res[3]))


if __name__ == '__main__':
    main()
