"""Timing probe for BASELINE configs[2]: WaveRNN dual-GRU decode, batch=256 utterances, persistent kernel.
    python tools/bench_wavernn.py [--frames 100] [--batch 256] [--layers 1]"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import wavernn_ref as O  # synthetic weights/inputs only
from ttscube_amd.networks.modules import WaveRNN


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=256)
    ap.add_argument('--frames', type=int, default=10)
    ap.add_argument('--layers', type=int, default=1)
    ap.add_argument('--H', type=int, default=512)
    ap.add_argument('--ablate', action='store_true', help='load the measurement build (make -C ttscube_amd/csrc ablate); with TTSC_WT_PROF=1 the tile kernel prints its per-phase times')
    ap.add_argument('--lib', default=None, help='load this build of the library instead (A/B experiments)')
    a = ap.parse_args()
    if a.lib:
        from ttscube_amd import _lib
        _lib.LIB_PATH = os.path.abspath(a.lib)
    if a.ablate:
        from ttscube_amd import _lib
        _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), 'libttscube_hip_ablate.so')
    for lowres in (True, False):
        up = 240 if lowres else 24
        sd = O.synthetic_state_dict(H=a.H, num_layers=a.layers, use_lowres=lowres, seed=1)
        net = WaveRNN(num_layers=a.layers, layer_size=a.H, upsample=up, use_lowres=lowres, output='mulaw')
        net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        net = net.cuda().eval()
        mel, x_low = O.synthetic_inputs(a.batch, a.frames, seed=2, upsample=up)
        X = {'mel': torch.from_numpy(mel).cuda(), 'x_low': torch.from_numpy(x_low).cuda()}
        net.decode(X, mode='philox', seed=1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        idx, wav, _ = net.decode(X, mode='philox', seed=2)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        L = idx.shape[1]
        print('%s net: B=%d L=%d steps  %.3f s  %.2f us/step  %.3f M samples/s' % (
            'hr' if lowres else 'lr', a.batch, L, dt, dt / L * 1e6, a.batch * L / dt / 1e6), flush=True)


if __name__ == '__main__':
    main()
