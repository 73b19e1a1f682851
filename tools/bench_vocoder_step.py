"""One `CubenetVocoder.training_step` (vocoder.py:136-156) on synthetic data at the reference's batch shape
(`x [16, 24000]`, `x_low [16, 2400]`, `mel [16, 101, 80]`): teacher-forced lr + hr WaveRNN, CE loss, backward, two Adam steps.
    python tools/bench_vocoder_step.py [--batch 16] [--frames 100] [--iters 2] [--torch-gru]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=16)
    ap.add_argument('--frames', type=int, default=100)
    ap.add_argument('--iters', type=int, default=2)
    ap.add_argument('--torch-gru', action='store_true', help='run the GRUs through torch.nn.GRU (MIOpen) for comparison')
    a = ap.parse_args()
    from ttscube_amd.networks.vocoder import CubenetVocoder
    from ttscube_amd.networks import training as T
    if a.torch_gru:
        T.gru_forward_train = lambda m, x: m(x)[0]
    torch.manual_seed(0)
    voc = CubenetVocoder(num_layers_lr=1, layer_size_lr=512, num_layers_hr=1, layer_size_hr=512, upsample=240, upsample_low=10,
                         learning_rate=1e-4, output='mulaw').cuda()
    rng = np.random.RandomState(0)
    B, F_ = a.batch, a.frames
    x = torch.from_numpy(rng.uniform(-0.9, 0.9, size=(B, F_ * 240)).astype(np.float32))
    batch = {'x': x, 'x_low': x[:, ::10].contiguous(), 'mel': torch.from_numpy(np.clip(rng.randn(B, F_ + 1, 80) - 2, -5, 1).astype(np.float32))}
    opts = (torch.optim.Adam(voc._wavernn_lr.parameters(), lr=1e-4), torch.optim.Adam(voc._wavernn_hr.parameters(), lr=1e-4))
    out = T.vocoder_training_step(voc, batch, opts)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.iters):
        out = T.vocoder_training_step(voc, batch, opts)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.iters
    print('%s vocoder training step  B=%d x %d samples: %.1f ms/step  %.3f M samples/s  %s' %
          ('torch-GRU' if a.torch_gru else 'hip-GRU  ', B, F_ * 240, dt * 1e3, B * F_ * 240 / dt / 1e6, {k: round(v, 4) for k, v in out.items()}), flush=True)


if __name__ == '__main__':
    main()
