"""Per-step time of the LSTM recurrence kernels at the shapes of the text -> audio path (Languasito2: BiLSTM(256) x 2 layers over
phonemes, BiLSTM(64) x 2 layers over frames), for one sentence and for a batch:  python tools/bench_lstm.py
TTSC_LSTM_SPLIT=1 forces the single-workgroup kernel (no split over workgroups)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ttscube_amd.hip_layers import LSTMHip


def main():
    for H, insz, T in ((256, 256, 67), (64, 641, 340), (256, 640, 120), (64, 641, 700), (512, 640, 174)):
        for B in (1, 64):
            m = torch.nn.LSTM(input_size=insz, hidden_size=H, num_layers=2, bidirectional=True, batch_first=True).cuda()
            l = LSTMHip(m)
            x = torch.randn(B, T, insz, device='cuda')
            for _ in range(3):
                l(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 5
            for _ in range(n):
                l(x)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / n
            print('BiLSTM(%3d) x 2 layers, B=%2d, T=%3d: %.3f ms  (%.2f us per step and layer)' % (H, B, T, dt * 1e3, dt * 1e6 / (2 * T)), flush=True)


if __name__ == '__main__':
    main()
