"""Step time of the register-resident split LSTM recurrence (H = 256, one bidirectional layer) by batch size and utterances per member group
(ttsc_lstm_set_group_size): separates what the kernel structure costs from what chip-wide hand-off traffic costs."""
import os
import sys

import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ttscube_amd import _lib
from ttscube_amd.hip_layers import LSTMHip


def main():
    H, T = int(os.environ.get('PROBE_H', '256')), 200
    m = nn.LSTM(input_size=64, hidden_size=H, num_layers=1, bidirectional=True, batch_first=True).cuda()
    h = LSTMHip(m)
    for B in (2, 8, 16, 32, 64, 128):
        x = torch.randn(B, T, 64).cuda()
        row = []
        for nb in (1, 2, 4, 8):
            if nb > B:
                continue
            with _lib.lstm_group_size(nb):
                for _ in range(2):
                    h(x)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    h(x)
                e1.record()
                torch.cuda.synchronize()
            row.append('NB=%d %.2f us' % (nb, e0.elapsed_time(e1) / 5 / T * 1e3))
        print('H=%d B=%3d (%3d pairs): ' % (H, B, 2 * B) + '   '.join(row), flush=True)
    _lib.check_split_status('probe')


if __name__ == '__main__':
    main()
