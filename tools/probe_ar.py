import os, sys, time, torch
sys.path.insert(0, '/root/repo')
import torch.nn as nn
from ttscube_amd.hip_layers import LSTMHip, linear_hip
m = nn.LSTM(input_size=1280, hidden_size=512, num_layers=2, batch_first=True).cuda()
h = LSTMHip(m)
x = torch.randn(1, 1, 1280).cuda()
y, st = h(x, return_state=True)
for T in (1, 1, 64):
    xx = torch.randn(1, T, 1280).cuda()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): y, st = h(xx, hx=st, return_state=True)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print('LSTM 2x512 T=%d: %.1f us per call (%.1f us/step)' % (T, dt * 1e6, dt * 1e6 / T))
w = torch.randn(256, 80).cuda(); b = torch.randn(256).cuda(); v = torch.randn(1, 1, 80).cuda()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): o = linear_hip(v, w, b, act='relu')
torch.cuda.synchronize(); print('linear 80->256 M=1: %.1f us' % ((time.perf_counter() - t0) / 50 * 1e6))
