"""One sentence through Cubegan.inference, 30 times (for rocprofv3: how much of the 9 ms is GPU work, how much is launching)."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import meldecoder_ref as M, hifigan_ref as R
from ttscube_amd.networks.cubegan import Cubegan


class Enc:
    phon2int = {'p%d' % i: i for i in range(50)}
    speaker2int = {'s0': 0}
    max_pitch = 300
    max_duration = 12


torch.manual_seed(0)
model = Cubegan(Enc(), conditioning=None, train=False)
sd = model.state_dict()
sd.update({'_languasito.' + k: v for k, v in M.fill_state_dict(M.named_shapes(model._languasito), 5).items()})
sd.update({'_generator.' + k: v for k, v in R.synthetic_state_dict(dict(R.CONFIG_V1), seed=6).items()})
model.load_state_dict(sd)
model = model.cuda().eval()
rng = np.random.RandomState(1234)
x = rng.randint(1, 51, size=(1, 67))
X = lambda: {'x_char': torch.from_numpy(x), 'x_speaker': torch.ones((1, 1), dtype=torch.long)}
for _ in range(5):
    model.inference(X())
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(30):
    w = model.inference(X())
torch.cuda.synchronize()
print('B=1: %.2f ms per sentence (%d samples)' % ((time.perf_counter() - t0) / 30 * 1e3, w.shape[2]))
