import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import meldecoder_ref as M
from ttscube_amd.networks.textcoder import CubenetTextcoder


class Enc:
    phon2int = {'p%d' % i: i for i in range(50)}
    speaker2int = {'s0': 0}
    max_pitch = 300
    max_duration = 12


tc = CubenetTextcoder(Enc())
tc.load_state_dict(M.fill_state_dict(M.named_shapes(tc), 7))
tc = tc.cuda().eval()
x = np.random.RandomState(1234).randint(1, 51, size=(1, 30))
Xt = lambda: {'x_char': torch.from_numpy(x), 'x_speaker': torch.ones((1, 1), dtype=torch.long)}
for i in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    mel = tc.inference(Xt())
    torch.cuda.synchronize()
    print('run %d: %.2f ms (%d frames)' % (i, (time.perf_counter() - t0) * 1e3, mel.shape[1]), flush=True)
