"""Generator forward on small batches (ms per forward, 50 back-to-back calls with the deferred range guard): for the branch-schedule thresholds."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from oracle import hifigan_ref as R
from ttscube_amd.hifigan.env import AttrDict
from ttscube_amd.hifigan.models import Generator

h = dict(R.CONFIG_V1)
g = Generator(AttrDict(h))
g.load_state_dict(R.synthetic_state_dict(h, seed=1234))
g = g.cuda().eval()
out = []
for B, T in ((1, 300), (2, 300), (4, 300), (8, 400), (16, 400), (4, 800)):
    mel = R.synthetic_mel(B, T, seed=7).cuda()
    ms, _ = bench.time_forward(g, mel, 50, 10, check='deferred')
    out.append('%dx%d %.3f' % (B, T, ms))
print('tiles=%s  ' % os.environ.get('TTSC_CHAIN_BRANCH_TILES', '256') + '  '.join(out))
