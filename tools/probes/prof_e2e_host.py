import cProfile, pstats, sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from oracle import hifigan_ref as R
from oracle import meldecoder_ref as MO
from ttscube_amd.io_utils.synthetic import synthetic_sentences
from ttscube_amd.networks.cubegan import Cubegan
class _Enc:
    phon2int = {'p%d' % i: i for i in range(50)}; speaker2int = {'s0': 0}; max_pitch = 300; max_duration = 12
torch.manual_seed(0)
tts = Cubegan(_Enc(), conditioning=None, train=False)
sd = tts.state_dict()
sd.update({'_languasito.' + k: v for k, v in MO.fill_state_dict(MO.named_shapes(tts._languasito), 5).items()})
sd.update({'_generator.' + k: v for k, v in R.synthetic_state_dict(dict(R.CONFIG_V1), seed=1234).items()})
tts.load_state_dict(sd); tts = tts.cuda().eval()
xc, lens = synthetic_sentences(64, seed=1234)
mk = lambda: {'x_char': torch.from_numpy(xc), 'x_speaker': torch.ones((64, 1), dtype=torch.long)}
for _ in range(3): tts.inference(mk(), check='deferred')
torch.cuda.synchronize()
# host time of the text part alone (no sync at the end)
t0 = time.perf_counter()
for _ in range(5):
    cond, _, flens = tts._languasito.inference(mk(), return_aux=True, check_status=False)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print('languasito host ms per call %.2f, drain %.2f' % ((t1 - t0) / 5 * 1e3, (t2 - t1) * 1e3))
pr = cProfile.Profile(); pr.enable()
for _ in range(5):
    tts._languasito.inference(mk(), return_aux=True, check_status=False)
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats('tottime').print_stats(18)
