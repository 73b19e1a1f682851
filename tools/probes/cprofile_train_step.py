import cProfile, pstats, random, sys, os, io
sys.path.insert(0, os.getcwd())
import torch
from ttscube_amd.io_utils.io_cubegan import CubeganCollate
from ttscube_amd.io_utils.synthetic import synthetic_encodings, synthetic_examples
from ttscube_amd.networks import training as T
from ttscube_amd.networks.cubegan import Cubegan
dev = torch.device('cuda', 0)
enc = synthetic_encodings()
torch.manual_seed(1234)
model = Cubegan(enc, conditioning=None, train=True).to(dev); model.train()
opts = T.cubegan_configure_optimizers(model)
batch = CubeganCollate(enc).collate_fn(list(synthetic_examples(16, 777, min_ph=30, max_ph=50)))
crop = random.Random(99)
for _ in range(3): T.cubegan_training_step(model, batch, opts, None, rng=crop)
torch.cuda.synchronize()
import time
t0=time.perf_counter()
for _ in range(3): T.cubegan_training_step(model, batch, opts, None, rng=crop)
torch.cuda.synchronize()
print('plain: %.1f ms/step' % ((time.perf_counter()-t0)/3*1e3))
pr = cProfile.Profile(); pr.enable()
for _ in range(3): T.cubegan_training_step(model, batch, opts, None, rng=crop)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO(); ps = pstats.Stats(pr, stream=s).sort_stats('tottime'); ps.print_stats(28); print(s.getvalue()[:6000])
s = io.StringIO(); ps = pstats.Stats(pr, stream=s).sort_stats('cumulative'); ps.print_stats(40); print(s.getvalue()[:7000])
