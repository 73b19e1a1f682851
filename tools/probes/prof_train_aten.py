"""torch-op view of one Cubegan training step (b = 16): which ATen ops (the glue between the HIP launches) cost device time.
    python tools/probes/prof_train_aten.py"""
import os
import random
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.getcwd())
from ttscube_amd.io_utils.io_cubegan import CubeganCollate  # noqa: E402
from ttscube_amd.io_utils.synthetic import synthetic_encodings, synthetic_examples  # noqa: E402
from ttscube_amd.networks import training as T  # noqa: E402
from ttscube_amd.networks.cubegan import Cubegan  # noqa: E402

dev = torch.device('cuda', 0)
enc = synthetic_encodings()
torch.manual_seed(1234)
model = Cubegan(enc, conditioning=None, train=True).to(dev)
model.train()
opts = T.cubegan_configure_optimizers(model)
batch = CubeganCollate(enc).collate_fn(list(synthetic_examples(16, 777, min_ph=30, max_ph=50)))
crop = random.Random(99)
for _ in range(3):
    T.cubegan_training_step(model, batch, opts, None, rng=crop)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False) as prof:
    T.cubegan_training_step(model, batch, opts, None, rng=crop)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by='self_cuda_time_total', row_limit=40, max_name_column_width=60))
