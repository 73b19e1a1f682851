"""cProfile of the host side of the b = 16 Cubegan step (3 warm steps): where the ~59 ms of Python / launch calls per step go."""
import cProfile
import os
import pstats
import random
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.test_distributed_gpu import _cubegan_setup
from ttscube_amd.networks import training as T


def main():
    model, batch, _ = _cubegan_setup(777, nitems=int(os.environ.get('PROBE_B', '16')))
    opts = T.cubegan_configure_optimizers(model)
    rng = random.Random(3)
    for _ in range(3):
        T.cubegan_training_step(model, batch, opts, rng=rng)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(3):
        T.cubegan_training_step(model, batch, opts, rng=rng)
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats('tottime').print_stats(45)
    st.sort_stats('cumulative').print_stats(60)


if __name__ == '__main__':
    main()
