"""One full `Cubegan.training_step` (cubegan.py:85-189) per iteration on synthetic data at config C4's per-GPU size
(b utterances, 50-frame / 12 000-sample crops): discriminator step + generator step + text step, four optimizers.
    python tools/bench_cubegan_step.py [--batch 16] [--iters 5]"""
import argparse
import os
import random
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))


def make_batch(B, nph, rng):
    from ttscube_amd.io_utils.io_cubegan import CubeganCollate, CubeganEncodings
    enc = CubeganEncodings()
    enc.phon2int = {str(i): i for i in range(50)}
    enc.speaker2int = {'a': 0}
    enc.max_pitch, enc.max_duration = 300, 10
    ex = []
    for b in range(B):
        durs = rng.randint(3, 9, size=nph)
        f2p = [p for p, d in enumerate(durs) for _ in range(d)]
        F_ = len(f2p)
        ex.append({'meta': {'phones': [str(v) for v in rng.randint(0, 50, size=nph)], 'speaker': 'a', 'frame2phon': f2p,
                            'phon2word': [0] * nph},
                   'mgc': np.clip(rng.randn(F_, 80) - 2, -5, 1), 'pitch': rng.randint(0, 300, size=F_).astype(np.float64),
                   'audio': rng.uniform(-0.5, 0.5, size=F_ * 240)})
    return CubeganCollate(enc).collate_fn(ex), enc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=16)
    ap.add_argument('--iters', type=int, default=5)
    ap.add_argument('--ragged', action='store_true', help="bench.py's batch: synthetic_examples(b, 777, min_ph=30, max_ph=50) through the class surface")
    a = ap.parse_args()
    from ttscube_amd.networks.cubegan import Cubegan
    from ttscube_amd.networks import training as T
    rng = np.random.RandomState(0)
    if a.ragged:
        from ttscube_amd.io_utils.io_cubegan import CubeganCollate
        from ttscube_amd.io_utils.synthetic import synthetic_encodings, synthetic_examples
        enc = synthetic_encodings()
        batch = CubeganCollate(enc).collate_fn(list(synthetic_examples(a.batch, 777, min_ph=30, max_ph=50)))
    else:
        batch, enc = make_batch(a.batch, 40, rng)
    torch.manual_seed(0)
    model = Cubegan(enc, conditioning=None, train=True).cuda()
    model.train()
    opts = T.cubegan_configure_optimizers(model)
    r = random.Random(1)
    step = (lambda: model.training_step(batch, 0, rng=r)) if a.ragged else (lambda: T.cubegan_training_step(model, batch, opts, rng=r))
    if a.ragged:
        model._optimizers = opts
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.iters):
        out = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.iters
    print('cubegan training step  b=%d x 12000 samples: %.1f ms/step  %.2f M samples/s  losses %s' %
          (a.batch, dt * 1e3, a.batch * 12000 / dt / 1e6, {k: round(v, 4) for k, v in out.items()}), flush=True)


if __name__ == '__main__':
    main()
