"""Generator training leg of config C4 (BASELINE.json configs[3]: crops of 50 frames -> 12 000 samples, b per GPU):
forward + backward through the HIP autograd path vs the torch-op formulation on the same device.
    python tools/bench_train.py [--batch 16] [--frames 50] [--iters 10]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from oracle import hifigan_ref as R  # noqa: E402  (synthetic weights only)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=16)
    ap.add_argument('--frames', type=int, default=50)
    ap.add_argument('--iters', type=int, default=10)
    ap.add_argument('--only', default='')
    a = ap.parse_args()
    from ttscube_amd.hifigan.env import AttrDict
    from ttscube_amd.hifigan.models import Generator
    from ttscube_amd.hifigan.autograd import generator_forward_with_grad
    from tests.torch_reference import generator_forward_train   # the torch-op formulation (test infrastructure)
    h = dict(R.CONFIG_V1)
    g = Generator(AttrDict(h))
    g.load_state_dict(R.synthetic_state_dict(h, seed=1))
    g = g.cuda()
    mel = R.synthetic_mel(a.batch, a.frames, seed=2).cuda().requires_grad_(True)
    params = [p for p in g.parameters() if p.requires_grad]
    samples = a.batch * a.frames * 240
    flops = 3 * 1168559.0 * samples
    for name, fn in (('hip', generator_forward_with_grad), ('torch', generator_forward_train)):
        if a.only and a.only != name:
            continue
        def step():
            y = fn(g, mel)
            torch.autograd.grad(y.abs().mean(), [mel] + params)
        step()
        step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.iters):
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.iters
        print('%-5s generator fwd+bwd  B=%d x %d frames: %.2f ms/step  %.2f M samples/s  %.1f TFLOP/s (3x fwd)' %
              (name, a.batch, a.frames, dt * 1e3, samples / dt / 1e6, flops / dt / 1e12), flush=True)


if __name__ == '__main__':
    main()
