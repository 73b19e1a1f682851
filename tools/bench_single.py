#!/usr/bin/env python
"""Device time of the HiFi-GAN generator on one small batch (default: one 3 s utterance, BASELINE configs[0] shape), for
profiling the small-problem dispatch under rocprofv3:  python tools/bench_single.py [--batch 1 --frames 300 --steps 20]."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    p = argparse.ArgumentParser()
    p.add_argument('--batch', type=int, default=1)
    p.add_argument('--frames', type=int, default=300)
    p.add_argument('--steps', type=int, default=20)
    p.add_argument('--calib-batch', type=int, default=0, help='calibrate on a batch of this size first (0: on the timed input)')
    a = p.parse_args()
    import torch
    from oracle import hifigan_ref as R   # synthetic weights / inputs only
    from ttscube_amd.hifigan.env import AttrDict
    from ttscube_amd.hifigan.models import Generator
    h = dict(R.CONFIG_V1)
    g = Generator(AttrDict(h))
    g.load_state_dict(R.synthetic_state_dict(h, seed=1234))
    g = g.cuda().eval()
    mel = R.synthetic_mel(a.batch, a.frames, seed=77).cuda()
    with torch.no_grad():
        if a.calib_batch:
            g(R.synthetic_mel(a.calib_batch, 800, seed=1234).cuda())
        for _ in range(3):
            g(mel)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.steps):
            g(mel)
        e1.record()
        torch.cuda.synchronize()
    print('B=%d T=%d: %.3f ms per forward' % (a.batch, a.frames, e0.elapsed_time(e1) / a.steps))


if __name__ == '__main__':
    main()
