"""Per-layer cost of the native discriminators at the training shapes (B = 16 crops of 8192 samples, real + generated = 32 sequences):
forward, data gradient and weight gradient of every convolution, timed apart with device events.  Run on a GPU box:
    python tools/probes/prof_disc_layers.py [--batch 32]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from ttscube_amd.hifigan import disc_hip  # noqa: E402
from ttscube_amd.hifigan.discriminators import MultiPeriodDiscriminator, MultiScaleDiscriminator  # noqa: E402


def timed(fn, reps=8):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--samples', type=int, default=8192)
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    mpd, msd = MultiPeriodDiscriminator().to(dev), MultiScaleDiscriminator().to(dev)
    y = torch.randn(args.batch, 1, args.samples, device=dev) * 0.1
    tot = {'fwd': 0.0, 'bwd': 0.0}
    print('%-6s %-3s %5s %5s %3s %2s %4s %7s %8s %8s %8s %8s' % ('disc', 'lyr', 'Cin', 'Cout', 'K', 's', 'grp', 'Lout', 'GFLOP', 'fwd ms', 'bwd ms', 'fwd TF/s'))
    for kind, discs in (('p', mpd.discriminators), ('s', msd.discriminators)):
        for d in discs:
            x = disc_hip._fold(y, d.period) if kind == 'p' else y
            hl = disc_hip._layers(d, kind)
            mods = list(d.convs) + [d.conv_post]
            slope = 1.0
            for i, (l, h) in enumerate(zip(mods, hl)):
                with torch.no_grad():
                    w = disc_hip._weight(l)
                    if w.dim() == 4:
                        w = w.squeeze(-1)
                    w = w.detach().clone()
                xin = x.detach().clone().requires_grad_(True)
                wq = w.clone().requires_grad_(True)
                bq = l.bias.detach().clone().requires_grad_(True)
                with torch.no_grad():
                    out = h(xin, wq, bq, in_slope=slope)
                t_f = timed(lambda: h(xin.detach(), wq.detach(), bq.detach(), in_slope=slope))
                o = h(xin, wq, bq, in_slope=slope)
                g = torch.randn_like(o)

                def bwd():
                    torch.autograd.grad(o, (xin, wq, bq), g, retain_graph=True)
                t_b = timed(bwd)
                P = d.period if kind == 'p' else 1
                Lout = out.shape[2] // P
                gf = 2.0 * out.shape[0] * out.shape[1] * out.shape[2] * (l.in_channels // l.groups) * l.kernel_size[0] / 1e9
                print('%-6s %-3d %5d %5d %3d %2d %4d %7d %8.2f %8.3f %8.3f %8.1f' % (
                    ('P%d' % d.period) if kind == 'p' else 'S', i, l.in_channels, l.out_channels, l.kernel_size[0], l.stride[0], l.groups,
                    out.shape[2], gf, t_f, t_b, gf / t_f))
                tot['fwd'] += t_f
                tot['bwd'] += t_b
                x = out
                slope = disc_hip.LRELU_SLOPE
            if kind == 's':
                y = torch.nn.functional.avg_pool1d(y, 4, 2, padding=2)
    print('sum: forward %.2f ms, backward (dgrad + wgrad + bias) %.2f ms at batch %d' % (tot['fwd'], tot['bwd'], args.batch))


if __name__ == '__main__':
    main()
