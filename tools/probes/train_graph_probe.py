"""Feasibility probe: the fixed-shape part of Cubegan.training_step (generator forward on the cropped conditioning, mel loss, discriminator
step with its optimizer, generator step's backward down to the gradient of the conditioning crop) captured ONCE into a hipGraph (torch.cuda.graph)
and replayed, against the same function launched eagerly.  The eager step is host-bound (profiles/r06_train_host_bound.log: 71.8 ms of enqueue
work, 0.2 ms of waiting); a replay has no host work, so its time is what the GPU side of this part costs.
    python tools/probes/train_graph_probe.py [--batch 16] [--iters 10]
(probe only: lr and the AdamW step number are baked into the captured launches)"""
import argparse
import itertools
import os
import random
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=16)
    ap.add_argument('--iters', type=int, default=10)
    a = ap.parse_args()
    from bench_cubegan_step import make_batch
    from ttscube_amd.networks.cubegan import Cubegan
    from ttscube_amd.networks import training as T
    from ttscube_amd.hifigan.autograd import generator_forward_with_grad
    from ttscube_amd.hifigan.wbank import AmaxPool
    from ttscube_amd.io_utils.melspec import mel_spectrogram
    rng = np.random.RandomState(0)
    batch, enc = make_batch(a.batch, 40, rng)
    torch.manual_seed(0)
    model = Cubegan(enc, conditioning=None, train=True).cuda()
    model.train()
    opts = T.cubegan_configure_optimizers(model)
    opt_g, opt_d, opt_t, opt_b = opts
    r = random.Random(1)
    for _ in range(3):
        T.cubegan_training_step(model, batch, opts, rng=r)     # arenas built, workspaces allocated
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.iters):
        T.cubegan_training_step(model, batch, opts, rng=r)
    torch.cuda.synchronize()
    print('whole step, eager: %.1f ms' % ((time.perf_counter() - t0) / a.iters * 1e3), flush=True)

    dev = model.get_device()
    discriminator_loss, feature_loss, generator_loss = T._gan_loss_fns()
    mpd, msd = T._discriminator_fns(model)
    B = a.batch
    cond_in = (torch.randn(B, 50, 80, device=dev) - 2).clamp(-5, 1).requires_grad_(True)
    y_in = torch.rand(B, 1, 12000, device=dev) - 0.5
    out = {}

    def gan_part():
        AmaxPool.of(dev).reset()
        cond_in.grad = None
        y = y_in
        y_g_hat = generator_forward_with_grad(model._generator, cond_in.permute(0, 2, 1).contiguous())
        m = min(y.shape[2], y_g_hat.shape[2])
        y, y_g_hat = y[:, :, :m], y_g_hat[:, :, :m]
        y_mel = mel_spectrogram(y.squeeze(1), 1024, 80, 24000, 240, 1024, 0, 12000)
        y_g_hat_mel = mel_spectrogram(y_g_hat.squeeze(1), 1024, 80, 24000, 240, 1024, 0, 12000)
        opt_d.zero_grad()
        y_df_hat_r, y_df_hat_g, _, _ = mpd(y, y_g_hat.detach(), False)
        loss_disc_f, _, _ = discriminator_loss(y_df_hat_r, y_df_hat_g)
        y_ds_hat_r, y_ds_hat_g, _, _ = msd(y, y_g_hat.detach(), False)
        loss_disc_s, _, _ = discriminator_loss(y_ds_hat_r, y_ds_hat_g)
        loss_disc_all = loss_disc_s + loss_disc_f
        loss_disc_all.backward()
        opt_d.step()
        opt_g.zero_grad()
        loss_mel = F.l1_loss(y_mel, y_g_hat_mel) * 45
        d_params = [p for p in itertools.chain(model._mpd.parameters(), model._msd.parameters()) if p.requires_grad]
        for p in d_params:
            p.requires_grad_(False)
        try:
            y_df_hat_r, y_df_hat_g, fmap_f_r, fmap_f_g = mpd(y, y_g_hat)
            y_ds_hat_r, y_ds_hat_g, fmap_s_r, fmap_s_g = msd(y, y_g_hat)
            loss_gen_all = (generator_loss(y_ds_hat_g)[0] + generator_loss(y_df_hat_g)[0] + feature_loss(fmap_s_r, fmap_s_g)
                            + feature_loss(fmap_f_r, fmap_f_g) + loss_mel)
            loss_gen_all.backward()
        finally:
            for p in d_params:
                p.requires_grad_(True)
        out['loss_g'], out['loss_d'] = loss_gen_all.detach(), loss_disc_all.detach()

    for _ in range(3):
        gan_part()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.iters):
        gan_part()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('fixed-shape part, eager: %.1f ms per call (host enqueue %.1f ms)  loss_g %.4f loss_d %.4f'
          % ((t2 - t0) / a.iters * 1e3, (t1 - t0) / a.iters * 1e3, float(out['loss_g']), float(out['loss_d'])), flush=True)

    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    try:
        with torch.cuda.stream(s):
            gan_part()              # once more on the capture stream (allocations of this stream's pool)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.cuda.graph(g, stream=s, capture_error_mode=os.environ.get('PROBE_CAPTURE_MODE', 'thread_local')):
            gan_part()
        print('captured in %.1f s' % (time.perf_counter() - t0), flush=True)
    except Exception as e:      # noqa
        print('CAPTURE FAILED: %r' % (e,), flush=True)
        raise
    torch.cuda.synchronize()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.iters):
        g.replay()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('fixed-shape part, graph replay: %.1f ms per call (host %.2f ms)  loss_g %.4f loss_d %.4f  cond grad |max| %.3e'
          % ((t2 - t0) / a.iters * 1e3, (t1 - t0) / a.iters * 1e3, float(out['loss_g']), float(out['loss_d']),
             float(cond_in.grad.abs().max()) if cond_in.grad is not None else float('nan')), flush=True)


if __name__ == '__main__':
    main()
