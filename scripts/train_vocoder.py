#!/usr/bin/env python
"""Counterpart of the reference's scripts/train_vocoder.py (same flags, same files written), without pytorch_lightning:
one process per GPU (`python -m torch.distributed.run --nproc-per-node N scripts/train_vocoder.py ...`), explicit RCCL
gradient exchange (ttscube_amd/distributed.py).

Files written (train_vocoder.py:36-75 of the reference): <base>.yaml (num_layers_lr, layer_size_lr, num_layers_hr,
layer_size_hr, upsample, sample_rate, output, sample_rate_low, hop_size), <base>.lr.best / <base>.hr.best (bare WaveRNN
state_dicts), <base>.last (CubenetVocoder state_dict, prefixes _wavernn_hr. / _wavernn_lr.); --resume reloads <base>.last.

Data: `--train-folder` / `--dev-folder` hold .wav files; io_utils.io_vocoder.VocoderDataset reads them exactly as the reference's
(normalise to 0.98 peak, low-rate copy, log10-mel — computed on the GPU here —, `data/cache` files, random hop-aligned crops
of `--maximum-segment-size` samples).  Every rank trains on its own slice of exactly ceil(N / world) files (`rank_shard`: wrap-padded,
so all ranks run the same number of steps and gradient exchanges), loaded by `--num-workers` background threads; `.lr.best` / `.hr.best` are
selected on the dev-set losses (train_vocoder.py:36-59 of the reference).  `--synthetic N` ignores the folders and uses N seeded
synthetic items per rank; it must be asked for explicitly — a missing or empty folder is an error."""
import os
import random
import sys
from argparse import ArgumentParser

import numpy as np
import torch
import torch.distributed as dist
import yaml

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ttscube_amd.distributed import FlatBucketReducer, broadcast_parameters  # noqa: E402
from ttscube_amd.networks import training as T  # noqa: E402
from ttscube_amd.io_utils.io_vocoder import VocoderCollate  # noqa: E402
from ttscube_amd.io_utils.loader import BatchLoader, equal_batches, rank_shard  # noqa: E402
from ttscube_amd.networks.vocoder import CubenetVocoder  # noqa: E402


def _synthetic_items(params, n, seed):
    rng = np.random.RandomState(seed)
    out = []
    for _ in range(n):
        m = params.maximum_segment_size
        x = (0.5 * np.sin(np.cumsum(rng.uniform(0.01, 0.2, size=m))) * rng.uniform(0.3, 1.0)).astype(np.float32)
        out.append((x, x[::params.sample_rate // params.sample_rate_low].copy(),
                    np.clip(rng.randn(m // params.hop_size + 1, 80) - 2, -5, 1).astype(np.float32)))
    return out


class _RankSlice:
    """this rank's ceil(N / world) items of a dataset (wrap-padded: every rank gets the same count -> same number of steps)"""

    def __init__(self, ds, rank, world):
        self.ds, self.idx = ds, rank_shard(len(ds), rank, world)

    def __len__(self):
        return len(self.idx)

    def __getitem__(self, i):
        return self.ds[self.idx[i]]


def _datasets(params, rank, world):
    if params.synthetic:
        return _synthetic_items(params, params.synthetic, 1234 + rank), _synthetic_items(params, max(2, params.synthetic // 4), 4321)
    from ttscube_amd.io_utils.io_vocoder import VocoderDataset
    for folder in (params.train_folder, params.dev_folder):
        if not os.path.isdir(folder):
            raise SystemExit('%s does not exist (pass --synthetic N to train on synthetic items)' % folder)
    from ttscube_amd.io_utils.vocoder import MelVocoder
    kw = dict(target_sample_rate=params.sample_rate, lowres_sample_rate=params.sample_rate_low, hop_size=params.hop_size,
              mel_vocoder=MelVocoder('cuda:%d' % int(os.environ.get('LOCAL_RANK', '0'))))   # cache-miss features on THIS rank's GPU
    train = VocoderDataset(params.train_folder, max_segment_size=params.maximum_segment_size, random_start=True, **kw)
    dev = VocoderDataset(params.dev_folder, max_segment_size=params.maximum_segment_size, random_start=False, **kw)
    if len(train) == 0 or len(dev) == 0:
        raise SystemExit('no usable .wav files under %s / %s' % (params.train_folder, params.dev_folder))
    return _RankSlice(train, rank, world), dev


def _train(params):
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    dev = torch.device('cuda', local)
    torch.cuda.set_device(dev)
    if world > 1:
        dist.init_process_group('nccl', device_id=dev)
    config = {'num_layers_lr': params.num_layers_lr, 'layer_size_lr': params.layer_size_lr, 'num_layers_hr': params.num_layers_hr,
              'layer_size_hr': params.layer_size_hr, 'upsample': params.upsample, 'sample_rate': params.sample_rate,
              'output': params.output, 'sample_rate_low': params.sample_rate_low, 'hop_size': params.hop_size}
    if rank == 0:
        yaml.dump(config, open('{0}.yaml'.format(params.output_base), 'w'))
    model = CubenetVocoder(num_layers_hr=params.num_layers_hr, layer_size_hr=params.layer_size_hr, num_layers_lr=params.num_layers_lr,
                           layer_size_lr=params.layer_size_lr, upsample=params.upsample,
                           upsample_low=params.sample_rate // params.sample_rate_low, learning_rate=params.lr, output=params.output)
    if params.resume:
        model.load('{0}.last'.format(params.output_base))
    model = model.to(dev)
    broadcast_parameters(model)
    opts = (torch.optim.Adam(model._wavernn_lr.parameters(), lr=params.lr), torch.optim.Adam(model._wavernn_hr.parameters(), lr=params.lr))
    reducers = (FlatBucketReducer(model._wavernn_lr.parameters()), FlatBucketReducer(model._wavernn_hr.parameters())) if world > 1 else None
    train, dev_items = _datasets(params, rank, world)
    collate = VocoderCollate().collate_fn
    best = {'lr': 9999.0, 'hr': 9999.0}
    for epoch in range(params.epochs):
        tot = {'lr': 0.0, 'hr': 0.0}
        nb = 0
        order = list(range(len(train)))
        random.Random(1000 * epoch + rank).shuffle(order)
        for batch in BatchLoader(train, equal_batches(order, params.batch_size), collate, params.num_workers):
            out = T.vocoder_training_step(model, batch, opts, reducers)
            tot['lr'] += out['lr']
            tot['hr'] += out['hr']
            nb += 1
        # validation (WaveRNN.validation_step, modules.py:541-551): teacher-forced loss on the dev set, no gradient; every rank takes
        # its shard of the dev set and the (sums, count) are all-reduced
        model.eval()
        vsum = torch.zeros(3, dtype=torch.float64, device=dev)
        with torch.no_grad():
            for b in BatchLoader(dev_items, equal_batches(list(range(rank, len(dev_items), world)), params.batch_size), collate, params.num_workers):
                b = {k: v.to(dev) for k, v in b.items()}
                vsum[0] += float(T.wavernn_loss(model._wavernn_hr, {'x': b['x'], 'x_low': b['x_low'], 'mel': b['mel']}))
                vsum[1] += float(T.wavernn_loss(model._wavernn_lr, {'x': b['x_low'], 'mel': b['mel']}))
                vsum[2] += 1
        model.train()
        if world > 1:
            dist.all_reduce(vsum)
        val, nv = {'hr': float(vsum[0]), 'lr': float(vsum[1])}, int(vsum[2])
        if rank == 0:
            for k, net in (('lr', model._wavernn_lr), ('hr', model._wavernn_hr)):
                v = val[k] / max(nv, 1)
                if v < best[k]:
                    best[k] = v
                    net.save('{0}.{1}.best'.format(params.output_base, k))
            model.save('{0}.last'.format(params.output_base))
            sys.stdout.write('epoch %d  train lr %.4f hr %.4f   val lr %.4f hr %.4f\n' % (epoch, tot['lr'] / max(nb, 1), tot['hr'] / max(nb, 1),
                                                                                          val['lr'] / max(nv, 1), val['hr'] / max(nv, 1)))
        if world > 1:
            dist.barrier()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    p = ArgumentParser(description='WaveRNN vocoder trainer (reference flags)')
    p.add_argument('--output-base', dest='output_base', default='data/vocoder')
    p.add_argument('--batch-size', dest='batch_size', default=16, type=int)
    p.add_argument('--num-workers', dest='num_workers', default=4, type=int)
    p.add_argument('--maximum-segment-size', dest='maximum_segment_size', type=int, default=24000)
    p.add_argument('--accelerator', dest='accelerator', default='gpu')
    p.add_argument('--devices', dest='devices', default=1, type=int)
    p.add_argument('--train-folder', dest='train_folder', default='data/processed/train')
    p.add_argument('--dev-folder', dest='dev_folder', default='data/processed/dev')
    p.add_argument('--sample-rate', dest='sample_rate', type=int, default=24000)
    p.add_argument('--sample-rate-low', dest='sample_rate_low', type=int, default=2400)
    p.add_argument('--layer-size-hr', dest='layer_size_hr', default=512, type=int)
    p.add_argument('--num-layers-hr', dest='num_layers_hr', default=1, type=int)
    p.add_argument('--layer-size-lr', dest='layer_size_lr', default=512, type=int)
    p.add_argument('--num-layers-lr', dest='num_layers_lr', default=1, type=int)
    p.add_argument('--hop-size', dest='hop_size', type=int, default=240)
    p.add_argument('--upsample', dest='upsample', default=240, type=int)
    p.add_argument('--lr', dest='lr', default=1e-4, type=float)
    p.add_argument('--output', dest='output', default='mol', help='mol|gm|beta|mulaw|raw (cube/networks/loss.py)')
    p.add_argument('--resume', dest='resume', action='store_true')
    p.add_argument('--epochs', type=int, default=1)
    p.add_argument('--synthetic', type=int, default=0, help='ignore the folders and train on N seeded synthetic items per rank')
    _train(p.parse_args())
