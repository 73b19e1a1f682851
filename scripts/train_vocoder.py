#!/usr/bin/env python
"""Counterpart of the reference's scripts/train_vocoder.py (same flags, same files written), without pytorch_lightning:
one process per GPU (`python -m torch.distributed.run --nproc-per-node N scripts/train_vocoder.py ...`), explicit RCCL
gradient exchange (ttscube_amd/distributed.py).

Files written (train_vocoder.py:36-75 of the reference): <base>.yaml (num_layers_lr, layer_size_lr, num_layers_hr,
layer_size_hr, upsample, sample_rate, output, sample_rate_low, hop_size), <base>.lr.best / <base>.hr.best (bare WaveRNN
state_dicts), <base>.last (CubenetVocoder state_dict, prefixes _wavernn_hr. / _wavernn_lr.); --resume reloads <base>.last.

Data: `--synthetic N` trains on N seeded synthetic items (no dataset/librosa in this image); otherwise `--train-folder`
must contain the reference's cache files `<id>.mgc.npy / .audio.npy / .audio_low.npy` (cube/io_utils/io_vocoder.py:46-63)."""
import glob
import os
import sys
from argparse import ArgumentParser

import numpy as np
import torch
import torch.distributed as dist
import yaml

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ttscube_amd.distributed import FlatBucketReducer, broadcast_parameters  # noqa: E402
from ttscube_amd.networks import training as T  # noqa: E402
from ttscube_amd.networks.vocoder import CubenetVocoder  # noqa: E402


def _items(params, rank):
    if params.synthetic:
        rng = np.random.RandomState(1234 + rank)
        for _ in range(params.synthetic):
            n = params.maximum_segment_size
            x = (0.5 * np.sin(np.cumsum(rng.uniform(0.01, 0.2, size=n))) * rng.uniform(0.3, 1.0)).astype(np.float32)
            yield {'x': x, 'x_low': x[::params.sample_rate // params.sample_rate_low].copy(),
                   'mel': np.clip(rng.randn(n // params.hop_size + 1, 80) - 2, -5, 1).astype(np.float32)}
        return
    for f in sorted(glob.glob(os.path.join(params.train_folder, '*.mgc.npy'))):
        base = f[:-len('.mgc.npy')]
        yield {'x': np.load(base + '.audio.npy').astype(np.float32), 'x_low': np.load(base + '.audio_low.npy').astype(np.float32),
               'mel': np.load(f).astype(np.float32)}


def _collate(items):
    """VocoderCollate (io_vocoder.py:85-112): zero-pad audio, pad mel with -5."""
    L = max(i['x'].shape[0] for i in items)
    Ll = max(i['x_low'].shape[0] for i in items)
    F_ = max(i['mel'].shape[0] for i in items)
    x = np.zeros((len(items), L), dtype=np.float32)
    xl = np.zeros((len(items), Ll), dtype=np.float32)
    mel = np.ones((len(items), F_, 80), dtype=np.float32) * -5
    for k, i in enumerate(items):
        x[k, :i['x'].shape[0]] = i['x']
        xl[k, :i['x_low'].shape[0]] = i['x_low']
        mel[k, :i['mel'].shape[0]] = i['mel']
    return {'x': torch.from_numpy(x), 'x_low': torch.from_numpy(xl), 'mel': torch.from_numpy(mel)}


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
    items = list(_items(params, rank))
    best = {'lr': 9999.0, 'hr': 9999.0}
    for epoch in range(params.epochs):
        tot = {'lr': 0.0, 'hr': 0.0}
        nb = 0
        for s in range(0, len(items), params.batch_size):
            out = T.vocoder_training_step(model, _collate(items[s:s + params.batch_size]), opts, reducers)
            tot['lr'] += out['lr']
            tot['hr'] += out['hr']
            nb += 1
        if rank == 0:
            for k, net in (('lr', model._wavernn_lr), ('hr', model._wavernn_hr)):
                v = tot[k] / max(nb, 1)
                if v < best[k]:
                    best[k] = v
                    net.save('{0}.{1}.best'.format(params.output_base, k))
            model.save('{0}.last'.format(params.output_base))
            sys.stdout.write('epoch %d  loss_lr %.4f  loss_hr %.4f\n' % (epoch, tot['lr'] / max(nb, 1), tot['hr'] / max(nb, 1)))
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
    p.add_argument('--train-folder', dest='train_folder', default='data/cache/train')
    p.add_argument('--dev-folder', dest='dev_folder', default='data/cache/dev')
    p.add_argument('--sample-rate', dest='sample_rate', type=int, default=24000)
    p.add_argument('--sample-rate-low', dest='sample_rate_low', type=int, default=2400)
    p.add_argument('--layer-size-hr', dest='layer_size_hr', default=512, type=int)
    p.add_argument('--num-layers-hr', dest='num_layers_hr', default=1, type=int)
    p.add_argument('--layer-size-lr', dest='layer_size_lr', default=512, type=int)
    p.add_argument('--num-layers-lr', dest='num_layers_lr', default=1, type=int)
    p.add_argument('--hop-size', dest='hop_size', type=int, default=240)
    p.add_argument('--upsample', dest='upsample', default=240, type=int)
    p.add_argument('--lr', dest='lr', default=1e-4, type=float)
    p.add_argument('--output', dest='output', default='mulaw', help='mulaw|raw (the HIP sampler implements the discrete outputs)')
    p.add_argument('--resume', dest='resume', action='store_true')
    p.add_argument('--epochs', type=int, default=1)
    p.add_argument('--synthetic', type=int, default=0, help='train on N synthetic items per rank')
    _train(p.parse_args())
