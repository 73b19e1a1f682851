#!/usr/bin/env python
"""Counterpart of the reference's scripts/train_cubegan.py (flags, files written), without pytorch_lightning: one
process per GPU under torch.distributed.run, explicit RCCL flat-bucket gradient exchange after each of the three
backward passes of the GAN step (cube/networks/cubegan.py:153,170,174).

Files (train_cubegan.py:38-91 of the reference): <base>.yaml {sample_rate, hop_size, conditioning}, <base>.encodings,
<base>.best / <base>.last (Cubegan state_dict), <base>.opt.last {'0'..'3': optimizer state, 'global_step'}; --resume
restores model AND optimizers (the reference's resume silently drops the optimizer state: attribute-name mismatch at
train_cubegan.py:135 vs cubegan.py:304).

Data: `--train-folder` / `--dev-folder` hold the reference's processed corpus (<id>.json / .mgc / .pitch / .wav, read by
io_utils.io_cubegan.CubeganDataset); every rank trains on its own slice of exactly ceil(N / world) items (`rank_shard`: wrap-padded
like DistributedSampler, so all ranks run the same number of steps and gradient exchanges), read and collated by `--num-workers`
background threads one batch ahead of the GPU step; the dev set is validated in rank shards and the loss sums are all-reduced.  `.best` is selected on the
DEV-set mel-L1 (Cubegan.validation_step / validation_epoch_end, cubegan.py:191-273), as the reference does.  With
`--synthetic N` the folders are ignored and N seeded synthetic examples per rank (+ N/4 for the dev set) are used instead —
that must be asked for explicitly: a missing or empty folder is an error, never a silent fall-back."""
import os
import random
import sys
from argparse import ArgumentParser

import numpy as np
import torch
import torch.distributed as dist
import yaml

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ttscube_amd.distributed import broadcast_parameters  # noqa: E402
from ttscube_amd.io_utils.io_cubegan import CubeganCollate, CubeganEncodings  # noqa: E402
from ttscube_amd.io_utils.loader import BatchLoader, equal_batches, rank_shard  # noqa: E402
from ttscube_amd.io_utils.synthetic import synthetic_examples  # noqa: E402
from ttscube_amd.networks import training as T  # noqa: E402
from ttscube_amd.networks.cubegan import Cubegan  # noqa: E402


def _train(params):
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    dev = torch.device('cuda', local)
    # The discriminators' strided / grouped convolutions run on MIOpen.  --miopen-find lets it search its algorithms exhaustively
    # (torch.backends.cudnn.benchmark: minutes at start-up, then 108 instead of 145 ms per b=16 step on MI355X).  Opt-in: the search
    # repeats for every new tensor shape, and the text encoder's torch convolutions see a new length with almost every batch.
    torch.backends.cudnn.benchmark = bool(getattr(params, 'miopen_find', False))
    torch.cuda.set_device(dev)
    if world > 1:
        dist.init_process_group('nccl', device_id=dev)
    conditioning = params.lm if params.lm not in (None, 'none') else None
    if params.synthetic:
        trainset = list(synthetic_examples(params.synthetic, 1234 + rank))          # rank-distinct data and crops, same count on every rank
        devset = list(synthetic_examples(max(2, params.synthetic // 4), 4321))
        my_items = list(range(len(trainset)))
        enc_source = list(synthetic_examples(params.synthetic, 1234)) if world > 1 else trainset
    else:
        from ttscube_amd.io_utils.io_cubegan import CubeganDataset
        for folder in (params.train_folder, params.dev_folder):
            if not os.path.isdir(folder):
                raise SystemExit('%s does not exist (pass --synthetic N to train on synthetic examples)' % folder)
        trainset, devset = CubeganDataset(params.train_folder), CubeganDataset(params.dev_folder)
        if len(trainset) == 0 or len(devset) == 0:
            raise SystemExit('no <id>.json/.mgc/.pitch/.wav items under %s / %s' % (params.train_folder, params.dev_folder))
        my_items = rank_shard(len(trainset), rank, world)             # this rank's slice: same length on every rank
        enc_source = trainset.meta_items() if not params.resume else []   # JSON + .pitch only: no wav / mgc decode for the encodings
    my_dev = list(range(rank, len(devset), world))                     # validation shard (sums are all-reduced below)
    enc = CubeganEncodings()
    if params.resume:
        enc.load('{0}.encodings'.format(params.output_base))
    else:
        enc.compute(enc_source)
    if rank == 0:
        yaml.dump({'sample_rate': params.sample_rate, 'hop_size': params.hop_size, 'conditioning': conditioning},
                  open('{0}.yaml'.format(params.output_base), 'w'))
        enc.save('{0}.encodings'.format(params.output_base))
    model = Cubegan(enc, lr=params.lr, conditioning=conditioning, train=True)
    if params.resume:
        model.load('{0}.last'.format(params.output_base))
        st = torch.load('{0}.opt.last'.format(params.output_base), map_location='cpu')
        model._global_step = st['global_step']
        model._loaded_optimizer_states = st
    model = model.to(dev)
    broadcast_parameters(model)
    opts = T.cubegan_configure_optimizers(model)
    reducers = T.cubegan_reducers(model, opts) if world > 1 else None   # reduce_scatters leave from bucket-ready gradient hooks
    collate = CubeganCollate(enc)
    crop_rng = random.Random(99 + rank)
    best = 9999.0
    val_rng = random.Random(7)
    for epoch in range(params.epochs):
        mel_loss, nb, prev = 0.0, 0, None
        order = list(my_items)
        random.Random(1000 * epoch + rank).shuffle(order)
        for batch in BatchLoader(trainset, equal_batches(order, params.batch_size), collate.collate_fn, params.num_workers):
            out = T.cubegan_training_step(model, batch, opts, reducers, rng=crop_rng)
            # the step's losses are read back when first looked at (training.StepLosses): look at the PREVIOUS step's after queueing this one, so
            # that the host never waits for the GPU inside the loop
            if prev is not None:
                mel_loss += prev['loss_mel']
            prev = out
            nb += 1
        if prev is not None:
            mel_loss += prev['loss_mel']
            prev = None
        # validation (train_cubegan.py:38-76 + cubegan.py:191-273): mean dev-set mel-L1 -> _val_loss -> .best.  Every rank validates
        # its shard of the dev set (nobody idles in a barrier long enough to trip the RCCL watchdog); (sum, count) are all-reduced.
        model.eval()
        vals = [T.cubegan_validation_step(model, b, rng=val_rng)['loss_mel']
                for b in BatchLoader(devset, equal_batches(my_dev, params.batch_size), collate.collate_fn, params.num_workers)]
        model.train()
        vs = torch.tensor([sum(vals), float(len(vals))], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(vs)
        model._val_loss = float(vs[0] / max(float(vs[1]), 1.0))
        if rank == 0:
            if model._val_loss < best:
                best = model._val_loss
                model.save('{0}.best'.format(params.output_base))
            model.save('{0}.last'.format(params.output_base))
            od = {str(i): o.state_dict() for i, o in enumerate(opts)}
            od['global_step'] = model._global_step
            torch.save(od, '{0}.opt.last'.format(params.output_base))
            sys.stdout.write('epoch %d  train mel-L1 %.4f  val mel-L1 %.4f  step %d  lr %.3e\n' %
                             (epoch, mel_loss / max(nb, 1), model._val_loss, model._global_step, model._current_lr))
            if params.generate_epoch and epoch % params.generate_epoch == 0 and not params.synthetic:
                from ttscube_amd.io_utils.runtime import cubegan_synthesize_dataset
                model.eval()
                # bounded: the other ranks wait in the barrier below while rank 0 synthesises (default 16 files, not the whole dev set)
                cubegan_synthesize_dataset(model, output_path='generated_files/free/', devset_path=params.dev_folder,
                                           limit=params.generate_limit, conditioning=conditioning)
                model.train()
        if world > 1:
            dist.barrier()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    p = ArgumentParser(description='Cubegan trainer (reference flags)')
    p.add_argument('--output-base', dest='output_base', default='data/cubegan')
    p.add_argument('--batch-size', dest='batch_size', default=16, type=int)
    p.add_argument('--num-workers', dest='num_workers', default=4, type=int)
    p.add_argument('--accelerator', dest='accelerator', default='gpu')
    p.add_argument('--devices', dest='devices', default=1, type=int)
    p.add_argument('--train-folder', dest='train_folder', default='data/processed/train')
    p.add_argument('--dev-folder', dest='dev_folder', default='data/processed/dev')
    p.add_argument('--sample-rate', dest='sample_rate', type=int, default=24000)
    p.add_argument('--hop-size', dest='hop_size', type=int, default=240)
    p.add_argument('--lr', dest='lr', default=2e-4, type=float)
    p.add_argument('--lm', dest='lm', default=None, help='external conditioning (none | fasttext:<lang> | hf:<model>); only none is built')
    p.add_argument('--resume', dest='resume', action='store_true')
    p.add_argument('--epochs', type=int, default=1)
    p.add_argument('--miopen-find', dest='miopen_find', action='store_true', help='let MIOpen search its convolution algorithms exhaustively (see _train)')
    p.add_argument('--synthetic', type=int, default=0, help='ignore the folders and train on N seeded synthetic examples per rank')
    p.add_argument('--generate-limit', dest='generate_limit', type=int, default=16, help='dev files synthesised by --generate-epoch (-1 = all)')
    p.add_argument('--generate-epoch', dest='generate_epoch', type=int, default=0, help='synthesise the dev set every N epochs (0 = never)')
    _train(p.parse_args())
