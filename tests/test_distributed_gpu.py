"""GPU: the RCCL calls of the gradient exchange (reduce_scatter_tensor + all_gather_into_tensor on backend "nccl"), run on
the one GPU a test box has (world_size 1, exchange forced) — the same code path `bench.py --mode train --gpus N` and the
training scripts take at N > 1; the world_size-2 arithmetic is covered on gloo in test_distributed_cpu.py."""
import os
import socket

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize('use_rs', [True, False])
def test_rccl_reduce_scatter_all_gather_world1(use_rs):
    from ttscube_amd.distributed import FlatBucketReducer, broadcast_parameters
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(_free_port())
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
    try:
        torch.manual_seed(3)
        net = torch.nn.Sequential(torch.nn.Linear(37, 29), torch.nn.Tanh(), torch.nn.Linear(29, 11)).to(dev)
        frozen = torch.nn.Linear(2, 2).to(dev)
        broadcast_parameters(net)   # no-op at world 1, must not raise
        x = torch.randn(16, 37, device=dev)
        net(x).pow(2).mean().backward()
        want = [p.grad.clone() for p in net.parameters()]
        red = FlatBucketReducer(list(net.parameters()) + list(frozen.parameters()), bucket_mb=0.001, use_reduce_scatter=use_rs,
                                force=True)
        red.reduce()
        red.reduce()   # second call reuses the buckets
        torch.cuda.synchronize()
        for p, w in zip(net.parameters(), want):
            assert torch.equal(p.grad, w)   # mean over a world of one is the identity, bit for bit
        assert all(p.grad is None for p in frozen.parameters())
        assert len(red._buckets) > 1 and red.bytes_exchanged > 0
    finally:
        dist.destroy_process_group()


def _cubegan_setup(seed, nitems=4):
    import random

    import numpy as np
    from ttscube_amd.io_utils.io_cubegan import CubeganCollate
    from ttscube_amd.io_utils.synthetic import synthetic_encodings, synthetic_examples
    from ttscube_amd.networks.cubegan import Cubegan
    enc = synthetic_encodings()
    torch.manual_seed(1234)
    model = Cubegan(enc, conditioning=None, train=True).cuda()
    model.train()
    batch = CubeganCollate(enc).collate_fn(list(synthetic_examples(nitems, seed, min_ph=30, max_ph=50)))
    return model, batch, random


def _run_steps(model, batch, random, nsteps, with_exchange):
    """`nsteps` Cubegan steps; returns the gradient arenas' checksums per step and the final parameters"""
    from ttscube_amd.networks import training as T
    opts = T.cubegan_configure_optimizers(model)
    reds = T.cubegan_reducers(model, opts, force=True, overlap=True, bucket_mb=4) if with_exchange else None
    rng = random.Random(7)
    early = []
    for _ in range(nsteps):
        out = T.cubegan_training_step(model, batch, opts, reducers=reds, rng=rng)
        assert all(v == v for v in out.values())
        if reds:
            early.append(tuple(r.launched_early for r in reds))
    torch.cuda.synchronize()
    return [p.detach().clone() for p in model.parameters()], early


def test_arena_reducer_hooks_and_side_streams_under_rccl_world1():
    """The exchange that SHIPS: ArenaReducer over the FlatAdamW arenas, reduce_scatters launched from bucket-ready
    hooks inside backward(), sub-graphs on side streams (hifigan/streams.py), backend nccl = RCCL.  At a world of one the averaged
    gradient is the local gradient, so five steps with the exchange must leave exactly the parameters of five steps without it: a
    chunk sent before all of its gradients were written (the stream-ordering race commit 188d005 fixed) shows up as a difference."""
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(_free_port())
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    model, batch, random = _cubegan_setup(777)
    ref, _ = _run_steps(model, batch, random, 5, with_exchange=False)
    model2, batch2, _ = _cubegan_setup(777)
    twice, _ = _run_steps(model2, batch2, random, 5, with_exchange=False)
    deterministic = all(torch.equal(a, b) for a, b in zip(ref, twice))
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
    try:
        model3, batch3, _ = _cubegan_setup(777)
        got, early = _run_steps(model3, batch3, random, 5, with_exchange=True)
    finally:
        dist.destroy_process_group()
    assert early[0] == (0, 0, 0) and early[1] == (0, 0, 0)          # layout step, order-recording step
    assert all(sum(e) > 0 for e in early[2:]), early                # then chunks leave while backward() is still running
    # the step itself is bit-reproducible (since round 4: the phoneme -> frame expansion no longer differentiates through torch.gather's
    # float-atomic scatter), so "the exchange changes nothing" can be asked for bit by bit
    assert deterministic, 'two identical 5-step runs without exchange differ: the training step lost its reproducibility'
    worst = max(float((a - b).abs().max() / (b.abs().max() + 1e-12)) for a, b in zip(got, ref))
    assert all(torch.equal(a, b) for a, b in zip(got, ref)), worst


def _world2_worker(rank, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    dev = torch.device('cuda', rank)
    torch.cuda.set_device(dev)
    dist.init_process_group('nccl', rank=rank, world_size=2, device_id=dev)
    try:
        from ttscube_amd.distributed import broadcast_parameters
        model, batch, random = _cubegan_setup(100 + rank)          # rank-distinct data, replicated parameters
        broadcast_parameters(model)
        params, early = _run_steps(model, batch, random, 4, with_exchange=True)
        chk = torch.stack([p.double().sum() for p in params]).cpu()
        q.put((rank, chk, early))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason='needs two GPUs (the driver\'s 8-GPU node); world 1 is covered above')
def test_arena_reducer_world2_replicas_stay_identical():
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_world2_worker, args=(r, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict()
    for _ in range(2):
        r, chk, early = q.get(timeout=600)
        res[r] = (chk, early)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert torch.equal(res[0][0], res[1][0])          # replicas identical after four exchanged steps
    assert all(sum(e) > 0 for e in res[0][1][2:])
