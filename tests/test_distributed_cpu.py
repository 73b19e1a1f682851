"""CPU, world_size=2, gloo: the flat-bucket gradient exchange used by the data-parallel training step."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q, use_rs):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from ttscube_amd.distributed import FlatBucketReducer, broadcast_parameters
    torch.manual_seed(100 + rank)                       # ranks start different on purpose
    net = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3))
    frozen = torch.nn.Linear(2, 2)                      # a parameter that never receives a grad
    broadcast_parameters(net)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(8, 7, generator=g)
    y = torch.randn(8, 3, generator=g)
    xs, ys = x[rank * 4:(rank + 1) * 4], y[rank * 4:(rank + 1) * 4]   # rank-distinct shard of the global batch
    loss = ((net(xs) - ys) ** 2).mean()
    loss.backward()
    partial = torch.nn.Linear(3, 1)                     # a parameter with a gradient on rank 0 only
    if rank == 0:
        partial(torch.ones(1, 3)).sum().backward()
    red = FlatBucketReducer(list(net.parameters()) + list(frozen.parameters()) + list(partial.parameters()), bucket_mb=0.0001,
                            use_reduce_scatter=use_rs)   # tiny buckets: several of them, none a multiple of the world size
    red.reduce()
    assert all(p.grad is None for p in frozen.parameters())        # no gradient anywhere -> stays None (as single-process)
    assert torch.allclose(partial.weight.grad, torch.full((1, 3), 0.5))   # mean of (ones, zeros)
    assert red.bytes_exchanged > 0
    first = [p.grad.clone() for p in net.parameters()]
    # second step, both zero_grad styles: the averaged gradients ALIAS the bucket (no unpack copy), so a kept .grad (set_to_none
    # =False) accumulates in place and is not packed again, a dropped one is re-packed from the fresh tensor
    for set_to_none in (False, True):
        net.zero_grad(set_to_none=set_to_none)
        ((net(xs) - ys) ** 2).mean().backward()
        red.reduce()
        for p, g0 in zip(net.parameters(), first):
            assert torch.allclose(p.grad, g0, atol=1e-7), 'second reduce (set_to_none=%s) changed the averaged gradient' % set_to_none
    q.put((rank, [p.grad.clone() for p in net.parameters()], [p.detach().clone() for p in net.parameters()]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('use_rs', [True, False])
def test_flat_bucket_allreduce_matches_single_process_gradient(use_rs):
    """use_rs=True is the production exchange (reduce_scatter_tensor into a shard buffer + all_gather_into_tensor), the same
    calls RCCL runs on the GPUs; False is the plain all_reduce."""
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, use_rs)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r, grads, params = q.get(timeout=120)
        res[r] = (grads, params)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # both ranks hold the same parameters (broadcast) and the same averaged gradients
    for a, b in zip(res[0][1], res[1][1]):
        assert torch.equal(a, b)
    for a, b in zip(res[0][0], res[1][0]):
        assert torch.allclose(a, b, atol=1e-7)
    # ... equal to the gradient of the mean loss over the whole global batch computed in one process
    net = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3))
    with torch.no_grad():
        for p, v in zip(net.parameters(), res[0][1]):
            p.copy_(v)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(8, 7, generator=g)
    y = torch.randn(8, 3, generator=g)
    ((net(x) - y) ** 2).mean().backward()
    for p, gr in zip(net.parameters(), res[0][0]):
        assert torch.allclose(p.grad, gr, atol=1e-6)


def _arena_worker(rank, world, port, q, overlap):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from ttscube_amd.distributed import ArenaReducer, broadcast_parameters
    from ttscube_amd.optim import FlatAdamW
    torch.manual_seed(100 + rank)
    net = torch.nn.Sequential(torch.nn.Linear(7, 50), torch.nn.Tanh(), torch.nn.Linear(50, 33), torch.nn.Tanh(), torch.nn.Linear(33, 3))
    frozen = torch.nn.Linear(2, 2)                      # never differentiated: stays out of the arenas, .grad stays None
    broadcast_parameters(net)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(8, 7, generator=g)
    y = torch.randn(8, 3, generator=g)
    xs, ys = x[rank * 4:(rank + 1) * 4], y[rank * 4:(rank + 1) * 4]
    # `partial`: differentiated on rank 0 ONLY, and placed in the MIDDLE of the arena: live everywhere (liveness is decided with a MAX over the
    # ranks), but on rank 1 its chunk never completes during backward().  Ranks that launched chunks "when ready" would pair different chunks
    # in RCCL / gloo; with the agreed issue order the averages stay right.
    partial = torch.nn.Linear(3, 1)
    broadcast_parameters(partial)
    ps = list(net.parameters())
    opt = FlatAdamW(ps[:2] + list(partial.parameters()) + ps[2:] + list(frozen.parameters()), lr=1e-3, betas=(0.8, 0.99))
    red = ArenaReducer(opt, bucket_mb=0.0008, overlap=overlap)   # ~200-element chunks: several per layer, parameters straddle them
    early = []
    grads = None
    for step in range(4):
        opt.zero_grad()
        red.arm()
        loss = ((net(xs) - ys) ** 2).mean()
        if rank == 0:
            loss = loss + partial(torch.ones(1, 3)).sum()
        loss.backward()
        red.reduce()                                      # step 0 lays the arenas out (no hooks yet), step 1 records the issue order, later steps overlap
        early.append(red.launched_early)
        grads = [p.grad.clone() for p in net.parameters()]
        assert torch.allclose(partial.weight.grad, torch.full((1, 3), 0.5)), (rank, step, partial.weight.grad)   # mean of (ones, nothing)
    assert all(p.grad is None for p in frozen.parameters())
    assert all(p.grad.data_ptr() >= opt.g.data_ptr() for p in net.parameters())     # gradients live in the arena, parameters too
    assert all(p.data_ptr() >= opt.p.data_ptr() and p.data_ptr() < opt.p.data_ptr() + opt.p.numel() * 4 for p in net.parameters())
    # a parameter that starts receiving gradients after the layout is an error, not a silent divergence
    frozen(torch.ones(1, 2)).sum().backward()
    try:
        opt._check_no_stragglers()
        late = False
    except Exception:
        late = True
    q.put((rank, grads, [p.detach().clone() for p in net.parameters()], early, late, red.bytes_exchanged))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('overlap', [True, False])
def test_arena_reducer_with_bucket_ready_hooks_matches_single_process_gradient(overlap):
    """the exchange over FlatAdamW's gradient arena with the reduce_scatters launched from post-accumulate-grad hooks
    while backward() is still running gives the gradient of the global batch, identically on both ranks."""
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_arena_worker, args=(r, world, port, q, overlap)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r, grads, params, early, late, nbytes = q.get(timeout=120)
        res[r] = (grads, params, early, late, nbytes)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for a, b in zip(res[0][0], res[1][0]):
        assert torch.equal(a, b)                       # the same averaged gradient, bit for bit, on both ranks
    net = torch.nn.Sequential(torch.nn.Linear(7, 50), torch.nn.Tanh(), torch.nn.Linear(50, 33), torch.nn.Tanh(), torch.nn.Linear(33, 3))
    with torch.no_grad():
        for p, v in zip(net.parameters(), res[0][1]):
            p.copy_(v)
    g = torch.Generator().manual_seed(0)
    x, y = torch.randn(8, 7, generator=g), torch.randn(8, 3, generator=g)
    ((net(x) - y) ** 2).mean().backward()
    for p, gr in zip(net.parameters(), res[0][0]):
        assert torch.allclose(p.grad, gr, atol=1e-6)
    for r in range(world):
        early, late, nbytes = res[r][2], res[r][3], res[r][4]
        assert late and nbytes > 0
        assert early[0] == 0 and early[1] == 0         # first step: layout; second: the issue order is recorded; both exchange after backward
        assert (early[2] > 0 and early[3] > 0) if overlap else (early[2] == 0 and early[3] == 0)


def test_utterance_sharding_is_a_partition():
    from ttscube_amd.api import TTSCube
    items = list(range(13))
    for world in (1, 2, 4, 8):
        parts = [TTSCube.shard(items, r, world) for r in range(world)]
        assert sum(parts, []) == items
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
