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


def _np(ts):
    """tensors -> numpy copies: pickled BY VALUE through the queue's pipe.  (A torch tensor travels as a file descriptor that the parent must fetch
    from the still-living child: a worker that exits first leaves the parent with an EOFError — seen as a flaky failure in round 6.)"""
    return [t.detach().cpu().numpy().copy() for t in ts]


def _th(arrs):
    return [torch.from_numpy(a) for a in arrs]


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
    q.put((rank, _np(p.grad for p in net.parameters()), _np(net.parameters())))
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
        grads, params = _th(grads), _th(params)
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
    q.put((rank, _np(grads), _np(net.parameters()), early, late, red.bytes_exchanged))
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
        grads, params = _th(grads), _th(params)
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


# ---- round 6: the three reducers of a step on ONE communicator, world 4, one rank's hooks delayed ---------------------------
def _three_reducer_worker(rank, world, port, q, slow_rank):
    import time
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from ttscube_amd.distributed import ArenaReducer, broadcast_parameters
    from ttscube_amd.optim import FlatAdamW
    torch.manual_seed(7)
    mk = lambda i, h, o: torch.nn.Sequential(torch.nn.Linear(i, h), torch.nn.Tanh(), torch.nn.Linear(h, h), torch.nn.Tanh(), torch.nn.Linear(h, o))
    nets = [mk(6, 40, 3), mk(5, 24, 2), mk(4, 32, 1)]          # "generator side", "discriminators", "text side"
    # a parameter differentiated on rank 0 only, inside the text-side arena: its chunk completes during backward() on rank 0 and NEVER on the
    # others.  If it left from a hook on rank 0 it would be launched BEFORE the discriminators' collectives there and AFTER them elsewhere.
    partial = torch.nn.Linear(3, 1)
    for m in nets + [partial]:
        broadcast_parameters(m)
    tp = list(nets[2].parameters())
    opts = [FlatAdamW(list(nets[0].parameters()), lr=1e-2, betas=(0.8, 0.99)), FlatAdamW(list(nets[1].parameters()), lr=1e-2, betas=(0.8, 0.99)),
            FlatAdamW(tp[:2] + list(partial.parameters()) + tp[2:], lr=1e-2, betas=(0.8, 0.99))]
    reds = [ArenaReducer(o, bucket_mb=0.0008, overlap=True) for o in opts]
    if rank == slow_rank:       # every gradient hook of this rank is late: its chunks complete long after its peers' (host-side skew)
        for m in nets:
            for p in m.parameters():
                p.register_post_accumulate_grad_hook(lambda _p: time.sleep(0.002))
    g = torch.Generator().manual_seed(100)
    data = [(torch.randn(4 * world, n[0].in_features, generator=g), torch.randn(4 * world, n[-1].out_features, generator=g)) for n in nets]
    sl = slice(4 * rank, 4 * rank + 4)
    def update(o):      # FlatAdamW.step is a HIP kernel (no CPU path): plain SGD over the same arenas stands in for it here
        with torch.no_grad():
            o.p.sub_(0.05 * o.g)

    early = []
    for step in range(5):
        # the order of cubegan_training_step: text backward (armed, exchange deferred), discriminator backward + exchange + update,
        # the deferred text exchange + update, generator backward + exchange + update
        opts[2].zero_grad()
        reds[2].arm()
        lt = ((nets[2](data[2][0][sl]) - data[2][1][sl]) ** 2).mean()
        if rank == 0:
            lt = lt + partial(torch.ones(1, 3)).sum()
        lt.backward()
        opts[1].zero_grad()
        reds[1].arm()
        ((nets[1](data[1][0][sl]) - data[1][1][sl]) ** 2).mean().backward()
        reds[1].reduce()
        update(opts[1])
        reds[2].reduce()
        update(opts[2])
        opts[0].zero_grad()
        reds[0].arm()
        ((nets[0](data[0][0][sl]) - data[0][1][sl]) ** 2).mean().backward()
        reds[0].reduce()
        update(opts[0])
        early.append([r.launched_early for r in reds])
    q.put((rank, _np(p for m in nets + [partial] for p in m.parameters()), early, [r._n_early for r in reds], [len(r._chunks) for r in reds]))
    dist.barrier()
    dist.destroy_process_group()


def test_three_reducers_on_one_communicator_stay_identical_under_skew():
    """World 4, three ArenaReducers armed the way the Cubegan step arms them, rank 2's hooks 2 ms late each, one parameter with a gradient on
    rank 0 only: five optimisation steps later all replicas hold the same bits, equal to the single-process run over the global batch."""
    world = 4
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_three_reducer_worker, args=(r, world, port, q, 2)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r, params, early, n_early, n_chunks = q.get(timeout=180)
        res[r] = (_th(params), early, n_early, n_chunks)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(1, world):
        assert all(torch.equal(a, b) for a, b in zip(res[0][0], res[r][0])), 'replica %d diverged' % r
        assert res[r][1] == res[0][1] or r == 2      # the same chunks left early on every rank that was not slowed down ...
        assert res[r][2] == res[0][2]                # ... and every rank agreed on which chunks MAY leave early
    # the single-process run over the global batch (mean over 16 items = mean over the ranks of the means over 4)
    torch.manual_seed(7)
    mk = lambda i, h, o: torch.nn.Sequential(torch.nn.Linear(i, h), torch.nn.Tanh(), torch.nn.Linear(h, h), torch.nn.Tanh(), torch.nn.Linear(h, o))
    nets = [mk(6, 40, 3), mk(5, 24, 2), mk(4, 32, 1)]
    partial = torch.nn.Linear(3, 1)
    g = torch.Generator().manual_seed(100)
    data = [(torch.randn(4 * world, n[0].in_features, generator=g), torch.randn(4 * world, n[-1].out_features, generator=g)) for n in nets]
    for step in range(5):
        for i in (2, 1, 0):
            ps = list(nets[i].parameters()) + (list(partial.parameters()) if i == 2 else [])
            loss = ((nets[i](data[i][0]) - data[i][1]) ** 2).mean() + (partial(torch.ones(1, 3)).sum() / world if i == 2 else 0.0)
            gs = torch.autograd.grad(loss, ps)
            with torch.no_grad():
                for p_, g_ in zip(ps, gs):
                    p_.sub_(0.05 * g_)
    solo = [p_ for m in nets + [partial] for p_ in m.parameters()]
    assert all(torch.allclose(a, b.detach(), atol=1e-5) for a, b in zip(res[0][0], solo))
    n_early, n_chunks = res[0][2], res[0][3]
    assert n_early[0] == n_chunks[0] and n_early[1] == n_chunks[1]       # full overlap where every rank has every gradient
    assert 0 < n_early[2] < n_chunks[2]                                   # the rank-0-only parameter's chunk(s) are exchanged late, everywhere
    assert res[0][1][0] == [0, 0, 0] and res[0][1][1] == [0, 0, 0] and all(v > 0 for v in res[0][1][4])   # layout, recording, then overlap


def test_rank_shards_give_every_rank_the_same_number_of_steps():
    """1 000 items, 8 ranks, b = 16 (train_cubegan.py's loaders under 8 GPUs): a rank with one batch more than its peers would block forever
    in its extra gradient exchange.  Every item is visited, no rank is idle, shard sizes are equal (wrap-padded)."""
    from ttscube_amd.io_utils.loader import equal_batches, rank_shard
    for n, world, b in ((1000, 8, 16), (1001, 8, 16), (7, 8, 16), (1000, 3, 128), (16, 8, 16)):
        shards = [rank_shard(n, r, world) for r in range(world)]
        assert len({len(s) for s in shards}) == 1
        assert set(i for s in shards for i in s) == set(range(n))
        batches = [equal_batches(s, b) for s in shards]
        assert len({len(bs) for bs in batches}) == 1                        # the same number of optimisation steps everywhere
        assert all([len(x) for x in bs] == [len(x) for x in batches[0]] for bs in batches)   # and the same batch sizes, step by step
        assert sum(len(s) for s in shards) - n < world                      # wrap-padding: fewer than `world` items are visited twice
