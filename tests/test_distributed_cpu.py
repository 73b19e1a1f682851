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


def test_utterance_sharding_is_a_partition():
    from ttscube_amd.api import TTSCube
    items = list(range(13))
    for world in (1, 2, 4, 8):
        parts = [TTSCube.shard(items, r, world) for r in range(world)]
        assert sum(parts, []) == items
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
