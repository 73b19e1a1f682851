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
