"""The N>1 gradient exchange (depthcore/ddp.py) on CPU: 2 gloo ranks must end with the mean of
the per-rank gradients, including a parameter that receives no gradient on one step."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _Net(nn.Module):
    def __init__(self):
        super().__init__()
        self.c1 = nn.Conv2d(3, 8, 3, padding=1)
        self.c2 = nn.Conv2d(8, 4, 3, padding=1)
        self.fc = nn.Linear(4, 2)          # never used, like encoder.fc

    def forward(self, x):
        return self.c2(torch.relu(self.c1(x)))


def _make_model():
    torch.manual_seed(0)
    return _Net()


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from depthcore.ddp import GradBuckets, broadcast_parameters
    m = _make_model()
    with torch.no_grad():
        for p in m.parameters():
            p.add_(rank)                 # make replicas differ, then broadcast must undo it
    broadcast_parameters([m], 0)
    gb = GradBuckets([("m." + n, p) for n, p in m.named_parameters()], bucket_mb=0.0001, world_size=world)
    assert len(gb.buckets) > 1
    for step in range(2):
        torch.manual_seed(100 + rank + 10 * step)
        x = torch.randn(2, 3, 8, 8)
        gb.zero()
        m(x).square().mean().backward()
        gb.finish()
    q.put((rank, [p.grad.numpy().copy() for n, p in m.named_parameters() if not n.startswith("fc.")]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_mean():
    import sys
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # single-process reference: mean of the two ranks' step-1 gradients
    m = _make_model()
    want = None
    for rank in range(world):
        torch.manual_seed(100 + rank + 10)
        x = torch.randn(2, 3, 8, 8)
        m.zero_grad()
        m(x).square().mean().backward()
        g = [p.grad.clone() for n, p in m.named_parameters() if not n.startswith("fc.")]
        want = g if want is None else [a + b for a, b in zip(want, g)]
    want = [w / world for w in want]
    for rank in range(world):
        for a, b in zip(res[rank], want):
            assert torch.allclose(torch.from_numpy(a), b, rtol=1e-5, atol=1e-7)
