"""The N>1 gradient exchange (depthcore/ddp.py) on CPU: 2 gloo ranks must end with the mean of
the per-rank gradients -- including a (non-fc) parameter that receives no gradient on ONE rank, gradients written
straight into their bucket slice by an op that uses ops._grad_dst (no pack copy), and the same sequence of collectives
on every rank."""
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


class _ScaleFn(torch.autograd.Function):
    """y = x * w (w: per-channel) whose backward writes dw where the depthcore ops do: ops._grad_dst."""

    @staticmethod
    def forward(ctx, x, w):
        from depthcore import ops
        ctx.save_for_backward(x, w)
        ctx.slot = ops._slot(w)
        return x * w.view(1, -1, 1, 1)

    @staticmethod
    def backward(ctx, g):
        from depthcore import ops
        x, w = ctx.saved_tensors
        dw = ops._grad_dst(ctx.slot, w)
        torch.sum(g * x, dim=(0, 2, 3), out=dw)
        return g * w.view(1, -1, 1, 1), dw


class _Net(nn.Module):
    def __init__(self):
        super().__init__()
        self.c1 = nn.Conv2d(3, 8, 3, padding=1)
        self.scale = nn.Parameter(torch.ones(8))
        self.c2 = nn.Conv2d(8, 4, 3, padding=1)
        self.aux = nn.Conv2d(8, 4, 1)      # used on rank 0 only: no gradient on the other rank
        self.fc = nn.Linear(4, 2)          # never used, like encoder.fc

    def forward(self, x, use_aux=True):
        h = _ScaleFn.apply(torch.relu(self.c1(x)), self.scale)
        y = self.c2(h)
        return y + self.aux(h) if use_aux else y


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
        m(x, use_aux=(rank == 0)).square().mean().backward()
        gb.finish()
        # the op that writes into its slice was not packed; every other gradient (stock torch ops) was
        nb = sum(len(b) for b in gb.buckets)
        assert gb.packed == nb - 1 - (2 if rank else 0), (gb.packed, nb)
        assert m.scale.grad.data_ptr() == m.scale._dc_grad_slot.view.data_ptr()
    assert all(p.grad.data_ptr() == p._dc_grad_slot.view.data_ptr() for b in gb.buckets for p in b)
    assert m.fc.weight.grad is None
    grads_step1 = [p.grad.numpy().copy() for n, p in m.named_parameters() if not n.startswith("fc.")]
    # a parameter frozen AFTER the buckets were built (Trainer.freeze_hidden_states at world > 1): its slice still travels
    # (zeros, same collective sequence on every rank) but `.grad` must stay None, or Adam would keep moving it
    m.scale.requires_grad = False
    m.scale.grad = None
    gb.zero()
    torch.manual_seed(300 + rank)
    m(torch.randn(2, 3, 8, 8), use_aux=(rank == 0)).square().mean().backward()
    gb.finish()
    assert m.scale.grad is None
    assert m.c1.weight.grad is not None and m.c1.weight.grad.data_ptr() == m.c1.weight._dc_grad_slot.view.data_ptr()
    assert float(m.scale._dc_grad_slot.view.abs().max()) == 0.0
    q.put((rank, grads_step1, list(gb.launch_order)))
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
    got = [q.get(timeout=120) for _ in range(world)]
    res = {r: g for r, g, _ in got}
    orders = [o for _, _, o in got]
    assert orders[0] == orders[1] == sorted(orders[0]) and len(orders[0]) > 1      # same collective sequence everywhere
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
        m(x, use_aux=(rank == 0)).square().mean().backward()
        g = [(p.grad.clone() if p.grad is not None else torch.zeros_like(p))       # no gradient = zeros in the mean
             for n, p in m.named_parameters() if not n.startswith("fc.")]
        want = g if want is None else [a + b for a, b in zip(want, g)]
    want = [w / world for w in want]
    for rank in range(world):
        for a, b in zip(res[rank], want):
            assert torch.allclose(torch.from_numpy(a), b, rtol=1e-5, atol=1e-7)


def _agree_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from depthcore.ddp import agree_all
    res = [agree_all(True), agree_all(rank != 1), agree_all(rank == 1), agree_all(False)]
    sub = dist.new_group(ranks=[0, 1])
    res.append(agree_all(rank == 0, sub))
    q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


def test_agree_all_is_the_minimum_over_the_ranks():
    """The collective yes/no behind Trainer.train_step's capture decisions: one rank's "no" is everybody's."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_agree_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert got[0] == got[1] == [True, False, False, False, False]


def test_agree_all_without_a_process_group_is_the_local_answer():
    from depthcore.ddp import agree_all
    assert agree_all(True) is True and agree_all(False) is False
