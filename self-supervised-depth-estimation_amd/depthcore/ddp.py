"""Data-parallel gradient exchange over RCCL/xGMI (torch.distributed backend "nccl" on ROCm).

One process per GPU, full replica, per-rank batch = opt.batch_size.  Gradients live in
a few flat fp32 buckets (parameters' .grad are views into them); a bucket's all-reduce is launched
asynchronously from the autograd hook of its last-arriving gradient, so the exchange of the pose
branch overlaps the backward of the depth branch.  Buckets are sized for xGMI (point-to-point links:
few large messages rather than many small ones).
"""
import contextlib

import torch
import torch.distributed as dist


class GradBuckets:
    """Single GPU: gradients stay where autograd puts them (no copies at all).
    Multi GPU: when the last gradient of a bucket has been produced, the bucket is packed with ONE
    multi-tensor copy, all-reduced asynchronously (overlapping the rest of backward), and unpacked
    (one multi-tensor copy) in finish()."""

    def __init__(self, named_params, bucket_mb=32, world_size=1, process_group=None):
        self.world = world_size
        self.pg = process_group
        # reverse registration order ~ order in which backward produces gradients
        params = [(n, p) for n, p in named_params if p.requires_grad and ".fc." not in n]  # fc never gets a grad
        self.all_params = [p for _, p in named_params]
        params = params[::-1]
        self.buckets = []
        cur, cur_n = [], 0
        limit = bucket_mb * (1 << 20) // 4
        for n, p in params:
            if cur and cur_n + p.numel() > limit:
                self.buckets.append(cur)
                cur, cur_n = [], 0
            cur.append(p)
            cur_n += p.numel()
        if cur:
            self.buckets.append(cur)
        self.flat, self.views, self.pending, self.handles, self.launched = [], [], [], [], []
        self.streams = []       # streams on which gradients are produced besides the current one (Trainer.overlap_streams)
        self.comm = None        # communication stream (created on first use)
        self.nbytes = sum(p.numel() * 4 for plist in self.buckets for p in plist)
        if self.world == 1:
            return
        for bi, plist in enumerate(self.buckets):
            n = sum(p.numel() for p in plist)
            flat = torch.zeros(n, dtype=plist[0].dtype, device=plist[0].device)
            off, views = 0, []
            for p in plist:
                views.append(flat[off:off + p.numel()].view_as(p))
                off += p.numel()
                p.register_post_accumulate_grad_hook(self._make_hook(bi))
            self.flat.append(flat)
            self.views.append(views)
            self.pending.append(len(plist))
            self.launched.append(False)

    def _launch(self, bi):
        plist = self.buckets[bi]
        flat = self.flat[bi]
        if flat.is_cuda:
            # Pack and exchange on a dedicated communication stream that waits for every stream gradients are produced
            # on (with Trainer.overlap_streams a bucket may hold gradients of both branches; all of them have been
            # *enqueued* by now, since this runs from the hook of the last one).  The producing streams are not blocked.
            if self.comm is None:
                self.comm = torch.cuda.Stream(flat.device)
            cur = torch.cuda.current_stream(flat.device)
            self.comm.wait_stream(cur)
            for st in self.streams:
                self.comm.wait_stream(st)
            ctx = torch.cuda.stream(self.comm)
        else:
            ctx = contextlib.nullcontext()
        with ctx:
            have = [i for i, p in enumerate(plist) if p.grad is not None]
            if len(have) != len(plist):
                flat.zero_()
            if have:
                torch._foreach_copy_([self.views[bi][i] for i in have], [plist[i].grad for i in have])
                if flat.is_cuda:
                    for i in have:
                        plist[i].grad.record_stream(self.comm)
            self.handles.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.pg, async_op=True))
        self.launched[bi] = True

    def _make_hook(self, bi):
        def hook(param):
            self.pending[bi] -= 1
            if self.pending[bi] == 0:
                self._launch(bi)
        return hook

    def zero(self):
        """model_optimizer.zero_grad(set_to_none=True) + reset of the bucket state."""
        for p in self.all_params:
            p.grad = None
        for bi in range(len(self.flat)):
            self.pending[bi] = len(self.buckets[bi])
            self.launched[bi] = False
        self.handles = []

    def finish(self):
        """Wait for the exchanges and turn sums into means (call before optimizer.step)."""
        if self.world == 1:
            return
        for bi in range(len(self.flat)):      # buckets with a parameter that got no gradient this step
            if not self.launched[bi]:
                self._launch(bi)
        for h in self.handles:
            h.wait()
        if self.comm is not None:
            torch.cuda.current_stream(self.flat[0].device).wait_stream(self.comm)
        inv = 1.0 / self.world
        for bi, plist in enumerate(self.buckets):
            self.flat[bi].mul_(inv)
            for p, v in zip(plist, self.views[bi]):
                p.grad = v


def broadcast_parameters(modules, src=0, process_group=None):
    for m in modules:
        for t in list(m.parameters()) + list(m.buffers()):
            dist.broadcast(t.data, src=src, group=process_group)
