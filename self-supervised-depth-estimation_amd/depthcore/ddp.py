"""Data-parallel gradient exchange over RCCL/xGMI (torch.distributed backend "nccl" on ROCm).

One process per GPU, full replica, per-rank batch = opt.batch_size.  Gradients live in
a few flat fp32 buckets (parameters' .grad are views into them); a bucket's all-reduce is launched
asynchronously from the autograd hook of its last-arriving gradient, so the exchange of the pose
branch overlaps the backward of the depth branch.  Buckets are sized for xGMI (point-to-point links:
few large messages rather than many small ones).
"""
import torch
import torch.distributed as dist


class GradBuckets:
    def __init__(self, named_params, bucket_mb=32, world_size=1, process_group=None):
        self.world = world_size
        self.pg = process_group
        # reverse registration order ~ order in which backward produces gradients
        params = [(n, p) for n, p in named_params if p.requires_grad and ".fc." not in n]  # fc never gets a grad
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
        self.flat, self.pending, self.handles = [], [], []
        for bi, plist in enumerate(self.buckets):
            n = sum(p.numel() for p in plist)
            flat = torch.zeros(n, dtype=plist[0].dtype, device=plist[0].device)
            off = 0
            for p in plist:
                p.grad = flat[off:off + p.numel()].view_as(p)
                off += p.numel()
                if self.world > 1:
                    p.register_post_accumulate_grad_hook(self._make_hook(bi))
            self.flat.append(flat)
            self.pending.append(len(plist))
        self.nbytes = sum(f.numel() * 4 for f in self.flat)

    def _make_hook(self, bi):
        def hook(param):
            self.pending[bi] -= 1
            if self.pending[bi] == 0:
                self.handles.append(dist.all_reduce(self.flat[bi], op=dist.ReduceOp.SUM, group=self.pg, async_op=True))
        return hook

    def zero(self):
        for bi, f in enumerate(self.flat):
            f.zero_()
            self.pending[bi] = len(self.buckets[bi])
        self.handles = []

    def finish(self):
        """Wait for the exchanges and turn sums into means (call before optimizer.step)."""
        if self.world == 1:
            return
        for bi, n in enumerate(self.pending):   # buckets with a parameter that got no gradient this step
            if n != 0:
                self.handles.append(dist.all_reduce(self.flat[bi], op=dist.ReduceOp.SUM, group=self.pg, async_op=True))
        for h in self.handles:
            h.wait()
        inv = 1.0 / self.world
        for f in self.flat:
            f.mul_(inv)


def broadcast_parameters(modules, src=0, process_group=None):
    for m in modules:
        for t in list(m.parameters()) + list(m.buffers()):
            dist.broadcast(t.data, src=src, group=process_group)
