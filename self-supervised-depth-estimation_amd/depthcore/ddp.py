"""Data-parallel gradient exchange over RCCL/xGMI (torch.distributed backend "nccl" on ROCm).

One process per GPU, full replica, per-rank batch = opt.batch_size (the reference itself has no data
parallelism).  Gradients live in a few flat fp32 buckets sized for xGMI (point-to-point links: few large messages
rather than many small ones):

  * every parameter owns a slice of a bucket (`GradSlot`); the depthcore autograd ops write a parameter gradient
    straight INTO that slice (ops._grad_dst), autograd adopts the slice as `.grad` without a copy, so for the kernels of
    this package there is no pack pass and no unpack pass.  Gradients produced by stock torch ops (the few library
    convolutions left) are packed with one multi-tensor copy per bucket;
  * a bucket's all-reduce is launched asynchronously from the autograd hook of its last-arriving gradient (in bucket
    order, so every rank issues the same sequence of collectives), on a dedicated communication stream, so the
    exchange of the branch that finishes first overlaps the backward of the other;
  * the mean is taken by the collective itself (ReduceOp.AVG on RCCL; gloo, used only for CPU rehearsal, has no AVG and
    gets SUM + one scale pass).

Semantics shared with torch's DistributedDataParallel: a parameter that received no gradient on this rank contributes
zeros and ends the step with the mean of the other ranks' gradients (possibly all zeros) as `.grad` -- at world size 1
its `.grad` stays None.  `encoder.fc.*` never gets a gradient on any rank (reference networks/resnet_encoder.py:82 keeps
it for checkpoint compatibility) and is excluded from the buckets.  BatchNorm running statistics are per rank, as in
the reference (no SyncBN); checkpoints are written by rank 0.
"""
import contextlib

import torch
import torch.distributed as dist


class GradSlot:
    """A parameter's slice of its flat bucket.  `armed` is set by GradBuckets.zero() and cleared by the first backward
    op that takes the slice as its output, so a parameter used twice in one graph gets an ordinary second gradient
    that autograd accumulates into the slice."""
    __slots__ = ("view", "armed")

    def __init__(self, view):
        self.view = view
        self.armed = False

    def take(self):
        self.armed = False
        return self.view.detach()       # a fresh tensor object on the same memory: autograd may adopt it as .grad


class GradBuckets:
    """Single GPU: gradients stay where autograd puts them (no buckets, no copies).
    Multi GPU: see the module docstring."""

    def __init__(self, named_params, bucket_mb=32, world_size=1, process_group=None, tail_mb=4):
        self.world = world_size
        self.pg = process_group
        named_params = list(named_params)
        # reverse of the order given ~ order in which backward produces gradients: pass the modules in the order the
        # forward runs them (Trainer does), so that the in-order exchange below starts as early as possible
        params = [(n, p) for n, p in named_params if p.requires_grad and ".fc." not in n]  # fc never gets a grad
        self.all_params = [p for _, p in named_params]
        params = params[::-1]
        self.buckets, self.names = [], []
        cur, cur_names, cur_n = [], [], 0
        limit = bucket_mb * (1 << 20) // 4
        for n, p in params:
            if cur and cur_n + p.numel() > limit:
                self.buckets.append(cur)
                self.names.append(cur_names)
                cur, cur_names, cur_n = [], [], 0
            cur.append(p)
            cur_names.append(n)
            cur_n += p.numel()
        if cur:
            self.buckets.append(cur)
            self.names.append(cur_names)
        # The LAST bucket's exchange cannot hide behind any backward work (its last gradient is the step's last): keep it small.
        # Its trailing parameters -- the first layers of the network that finishes last, a few MB of stem / layer1 / layer2
        # weights whose backward passes are the longest of the step -- become a bucket of their own; what stays in front of
        # them is exchanged while those layers still run.  (resnet18 pairs encoder: 24 MB -> 21 MB + 2.7 MB.)
        tail_limit = int(tail_mb * (1 << 20) // 4)
        if self.buckets and tail_limit > 0 and len(self.buckets[-1]) > 1:
            last, last_names = self.buckets[-1], self.names[-1]
            n_tail, k = 0, len(last)
            while k > 1 and n_tail + last[k - 1].numel() <= tail_limit:
                k -= 1
                n_tail += last[k].numel()
            if k < len(last):
                self.buckets[-1], self.names[-1] = last[:k], last_names[:k]
                self.buckets.append(last[k:])
                self.names.append(last_names[k:])
        self.flat, self.views, self.pending, self.handles, self.launched = [], [], [], [], []
        self.launch_order = []  # bucket indices in the order their collectives were issued this step (identical on all ranks)
        self.next = 0           # next bucket to exchange
        self.packed = 0         # gradients that had to be copied into their slice this step (0 on the all-depthcore path)
        self.streams = []       # streams on which gradients are produced besides the current one (Trainer.overlap_streams)
        self.comm = None        # communication stream (created on first use)
        self.timing = False     # diagnostics (bench.py, after its timed region): event pairs around every bucket's exchange
        self._t0 = None         # ... and one at zero(), the start of the backward they are measured from
        self._tev = []
        self.nbytes = sum((p.numel() + 3) // 4 * 16 for plist in self.buckets for p in plist)    # bytes exchanged per step
        if self.world == 1:
            return
        backend = dist.get_backend(process_group)
        self.avg_op = dist.ReduceOp.AVG if backend == "nccl" else None
        for bi, plist in enumerate(self.buckets):
            # every slice starts on a 16-byte boundary (the kernels write gradients into it with 16-byte vector stores;
            # 1-element dispconv biases and 3-element rel_h / rel_w would otherwise leave their successors misaligned);
            # the padding stays zero and is exchanged along with the rest
            offs, n = [], 0
            for p in plist:
                offs.append(n)
                n += (p.numel() + 3) // 4 * 4
            flat = torch.zeros(n, dtype=plist[0].dtype, device=plist[0].device)
            views = []
            for p, off in zip(plist, offs):
                v = flat[off:off + p.numel()].view_as(p)
                views.append(v)
                p._dc_grad_slot = GradSlot(v)
                p.register_post_accumulate_grad_hook(self._make_hook(bi))
            self.flat.append(flat)
            self.views.append(views)
            self.pending.append(len(plist))
            self.launched.append(False)

    def _launch(self, bi):
        plist = self.buckets[bi]
        flat = self.flat[bi]
        if flat.is_cuda:
            # Exchange on a dedicated communication stream that waits for every stream gradients are produced
            # on (with Trainer.overlap_streams a bucket may hold gradients of both branches; all of them have been
            # *enqueued* by now, since this runs from the hook of the last one).  The producing streams are not blocked.
            if self.comm is None:
                self.comm = torch.cuda.Stream(flat.device)
            cur = torch.cuda.current_stream(flat.device)
            self.comm.wait_stream(cur)
            for st in self.streams:
                self.comm.wait_stream(st)
            # weight gradients produced on the companion streams of ops.WgradLanes (their kernels write straight into this
            # bucket's slices): every lane with work enqueued so far -- this runs from the hook of the bucket's LAST gradient,
            # whose lane was registered when its backward returned
            from .ops import WgradLanes
            for lane in WgradLanes._used:
                self.comm.wait_stream(lane)
            ctx = torch.cuda.stream(self.comm)
        else:
            ctx = contextlib.nullcontext()
        with ctx:
            views = self.views[bi]
            missing = [i for i, p in enumerate(plist) if p.grad is None]
            if missing:          # no gradient on this rank this step: contributes zeros (module docstring)
                torch._foreach_zero_([views[i] for i in missing])
            stray = [i for i, p in enumerate(plist)
                     if p.grad is not None and p.grad.data_ptr() != views[i].data_ptr()]
            if stray:            # produced by a stock torch op (or not adopted by autograd): pack
                torch._foreach_copy_([views[i] for i in stray], [plist[i].grad for i in stray])
                if flat.is_cuda:
                    for i in stray:
                        plist[i].grad.record_stream(self.comm)
                self.packed += len(stray)
            op = self.avg_op if self.avg_op is not None else dist.ReduceOp.SUM
            ev = None
            if self.timing and flat.is_cuda:
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record(self.comm)
            self.handles.append(dist.all_reduce(flat, op=op, group=self.pg, async_op=True))
            if ev is not None:
                # (RCCL runs the collective on its own internal stream; wait() makes the communication stream -- which nothing but
                # the exchange uses -- wait for it without blocking the host, so the end event brackets the collective; gloo's handle
                # completes on the host)
                self.handles[-1].wait()
                ev[1].record(self.comm)
                self._tev.append((bi, ev))
        self.launched[bi] = True
        self.launch_order.append(bi)

    def _make_hook(self, bi):
        def hook(param):
            self.pending[bi] -= 1
            # Collectives are issued strictly in bucket order, whatever order the gradients arrive in: every rank then
            # issues the same sequence even if a parameter gets no gradient on one of them (finish() issues the rest).
            while self.next < len(self.flat) and self.pending[self.next] == 0:
                self._launch(self.next)
                self.next += 1
        return hook

    def zero(self):
        """model_optimizer.zero_grad(set_to_none=True) + reset of the bucket state (arms the gradient slots)."""
        for p in self.all_params:
            p.grad = None
        for bi in range(len(self.flat)):
            self.pending[bi] = len(self.buckets[bi])
            self.launched[bi] = False
            for p in self.buckets[bi]:
                p._dc_grad_slot.armed = True
        self.handles = []
        self.launch_order = []
        self.packed = 0
        self.next = 0
        if self.timing and self.flat and self.flat[0].is_cuda:
            self._t0 = torch.cuda.Event(enable_timing=True)
            self._t0.record(torch.cuda.current_stream(self.flat[0].device))
            self._tev = []

    def finish(self):
        """Wait for the exchanges; afterwards every bucketed parameter's `.grad` is its (averaged) slice.
        Call between backward() and optimizer.step()."""
        if self.world == 1:
            return
        while self.next < len(self.flat):     # buckets holding a parameter that got no gradient this step
            self._launch(self.next)
            self.next += 1
        for h in self.handles:
            h.wait()
        if self.comm is not None:
            torch.cuda.current_stream(self.flat[0].device).wait_stream(self.comm)
        if self.avg_op is None:
            torch._foreach_mul_(self.flat, 1.0 / self.world)
        for bi, plist in enumerate(self.buckets):
            for p, v in zip(plist, self.views[bi]):
                if not p.requires_grad:
                    # frozen after construction (Trainer.freeze_hidden_states): its slice is exchanged as zeros on every
                    # rank, but `.grad` stays None so that Adam skips it -- step count and moments untouched, exactly as
                    # at world size 1 (a zero gradient would still move it along its decaying first moment)
                    p.grad = None
                    continue
                if p.grad is None or p.grad.data_ptr() != v.data_ptr():
                    p.grad = v


def broadcast_parameters(modules, src=0, process_group=None):
    """Initial weights AND buffers from rank `src` (afterwards BatchNorm statistics evolve per rank)."""
    for m in modules:
        for t in list(m.parameters()) + list(m.buffers()):
            dist.broadcast(t.data, src=src, group=process_group)


def agree_all(ok, process_group=None, device=None):
    """A per-rank yes/no made collective: True only if EVERY rank of `process_group` said yes (all-reduce MIN of a flag).
    Decisions that choose which communicator the next collectives run on (graph replay on the capture group vs eager steps on
    the base group, Trainer.train_step) must be taken through this, or one rank's fallback leaves the others waiting on a
    communicator it never enters.  Every rank must call it at the same point of the same step.  The flag lives where the group's
    backend reduces (device memory for RCCL, host memory for gloo)."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(process_group) == 1:
        return bool(ok)
    on_host = dist.get_backend(process_group) == "gloo" or device is None
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device="cpu" if on_host else device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=process_group)
    return bool(int(flag.item()))


def bucket_timeline(gb):
    """After a step run with `gb.timing = True` (and a device synchronisation): per bucket in launch order, the GPU time from
    the start of the backward (zero()) at which its exchange could start -- the communication stream has waited for every
    producing stream -- and its duration: [(bucket, bytes, start_ms, ms)]."""
    if gb._t0 is None:
        return []
    return [(bi, gb.flat[bi].numel() * 4, round(gb._t0.elapsed_time(e0), 3), round(e0.elapsed_time(e1), 3)) for bi, (e0, e1) in gb._tev]
