"""torch.autograd bindings of the C ABI in include/depthcore.h.

Every op enqueues hand-written gfx950 kernels on the current HIP stream; tensors
are allocated by PyTorch's caching allocator and passed down as raw pointers.
"""
import collections
import contextlib
import ctypes
import os
import threading
import weakref

import torch

from . import _lib
from ._lib import PhotoDesc, check, ptr, stream


DepthcoreError = _lib.DepthcoreError


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


# ---- decision observers -------------------------------------------------------------------------------------------------
# Instrumentation hook (no test logic lives here): a callable registered with `add_decision_observer` is told every discrete
# decision the forward takes -- `("relu", y)` with the fused op's output (the decision is y > 0), `("maxpool", code)` with the
# argmax codes -- in call order.  ReLUs that are folded into a convolution's loader never materialise their output: they hand
# over a thunk, which is only evaluated while an observer is registered.  The parity tests build their tape on this
# (tests/kink_tape.py); nothing in the training path registers one.
_decision_observers = []


def add_decision_observer(fn):
    _decision_observers.append(fn)


def remove_decision_observer(fn):
    _decision_observers.remove(fn)


# ---- matrix-core precision of the convolutions (include/depthcore.h: dc_set_matrix_precision) ----------------------------
# "f32": exact fp32 MFMA (the reference's arithmetic, the default).  "bf16": the reduced-precision-networks policy of BASELINE
# configs[4] -- convolution operands rounded to bf16 on their way into LDS, fp32 accumulation, everything else (tensors in HBM,
# master weights, BatchNorm statistics, the photometric loss) fp32.  The C library keeps the setting per calling thread; a
# convolution records the precision of its forward and its backward (run by autograd's thread) uses the same.
PRECISIONS = {"f32": _lib.PREC_F32, "bf16": _lib.PREC_BF16}
_precision = [_lib.PREC_F32]
_tls = threading.local()


class matrix_precision:
    """with ops.matrix_precision("bf16"): ...  -- convolutions issued inside use the bf16 matrix cores."""

    def __init__(self, name):
        if name not in PRECISIONS:
            raise _lib.DepthcoreError("matrix precision must be one of %s, got %r" % (sorted(PRECISIONS), name))
        self.prec = PRECISIONS[name]

    def __enter__(self):
        self.prev, _precision[0] = _precision[0], self.prec
        return self

    def __exit__(self, *exc):
        _precision[0] = self.prev
        return False


def _use_precision(prec):
    """Make `prec` the calling thread's library setting (one C call, only when it changes)."""
    if getattr(_tls, "prec", _lib.PREC_F32) != prec:
        rc = _lib.lib().dc_set_matrix_precision(prec)
        if rc < 0:
            check(rc, "dc_set_matrix_precision")
        _tls.prec = prec
    return prec


def _record_kink(kind, t):
    """`t`: the tensor that carries the decision, or a thunk that forms it (folded ReLUs never materialise theirs)."""
    if _decision_observers:
        t = t() if callable(t) else t
        for fn in _decision_observers:
            fn(kind, t)


def _slot(param):
    """forward(): the parameter's slice of its flat DDP gradient bucket, if depthcore.ddp.GradBuckets gave it one."""
    return getattr(param, "_dc_grad_slot", None)


def _grad_dst(slot, like):
    """backward(): where a parameter gradient is written -- the armed bucket slice (autograd adopts it as `.grad`, so the
    gradient exchange needs no pack copy) or, at world size 1 / on a second use of the parameter, a fresh tensor."""
    if slot is not None and slot.armed:
        return slot.take()
    return torch.empty_like(like) if like is not None else None


# ---- weight gradients off the backward's critical path ----------------------------------------------------------------------
class WgradLanes:
    """Only the data gradients feed the next backward op; a weight gradient is not read before the optimiser.  While active
    (`with WgradLanes.active(): loss.backward()`), the convolutions launch their weight-gradient kernels on a companion
    HIP stream of the stream their backward runs on ("lane"): the kernels of the data-gradient chain and the weight-gradient
    kernels then share the chip, one side's blocks filling the other's partial rounds and prologues, instead of queueing
    behind each other.  Leaving the context joins every lane into the current stream (before the optimiser / the gradient
    exchange read the gradients).

    A parameter used MORE than once in the graph keeps its weight gradients on the backward's own stream: autograd sums
    the uses' gradients right away, on a stream that knows nothing about the lane.  Uses are counted by the forwards."""
    _on = False
    _lanes = {}          # (device index, raw handle of the backward's stream) -> companion stream
    _uses = {}           # id(parameter) -> forward uses since the last join
    _used = {}           # lane with work enqueued since the last join -> the stream whose backward it accompanies
    _held = collections.deque()     # (event recorded behind a lane's kernels, the tensors they read): kept alive until it completes
    _record_stream = os.environ.get("DC_LANES_RECORD_STREAM", "0") == "1"

    @classmethod
    @contextlib.contextmanager
    def active(cls, on=True):
        prev, cls._on = cls._on, bool(on)
        try:
            yield
        finally:
            cls._on = prev
            cls.join()

    @classmethod
    def join(cls):
        for lane, parent in cls._used.items():
            torch.cuda.current_stream(lane.device).wait_stream(lane)
            # the tensors a lane reads (held below) go back to the pool of the stream that OWNS them -- the backward's stream,
            # e.g. the pose side stream -- so that stream waits for the lane too before its blocks can be handed out again
            parent.wait_stream(lane)
        cls._used = {}
        cls._uses = {}
        cls._held.clear()        # (whatever reuses this memory is enqueued behind the waits above)

    @classmethod
    def count_use(cls, param):
        cls._uses[id(param)] = cls._uses.get(id(param), 0) + 1

    @classmethod
    @contextlib.contextmanager
    def lane(cls, param, *reads):
        """Inside: the current stream is the lane (or unchanged when the lane cannot be used).  `reads`: tensors of the
        backward's stream that the weight-gradient kernel reads."""
        if not (cls._on and param is not None and reads[0].is_cuda and cls._uses.get(id(param), 0) == 1 and param.grad is None):
            yield
            return
        dev = reads[0].device
        cur = torch.cuda.current_stream(dev)
        key = (dev.index, cur.cuda_stream)
        lane = cls._lanes.get(key)
        if lane is None:
            lane = cls._lanes[key] = torch.cuda.Stream(dev)       # (normal priority, below the step stream's: Trainer.on_step_stream)
        lane.wait_stream(cur)
        with torch.cuda.stream(lane):
            yield
        # The tensors the lane reads belong to the backward's stream and autograd frees them as soon as this backward returns.
        # They are kept alive HERE until the lane's kernels have finished (an event behind them, polled on the next calls),
        # so that their memory goes back to the allocator only when it is really free.  (`record_stream` gives the same
        # safety by deferring the REUSE, but the allocator then grows a whole pool of not-yet-reusable blocks: 16 GB reserved
        # for 3.5 GB allocated at C2, 84 GB for 16.6 GB at C3 in tools/soak.py.)
        if cls._record_stream:               # (A/B of the first version: DC_LANES_RECORD_STREAM=1)
            for t in reads:
                t.record_stream(lane)
        else:
            ev = torch.cuda.Event()
            ev.record(lane)
            cls._held.append((ev, reads))
            while cls._held and cls._held[0][0].query():
                cls._held.popleft()
        cls._used[lane] = cur


# ----------------------------------------------------------------------------------------------
# fused photometric loss  (reference trainer.py:465-622)
# ----------------------------------------------------------------------------------------------
class PhotoConfig:
    """Non-differentiable inputs + options of one fused photometric step.

    target/src: inputs[("color", 0|-1|+1, 0)];  color_s[s]: inputs[("color", 0, s)];
    K / inv_K: inputs[("K"|"inv_K", 0)];  noise[s]: the tie-break randn of trainer.py:594-595
    (None -> on-device counter RNG).  `materialize` asks for the log tensors of
    generate_images_pred (depth / sample / color / identity_selection).
    packed: optional (target, src_m1, src_p1) as pixel-interleaved RGBx (B,H,W,4) tensors -- `pack_rgbx(x)` or the data step's
    ("color_packed", f, 0) -- the layout the kernels gather from; without it every forward repacks the three frames.
    """

    def __init__(self, target, src_m1, src_p1, color_s, K, inv_K, noise=None, min_depth=0.1, max_depth=100.0,
                 smoothness=1e-3, disable_automasking=False, avg_reprojection=False, no_ssim=False,
                 align_corners=False, materialize=False, rng_seed=0, packed=None):
        self.target, self.src = _c(target), (_c(src_m1), _c(src_p1))
        self.packed = None if packed is None else tuple(_c(t) for t in packed)
        self.color_s = [_c(c) for c in color_s]
        self.K, self.inv_K = _c(K), _c(inv_K)
        self.noise = None if noise is None else [_c(n) for n in noise]
        self.min_depth, self.max_depth, self.smoothness = float(min_depth), float(max_depth), float(smoothness)
        self.flags = ((_lib.OPT_NO_AUTOMASK if disable_automasking else 0)
                      | (_lib.OPT_AVG_REPROJ if avg_reprojection else 0)
                      | (_lib.OPT_NO_SSIM if no_ssim else 0)
                      | (_lib.OPT_ALIGN_CORNERS if align_corners else 0))
        self.materialize = materialize
        # an int, or a one-element int64 device tensor read by the kernel at run time (hipGraph replays)
        self.rng_seed_dev = rng_seed if torch.is_tensor(rng_seed) else None
        self.rng_seed = 0 if torch.is_tensor(rng_seed) else int(rng_seed)
        self.extras = {}          # filled by forward: argmin maps + optional log tensors
        self.n_masks = 0          # set by photometric_loss(pred_masks=...)
        self.n_Ts = 0             # set by photometric_loss(T_scales=...): 2 * num_scales per-scale poses, else 0


def _fill_desc(cfg, T0, T1, disps, no_grad=False, masks=(), Ts=(), mode=0):
    B, _, H, W = cfg.target.shape
    ns = len(disps)
    if ns < 1 or ns > _lib.MAX_SCALES:
        raise _lib.DepthcoreError("1..4 scales supported, got %d" % ns)
    if cfg.target.shape[1] != 3:
        raise _lib.DepthcoreError("images must be (B,3,H,W)")
    d = PhotoDesc()
    d.B, d.H, d.W, d.num_scales = B, H, W, ns
    d.flags = cfg.flags | (_lib.OPT_NO_GRAD if no_grad else 0) | (_lib.OPT_PRED_MASK if masks else 0) | mode
    if masks:
        if not (cfg.flags & _lib.OPT_NO_AUTOMASK):
            raise _lib.DepthcoreError("predictive masks need disable_automasking (reference trainer.py:116-117)")
        if len(masks) != ns or any(tuple(m.shape) != (B, 2, H, W) for m in masks):
            raise _lib.DepthcoreError("predictive masks: one (B,2,H,W) full-resolution tensor per scale")
        for s in range(ns):
            d.pred_mask[s] = ptr(masks[s])
    d.min_depth, d.max_depth, d.smoothness = cfg.min_depth, cfg.max_depth, cfg.smoothness
    d.target = ptr(cfg.target)
    for f in range(2):
        if cfg.src[f].shape != cfg.target.shape:
            raise _lib.DepthcoreError("source / target shape mismatch")
        d.source[f] = ptr(cfg.src[f])
    if Ts and len(Ts) != 2 * ns:
        raise _lib.DepthcoreError("per-scale poses: one (T_-1, T_+1) pair per scale")
    for t in (cfg.K, cfg.inv_K, *((T0, T1) if not Ts else Ts)):
        if tuple(t.shape) != (B, 4, 4):
            raise _lib.DepthcoreError("K / inv_K / T must be (B,4,4), got %s" % (tuple(t.shape),))
    d.K, d.inv_K = ptr(cfg.K), ptr(cfg.inv_K)
    if Ts:                                    # posecnn: one pose per (scale, frame), trainer.py:490-499
        for s in range(ns):
            d.T_scale[s][0], d.T_scale[s][1] = ptr(Ts[2 * s]), ptr(Ts[2 * s + 1])
    else:
        d.T[0], d.T[1] = ptr(T0), ptr(T1)
    if cfg.packed is not None:
        if len(cfg.packed) != 3 or any(tuple(t.shape) != (B, H, W, 4) for t in cfg.packed):
            raise _lib.DepthcoreError("packed must be three (B,H,W,4) RGBx tensors (target, source -1, source +1)")
        for k in range(3):
            d.packed[k] = ptr(cfg.packed[k])
    nch = 1 if (cfg.flags & _lib.OPT_AVG_REPROJ) else 2
    for s in range(ns):
        shp = (B, 1, H >> s, W >> s)
        if tuple(disps[s].shape) != shp:
            raise _lib.DepthcoreError("disp[%d] must be %s, got %s" % (s, shp, tuple(disps[s].shape)))
        if tuple(cfg.color_s[s].shape) != (B, 3, H >> s, W >> s):
            raise _lib.DepthcoreError("color_s[%d] has shape %s" % (s, tuple(cfg.color_s[s].shape)))
        d.disp[s] = ptr(disps[s])
        d.color_s[s] = ptr(cfg.color_s[s])
        if cfg.noise is not None and not (cfg.flags & _lib.OPT_NO_AUTOMASK):
            if tuple(cfg.noise[s].shape) != (B, nch, H, W):
                raise _lib.DepthcoreError("noise[%d] must be %s" % (s, (B, nch, H, W)))
            d.noise[s] = ptr(cfg.noise[s])
    d.rng_seed = cfg.rng_seed
    if cfg.rng_seed_dev is not None:
        d.rng_seed_dev = ptr(cfg.rng_seed_dev, torch.int64)
    return d


class _PhotoLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cfg, T0, T1, *tensors):
        L = _lib.lib()
        # tensors = disps..., [predictive masks (one per scale)], [per-scale poses (T_-1, T_+1 per scale)]
        nm, nT = cfg.n_masks, cfg.n_Ts
        Ts = [_c(x.detach()) for x in tensors[len(tensors) - nT:]] if nT else []
        tensors = tensors[:len(tensors) - nT]
        T0, T1 = (None, None) if nT else (_c(T0.detach()), _c(T1.detach()))
        disps = [_c(x.detach()) for x in tensors[:len(tensors) - nm]]
        masks = [_c(x.detach()) for x in tensors[len(tensors) - nm:]] if nm else []
        dev = cfg.target.device
        B, _, H, W = cfg.target.shape
        ns = len(disps)
        # evaluation (nothing requires a gradient): the forward skips the gradient emission and its 24 B/pixel/scale
        ctx.no_grad = not any(ctx.needs_input_grad)
        # the all-the-way / split choice (dc_set_photo_full) is read ONCE, here, and pinned in the desc of the forward and of its
        # backward: the workspace layout both derive cannot change between them whatever the setter does in the meantime
        ctx.mode = _lib.OPT_PHOTO_FULL if L.dc_get_photo_full() else _lib.OPT_PHOTO_SPLIT
        d = _fill_desc(cfg, T0, T1, disps, ctx.no_grad, masks, Ts, ctx.mode)
        wsz = L.dc_photo_workspace(ctypes.byref(d))
        ws = torch.empty(wsz, dtype=torch.uint8, device=dev)
        d.workspace, d.workspace_bytes = ws.data_ptr(), wsz
        losses = torch.empty(ns + 1, dtype=torch.float32, device=dev)
        d.losses = ptr(losses)
        argmin = [torch.empty(B, H, W, dtype=torch.uint8, device=dev) for _ in range(ns)]
        ex = {"argmin": argmin}
        for s in range(ns):
            d.argmin[s] = argmin[s].data_ptr()
        if cfg.materialize:
            ex["depth"] = [torch.empty(B, 1, H, W, device=dev) for _ in range(ns)]
            ex["sample"] = [[torch.empty(B, H, W, 2, device=dev) for _ in range(2)] for _ in range(ns)]
            ex["color"] = [[torch.empty(B, 3, H, W, device=dev) for _ in range(2)] for _ in range(ns)]
            automask = not (cfg.flags & _lib.OPT_NO_AUTOMASK)
            ex["identity_selection"] = [torch.empty(B, H, W, device=dev) for _ in range(ns)] if automask else None
            for s in range(ns):
                d.depth[s] = ptr(ex["depth"][s])
                for f in range(2):
                    d.sample[s][f] = ptr(ex["sample"][s][f])
                    d.color[s][f] = ptr(ex["color"][s][f])
                if automask:
                    d.identity_selection[s] = ptr(ex["identity_selection"][s])
        check(L.dc_photo_fwd(ctypes.byref(d), stream(cfg.target)), "dc_photo_fwd")
        cfg.extras = ex
        ctx.cfg, ctx.ws, ctx.argmin, ctx.nm, ctx.nT = cfg, ws, argmin, nm, nT
        ctx.save_for_backward(*(() if nT else (T0, T1)), *disps, *masks, *Ts)
        return losses

    @staticmethod
    def backward(ctx, g_losses):
        L = _lib.lib()
        cfg = ctx.cfg
        rest = list(ctx.saved_tensors)
        T0, T1 = (None, None) if ctx.nT else (rest.pop(0), rest.pop(0))
        Ts = [rest.pop() for _ in range(ctx.nT)][::-1]
        disps, masks = (rest[:len(rest) - ctx.nm], rest[len(rest) - ctx.nm:]) if ctx.nm else (rest, [])
        d = _fill_desc(cfg, T0, T1, disps, ctx.no_grad, masks, Ts, ctx.mode)
        d.workspace, d.workspace_bytes = ctx.ws.data_ptr(), ctx.ws.numel()
        g = _c(g_losses.to(torch.float32))
        d.g_losses = ptr(g)
        d_disp = [torch.empty_like(x) for x in disps]
        dT = [None, None] if ctx.nT else [torch.empty_like(T0), torch.empty_like(T1)]
        dTs = [torch.empty_like(t) for t in Ts]
        for s in range(len(disps)):
            d.argmin[s] = ctx.argmin[s].data_ptr()
            d.d_disp[s] = ptr(d_disp[s])
            if ctx.nT:
                d.d_T_scale[s][0], d.d_T_scale[s][1] = ptr(dTs[2 * s]), ptr(dTs[2 * s + 1])
        if not ctx.nT:
            d.d_T[0], d.d_T[1] = ptr(dT[0]), ptr(dT[1])
        d_masks = [torch.empty_like(m) for m in masks]
        for s in range(len(d_masks)):
            d.d_pred_mask[s] = ptr(d_masks[s])
        check(L.dc_photo_bwd(ctypes.byref(d), stream(cfg.target)), "dc_photo_bwd")
        return (None, dT[0], dT[1], *d_disp, *d_masks, *dTs)


def pack_rgbx(x):
    """(B,3,H,W) float32 -> (B,H,W,4) pixel-interleaved RGBx: the layout the fused photometric kernels gather from (`PhotoConfig(
    packed=...)`).  The data step (depthcore.data.GpuPreprocessor) writes it directly as ("color_packed", f, 0)."""
    xx = _c(x.detach())
    B, C, H, W = xx.shape
    if C != 3:
        raise _lib.DepthcoreError("pack_rgbx takes (B,3,H,W) images")
    out = torch.empty(B, H, W, 4, dtype=torch.float32, device=xx.device)
    check(_lib.lib().dc_pack_rgbx(ptr(xx), ptr(out), B, H * W, stream(xx)), "dc_pack_rgbx")
    return out


def photometric_loss(cfg, T_m1, T_p1, disps, pred_masks=None, T_scales=None):
    """-> losses tensor (num_scales+1,): [loss/0, ..., loss]  (trainer.py:618-621).
    T_scales: `pose_model_type == "posecnn"` (trainer.py:490-499) -- a list of (T_-1, T_+1) pairs, one per scale, each built from
    the translation scaled by that scale's mean inverse depth; replaces T_m1 / T_p1 (pass None) and each gets its own gradient.
    pred_masks: opt.predictive_mask (trainer.py:571-584, needs disable_automasking) -- one FULL-resolution (B,2,H,W) mask per
    scale (the caller upsamples, trainer.py:574-577); the reprojection losses are multiplied by them inside the kernels and
    the masks get their gradient.  The BCE weighting term (trainer.py:579-581) is not part of this op."""
    cfg.n_masks = len(pred_masks) if pred_masks else 0
    flat_T = [t for pair in (T_scales or []) for t in pair]
    cfg.n_Ts = len(flat_T)
    return _PhotoLoss.apply(cfg, T_m1, T_p1, *disps, *(pred_masks or []), *flat_T)


def photo_algorithmic_bytes(cfg, T0, T1, disps, backward):
    d = _fill_desc(cfg, _c(T0.detach()), _c(T1.detach()), [_c(x.detach()) for x in disps])
    return _lib.lib().dc_photo_algorithmic_bytes(ctypes.byref(d), int(backward))


# ----------------------------------------------------------------------------------------------
# a5  transformation_from_parameters                                   (reference layers.py:28-103)
# ----------------------------------------------------------------------------------------------
class _PoseMatrix(torch.autograd.Function):
    @staticmethod
    def forward(ctx, axisangle, translation, invert):
        L = _lib.lib()
        aa = _c(axisangle.detach().reshape(-1, 3))
        tr = _c(translation.detach().reshape(-1, 3))
        B = aa.shape[0]
        M = torch.empty(B, 4, 4, dtype=torch.float32, device=aa.device)
        check(L.dc_pose_matrix_fwd(ptr(aa), ptr(tr), int(bool(invert)), ptr(M), B, stream(aa)), "dc_pose_matrix_fwd")
        ctx.save_for_backward(aa, tr)
        ctx.invert = int(bool(invert))
        ctx.shapes = (axisangle.shape, translation.shape)
        return M

    @staticmethod
    def backward(ctx, gM):
        L = _lib.lib()
        aa, tr = ctx.saved_tensors
        B = aa.shape[0]
        daa, dtr = torch.empty_like(aa), torch.empty_like(tr)
        g_c = _c(gM)      # named: stays alive until the launch is enqueued
        check(L.dc_pose_matrix_bwd(ptr(aa), ptr(tr), ctx.invert, ptr(g_c), ptr(daa), ptr(dtr), B, stream(aa)),
              "dc_pose_matrix_bwd")
        return daa.reshape(ctx.shapes[0]), dtr.reshape(ctx.shapes[1]), None


def pose_matrix(axisangle, translation, invert=False):
    return _PoseMatrix.apply(axisangle, translation, invert)


class _PoseHead(torch.autograd.Function):
    """The tail of PoseDecoder / PoseCNN and the cam_T_cam of their callers as one launch each way (dc_pose_head_fwd / _bwd).
    Outputs: vec (N, nf, 1, 6) -- NOT differentiable (the axisangle / translation entries of `outputs` are records; the loss
    reaches the pose networks through the matrices) -- and one (rows,4,4) matrix tensor per group."""

    @staticmethod
    def forward(ctx, y, nf, groups):
        L = _lib.lib()
        yy = _c(y.detach())
        N, C = yy.shape[0], yy.shape[1]
        P = yy.shape[2] * yy.shape[3]
        if C != 6 * nf:
            raise _lib.DepthcoreError("pose head: %d channels for %d predicted frames" % (C, nf))
        gs = (_lib.PoseGroup * len(groups))(*[_lib.PoseGroup(int(r0), int(rows), int(slot), int(bool(inv))) for r0, rows, slot, inv in groups])
        vec = torch.empty(N, nf, 1, 6, dtype=torch.float32, device=yy.device)
        Ms = [torch.empty(int(g[1]), 4, 4, dtype=torch.float32, device=yy.device) for g in groups]
        mp = (ctypes.c_void_p * len(groups))(*[m.data_ptr() for m in Ms])
        check(L.dc_pose_head_fwd(ptr(yy), N, nf, P, 0.01, gs, len(groups), ptr(vec), mp, stream(yy)), "dc_pose_head_fwd")
        ctx.save_for_backward(vec)
        ctx.cfg = (N, nf, P, yy.shape, tuple(groups))
        ctx.mark_non_differentiable(vec)
        return (vec,) + tuple(Ms)

    @staticmethod
    def backward(ctx, _gvec, *gMs):
        L = _lib.lib()
        vec, = ctx.saved_tensors
        N, nf, P, yshape, groups = ctx.cfg
        gs = (_lib.PoseGroup * len(groups))(*[_lib.PoseGroup(int(r0), int(rows), int(slot), int(bool(inv))) for r0, rows, slot, inv in groups])
        keep = [None if g is None else _c(g) for g in gMs]       # named: alive until the launch is enqueued
        dp = (ctypes.c_void_p * len(groups))(*[None if g is None else g.data_ptr() for g in keep])
        dy = torch.empty(yshape, dtype=torch.float32, device=vec.device)
        check(L.dc_pose_head_bwd(ptr(vec), N, nf, P, 0.01, gs, len(groups), dp, ptr(dy), stream(vec)), "dc_pose_head_bwd")
        return dy, None, None


def pose_head(y, nf, groups):
    """y (N, 6 nf, h, w), the pose network's last convolution -> (axisangle (N,nf,1,3), translation (N,nf,1,3), [cam_T_cam per
    group]); groups: (row0, rows, slot, invert) -- rows [row0, row0+rows) of predicted frame `slot`.  pose_decoder.py:50-54 +
    trainer.py:416-419 / 436-440 in one launch each way."""
    out = _PoseHead.apply(y, int(nf), tuple(tuple(int(v) for v in g) for g in groups))
    vec = out[0]
    return vec[..., :3], vec[..., 3:], list(out[1:])


# ----------------------------------------------------------------------------------------------
# a6  disp_to_depth                                                      (reference layers.py:16-25)
# ----------------------------------------------------------------------------------------------
class _DispToDepth(torch.autograd.Function):
    @staticmethod
    def forward(ctx, disp, min_depth, max_depth):
        L = _lib.lib()
        d = _c(disp.detach())
        scaled, depth = torch.empty_like(d), torch.empty_like(d)
        check(L.dc_disp_to_depth_fwd(ptr(d), ptr(scaled), ptr(depth), d.numel(), float(min_depth), float(max_depth),
                                     stream(d)), "dc_disp_to_depth_fwd")
        ctx.save_for_backward(d)
        ctx.lim = (float(min_depth), float(max_depth))
        return scaled, depth

    @staticmethod
    def backward(ctx, gs, gd):
        L = _lib.lib()
        (d,) = ctx.saved_tensors
        out = torch.empty_like(d)
        # keep both contiguous temporaries alive across the call: two unnamed temporaries would be freed and
        # could be handed the SAME block by the caching allocator before the launch
        gs_c = _c(gs) if gs is not None else None
        gd_c = _c(gd) if gd is not None else None
        check(L.dc_disp_to_depth_bwd(ptr(d), ptr(gs_c), ptr(gd_c), ptr(out), d.numel(), ctx.lim[0], ctx.lim[1],
                                     stream(d)), "dc_disp_to_depth_bwd")
        return out, None, None


def disp_to_depth(disp, min_depth, max_depth):
    return _DispToDepth.apply(disp, min_depth, max_depth)


# ----------------------------------------------------------------------------------------------
# a7  BackprojectDepth                                                  (reference layers.py:139-168)
# ----------------------------------------------------------------------------------------------
def pix_coords(B, H, W, device):
    L = _lib.lib()
    pc = torch.empty(B, 3, H * W, dtype=torch.float32, device=device)
    check(L.dc_pix_coords(ptr(pc), B, H, W, stream(pc)), "dc_pix_coords")
    return pc


class _Backproject(torch.autograd.Function):
    @staticmethod
    def forward(ctx, depth, inv_K):
        L = _lib.lib()
        d, ik = _c(depth.detach()), _c(inv_K.detach())
        B, _, H, W = d.shape
        cam = torch.empty(B, 4, H * W, dtype=torch.float32, device=d.device)
        check(L.dc_backproject_fwd(ptr(d), ptr(ik), ptr(cam), B, H, W, stream(d)), "dc_backproject_fwd")
        ctx.save_for_backward(ik)
        ctx.shape = d.shape
        return cam

    @staticmethod
    def backward(ctx, g):
        L = _lib.lib()
        (ik,) = ctx.saved_tensors
        B, _, H, W = ctx.shape
        dd = torch.empty(ctx.shape, dtype=torch.float32, device=g.device)
        g_c = _c(g)      # named: stays alive until the launch is enqueued
        check(L.dc_backproject_bwd(ptr(g_c), ptr(ik), ptr(dd), B, H, W, stream(g_c)), "dc_backproject_bwd")
        return dd, None


def backproject(depth, inv_K):
    return _Backproject.apply(depth, inv_K)


# ----------------------------------------------------------------------------------------------
# a8  Project3D                                                         (reference layers.py:171-193)
# ----------------------------------------------------------------------------------------------
class _Project3D(torch.autograd.Function):
    @staticmethod
    def forward(ctx, points, K, T, H, W, eps):
        L = _lib.lib()
        p, k, t = _c(points.detach()), _c(K.detach()), _c(T.detach())
        B = p.shape[0]
        grid = torch.empty(B, H, W, 2, dtype=torch.float32, device=p.device)
        check(L.dc_project3d_fwd(ptr(p), ptr(k), ptr(t), ptr(grid), B, H, W, float(eps), stream(p)), "dc_project3d_fwd")
        ctx.save_for_backward(p, k, t)
        ctx.dims = (B, H, W, float(eps))
        return grid

    @staticmethod
    def backward(ctx, g):
        L = _lib.lib()
        p, k, t = ctx.saved_tensors
        B, H, W, eps = ctx.dims
        dp, dT = torch.empty_like(p), torch.empty_like(t)
        ws = torch.empty(L.dc_project3d_bwd_workspace(B, H, W), dtype=torch.uint8, device=p.device)
        g_c = _c(g)      # named: stays alive until the launch is enqueued
        check(L.dc_project3d_bwd(ptr(p), ptr(k), ptr(t), ptr(g_c), ptr(dp), ptr(dT), ws.data_ptr(), B, H, W, eps,
                                 stream(p)), "dc_project3d_bwd")
        return dp, None, dT, None, None, None


def project3d(points, K, T, H, W, eps=1e-7):
    return _Project3D.apply(points, K, T, H, W, eps)


# ----------------------------------------------------------------------------------------------
# a9  F.grid_sample(bilinear, border)                                      (reference trainer.py:508)
# ----------------------------------------------------------------------------------------------
class _GridSample(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, grid, align_corners):
        L = _lib.lib()
        im, g = _c(img.detach()), _c(grid.detach())
        B, C, H, W = im.shape
        Ho, Wo = g.shape[1], g.shape[2]
        out = torch.empty(B, C, Ho, Wo, dtype=torch.float32, device=im.device)
        check(L.dc_grid_sample_fwd(ptr(im), ptr(g), ptr(out), B, C, H, W, Ho, Wo, int(bool(align_corners)), stream(im)),
              "dc_grid_sample_fwd")
        ctx.save_for_backward(im, g)
        ctx.ac = int(bool(align_corners))
        return out

    @staticmethod
    def backward(ctx, go):
        L = _lib.lib()
        im, g = ctx.saved_tensors
        B, C, H, W = im.shape
        Ho, Wo = g.shape[1], g.shape[2]
        dg = torch.empty_like(g)
        g_c = _c(go)      # named: stays alive until the launch is enqueued
        check(L.dc_grid_sample_bwd(ptr(im), ptr(g), ptr(g_c), ptr(dg), B, C, H, W, Ho, Wo, ctx.ac, stream(im)),
              "dc_grid_sample_bwd")
        return None, dg, None       # images are leaves without grad on this path


def grid_sample_border(img, grid, align_corners=False):
    return _GridSample.apply(img, grid, align_corners)


# ----------------------------------------------------------------------------------------------
# a10 F.interpolate(bilinear, align_corners=False)                         (reference trainer.py:474)
# ----------------------------------------------------------------------------------------------
class _UpsampleBilinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, Ho, Wo):
        L = _lib.lib()
        xx = _c(x.detach())
        B, C, h, w = xx.shape
        out = torch.empty(B, C, Ho, Wo, dtype=torch.float32, device=xx.device)
        check(L.dc_upsample_bilinear_fwd(ptr(xx), ptr(out), B * C, h, w, Ho, Wo, stream(xx)), "dc_upsample_bilinear_fwd")
        ctx.dims = (B, C, h, w, Ho, Wo)
        return out

    @staticmethod
    def backward(ctx, go):
        L = _lib.lib()
        B, C, h, w, Ho, Wo = ctx.dims
        dx = torch.empty(B, C, h, w, dtype=torch.float32, device=go.device)
        g_c = _c(go)      # named: stays alive until the launch is enqueued
        check(L.dc_upsample_bilinear_bwd(ptr(g_c), ptr(dx), B * C, h, w, Ho, Wo, stream(g_c)), "dc_upsample_bilinear_bwd")
        return dx, None, None


def upsample_bilinear(x, Ho, Wo):
    return _UpsampleBilinear.apply(x, Ho, Wo)


# ----------------------------------------------------------------------------------------------
# a3  upsample(x): nearest x2                                          (reference layers.py:196-199)
# ----------------------------------------------------------------------------------------------
class _UpsampleNearest2x(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        xx = _c(x.detach())
        B, C, h, w = xx.shape
        out = torch.empty(B, C, 2 * h, 2 * w, dtype=torch.float32, device=xx.device)
        check(_lib.lib().dc_upsample_nearest2x_fwd(ptr(xx), ptr(out), B * C, h, w, stream(xx)), "dc_upsample_nearest2x_fwd")
        return out

    @staticmethod
    def backward(ctx, go):
        g = _c(go)
        B, C, H2, W2 = g.shape
        dx = torch.empty(B, C, H2 // 2, W2 // 2, dtype=torch.float32, device=g.device)
        check(_lib.lib().dc_upsample_nearest2x_bwd(ptr(g), ptr(dx), B * C, H2 // 2, W2 // 2, stream(g)), "dc_upsample_nearest2x_bwd")
        return dx


def upsample_nearest2x(x):
    return _UpsampleNearest2x.apply(x)


# ----------------------------------------------------------------------------------------------
# a11 SSIM                                                              (reference layers.py:218-248)
# ----------------------------------------------------------------------------------------------
class _SSIM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y):
        L = _lib.lib()
        xx, yy = _c(x.detach()), _c(y.detach())
        B, C, H, W = xx.shape
        out = torch.empty_like(xx)
        check(L.dc_ssim_fwd(ptr(xx), ptr(yy), ptr(out), B * C, H, W, stream(xx)), "dc_ssim_fwd")
        ctx.save_for_backward(xx, yy)
        return out

    @staticmethod
    def backward(ctx, go):
        L = _lib.lib()
        xx, yy = ctx.saved_tensors
        B, C, H, W = xx.shape
        dx = torch.empty_like(xx) if ctx.needs_input_grad[0] else None
        dy = torch.empty_like(yy) if ctx.needs_input_grad[1] else None
        g_c = _c(go)      # named: stays alive until the launch is enqueued
        check(L.dc_ssim_bwd(ptr(xx), ptr(yy), ptr(g_c), ptr(dx), ptr(dy), B * C, H, W, stream(xx)), "dc_ssim_bwd")
        return dx, dy


def ssim(x, y):
    return _SSIM.apply(x, y)


# ----------------------------------------------------------------------------------------------
# a13 get_smooth_loss                                                   (reference layers.py:202-215)
# ----------------------------------------------------------------------------------------------
class _Smooth(torch.autograd.Function):
    @staticmethod
    def forward(ctx, disp, img):
        L = _lib.lib()
        d, im = _c(disp.detach()), _c(img.detach())
        B, _, h, w = d.shape
        C = im.shape[1]
        out = torch.empty(1, dtype=torch.float32, device=d.device)
        ws = torch.empty(L.dc_smooth_workspace(B, h, w), dtype=torch.uint8, device=d.device)
        check(L.dc_smooth_fwd(ptr(d), ptr(im), ptr(out), ws.data_ptr(), B, C, h, w, stream(d)), "dc_smooth_fwd")
        ctx.save_for_backward(d, im)
        return out.reshape(())

    @staticmethod
    def backward(ctx, g):
        L = _lib.lib()
        d, im = ctx.saved_tensors
        B, _, h, w = d.shape
        dd = torch.empty_like(d)
        g_c = _c(g.reshape(1))      # named: stays alive until the launch is enqueued
        check(L.dc_smooth_bwd(ptr(d), ptr(im), ptr(g_c), ptr(dd), B, im.shape[1], h, w, stream(d)),
              "dc_smooth_bwd")
        return dd, None


def smooth_loss(disp, img):
    return _Smooth.apply(disp, img)


# ----------------------------------------------------------------------------------------------
# measurement hook (bench.py): hipEvent timing of the dominant photometric kernels
# ----------------------------------------------------------------------------------------------
def profile_enable(max_launches):
    check(_lib.lib().dc_profile_enable(int(max_launches)), "dc_profile_enable")


def profile_collect():
    """-> dict(fwd_ms, fwd_launches, bwd_ms, bwd_launches, fwd_chain_ms, bwd_chain_ms): the dominant kernel of each
    direction and the whole launch chain of each direction; synchronises on the recorded events."""
    fm, bm, fc, bc = ctypes.c_double(0), ctypes.c_double(0), ctypes.c_double(0), ctypes.c_double(0)
    fn, bn = ctypes.c_int(0), ctypes.c_int(0)
    check(_lib.lib().dc_profile_collect(ctypes.byref(fm), ctypes.byref(fn), ctypes.byref(bm), ctypes.byref(bn),
                                        ctypes.byref(fc), ctypes.byref(bc)), "dc_profile_collect")
    return {"fwd_ms": fm.value, "fwd_launches": fn.value, "bwd_ms": bm.value, "bwd_launches": bn.value,
            "fwd_chain_ms": fc.value, "bwd_chain_ms": bc.value}


def conv_profile_enable(max_launches, every=1):
    """hipEvent pairs around every `every`-th Winograd conv launch of each kind (0 launches: disable and free)."""
    check(_lib.lib().dc_conv_profile_enable(int(max_launches), int(every)), "dc_conv_profile_enable")


def conv_profile_collect(kind):
    """kind 0: wino_ps_kernel (conv forward / data gradient), 1: wino_wgrad_kernel.
    -> dict(ms, flops (SURVEY 8d algorithmic), executed_flops (issued to the matrix cores), bytes (algorithmic), launches)."""
    ms, fl, ex, by = ctypes.c_double(0), ctypes.c_double(0), ctypes.c_double(0), ctypes.c_double(0)
    n = ctypes.c_int(0)
    check(_lib.lib().dc_conv_profile_collect(int(kind), ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(ex), ctypes.byref(by),
                                             ctypes.byref(n)), "dc_conv_profile_collect")
    return {"ms": ms.value, "flops": fl.value, "executed_flops": ex.value, "bytes": by.value, "launches": n.value}


# ----------------------------------------------------------------------------------------------
# a2/a3 fused decoder block: act(conv3x3(pad1(cat(up2?(x0), x1))) + bias)
#                                   (reference layers.py:106-136,196-199; networks/depth_decoder.py:50-66)
# ----------------------------------------------------------------------------------------------
ACT_NONE, ACT_ELU, ACT_SIGMOID, ACT_RELU, ACT_TANH = 0, 1, 2, 3, 4
PAD_REFLECT, PAD_ZERO = 0, 1


class _Conv3x3(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x0, x1, weight, bias, up0, act, pad, fork0=None):
        L = _lib.lib()
        a0 = _c(x0.detach())
        a1 = _c(x1.detach()) if x1 is not None else None
        w = _c(weight.detach())
        bs = _c(bias.detach()) if bias is not None else None
        B, C0, h0, w0 = a0.shape
        H, W = (h0 * 2, w0 * 2) if up0 else (h0, w0)
        C1 = 0 if a1 is None else a1.shape[1]
        Co = w.shape[0]
        if tuple(w.shape) != (Co, C0 + C1, 3, 3):
            raise _lib.DepthcoreError("weight %s does not match inputs (%d+%d channels)" % (tuple(w.shape), C0, C1))
        if a1 is not None and tuple(a1.shape) != (B, C1, H, W):
            raise _lib.DepthcoreError("skip tensor %s must be %s" % (tuple(a1.shape), (B, C1, H, W)))
        y = torch.empty(B, Co, H, W, dtype=torch.float32, device=a0.device)
        ctx.prec = _use_precision(_precision[0])
        ws = torch.empty(L.dc_conv3x3_fwd_workspace(C0, C1, B, Co, H, W), dtype=torch.uint8, device=a0.device)
        check(L.dc_conv3x3_fwd(ptr(a0), C0, int(up0), ptr(a1), C1, ptr(w), ptr(bs), ptr(y), ws.data_ptr(), B, Co, H, W,
                               int(act), int(pad), stream(a0)), "dc_conv3x3_fwd")
        ctx.save_for_backward(a0, a1, w, y)
        if act == ACT_RELU:
            _record_kink("relu", y)
        ctx.cfg = (int(up0), int(act), int(pad), bias is not None)
        ctx.slots = (_slot(weight), _slot(bias) if bias is not None else None)
        if ctx.needs_input_grad[2]:
            WgradLanes.count_use(weight)     # (no lane of its own; a weight shared with a laned op must not look single-use)
        # x0 shared with another fused block (the decoder's x feeds dispconv AND the next upconv): a pair GradFork -- whichever
        # backward runs first parks its dx0, the second adds it in the pass that writes its own.  x1 an encoder feature map
        # with a SkipSum: dx1 is offered to the feature's primary consumer (SkipSum)
        ctx.fork0 = fork0 if (fork0 is not None and x0.requires_grad) else None
        ctx.skip1 = skip_of(x1) if (x1 is not None and x1.requires_grad) else None
        return y

    @staticmethod
    def backward(ctx, gy):
        L = _lib.lib()
        a0, a1, w, y = ctx.saved_tensors
        up0, act, pad, has_bias = ctx.cfg
        B, C0 = a0.shape[0], a0.shape[1]
        C1 = 0 if a1 is None else a1.shape[1]
        Co, H, W = y.shape[1], y.shape[2], y.shape[3]
        need = ctx.needs_input_grad
        dx0 = torch.empty_like(a0) if need[0] else None
        dx1 = torch.empty_like(a1) if (a1 is not None and need[1]) else None
        dw = _grad_dst(ctx.slots[0], w) if need[2] else None
        db = None
        if has_bias and need[3]:
            db = _grad_dst(ctx.slots[1], None)
            if db is None:
                db = torch.empty(Co, dtype=torch.float32, device=y.device)
        _use_precision(ctx.prec)
        ws = torch.empty(L.dc_conv3x3_bwd_workspace(C0, C1, B, Co, H, W), dtype=torch.uint8, device=y.device)
        g_c = _c(gy)      # named: stays alive until the launch is enqueued
        fork0 = ctx.fork0 if dx0 is not None else None
        first = fork0 is not None and fork0.arrive() == 0
        add0 = fork0.take() if (fork0 is not None and not first) else None
        if add0 is not None and (add0.shape != a0.shape or not add0.is_contiguous()):
            raise _lib.DepthcoreError("GradFork: parked gradient %s does not match the shared input %s" % (tuple(add0.shape), tuple(a0.shape)))
        check(L.dc_conv3x3_bwd_add(ptr(a0), C0, up0, ptr(a1), C1, ptr(w), ptr(y), ptr(g_c), ptr(dx0), ptr(dx1), ptr(add0), None, ptr(dw),
                                   ptr(db), ws.data_ptr(), B, Co, H, W, act, pad, stream(a0)), "dc_conv3x3_bwd_add")
        if first:
            fork0.park(dx0)            # the other block adds it and returns the sum
            dx0 = None
        if ctx.skip1 is not None:
            dx1 = ctx.skip1.offer(dx1)
        return dx0, dx1, dw, db, None, None, None, None


def conv3x3_block(x0, x1, weight, bias, up0=False, act=ACT_NONE, pad=PAD_REFLECT, fork0=None):
    return _Conv3x3.apply(x0, x1, weight, bias, up0, act, pad, fork0)


# ----------------------------------------------------------------------------------------------
# a1 the gradient of a residual block's input: summed inside conv1's data-gradient kernel, not by autograd
# ----------------------------------------------------------------------------------------------
_parked_forks = weakref.WeakSet()     # GradForks holding a parked gradient (assert_no_dangling_sums)


class GradFork:
    """The input x of a residual block without a downsample branch has two consumers: conv1 and the skip connection into the
    last BatchNorm (+ add + ReLU).  Autograd would add their two gradients in a separate elementwise pass (31 such passes per
    C2 step, 42 at C3).  With a GradFork shared by the two ops, the last BatchNorm's backward -- which always runs first:
    conv1's output feeds it -- parks the skip's gradient here and reports NO gradient for its `res` input; conv1's backward
    picks it up and its data-gradient kernel adds it in the store epilogue (dc_wino3x3_dgrad_add / dc_conv1x1_dgrad_add), so
    x receives the complete gradient from conv1 alone.  One backward pass per forward (no double backward / retain_graph
    replays): a fork that is asked twice, or whose parked gradient is never collected, raises -- the latter from
    `assert_no_dangling_sums()` (a pair fork whose second reader never runs its backward, e.g. a ("disp", i > 0) output of
    the depth decoder that is left out of the loss: the parked gradient would be lost silently otherwise)."""
    __slots__ = ("addend", "armed", "pair", "arrived", "__weakref__")

    def __init__(self, pair=False):
        self.addend = None
        self.armed = True
        # pair: x feeds TWO convolutions with the addend epilogue (Bottleneck conv1 and the 1x1 `downsample`): whichever
        # backward runs first parks its data gradient and reports none, the second adds it (no assumption about the order)
        self.pair = pair
        self.arrived = 0

    def arrive(self):
        """pair forks: 0 for the first convolution backward to get here, 1 for the second."""
        if not self.pair or self.arrived >= 2:
            raise _lib.DepthcoreError("GradFork: more backward calls than the two convolutions that share the input")
        self.arrived += 1
        return self.arrived - 1

    def park(self, dres):
        if not self.armed or self.addend is not None:
            raise _lib.DepthcoreError("GradFork: the skip gradient was produced twice (a second backward through the same block?)")
        self.addend = dres
        _parked_forks.add(self)

    def take(self):
        if not self.armed:
            raise _lib.DepthcoreError("GradFork: conv1's backward ran twice for one forward")
        self.armed = False
        a, self.addend = self.addend, None
        _parked_forks.discard(self)
        if self.pair and a is None:
            raise _lib.DepthcoreError("GradFork: the first convolution's data gradient was never parked")
        return a


class SkipSum:
    """A tensor with a PRIMARY consumer whose data-gradient kernel can add another gradient on its way out (the stem output's
    max-pool, a stage's first block) and SECONDARY consumers that run their backward earlier (the depth decoder's skip
    connections, networks/depth_decoder.py:57-59).  The producer tags the tensor (`t._dc_skip = SkipSum()`); the primary
    `arm()`s it in its forward when its backward is certain to run; a secondary `offer()`s its gradient in its backward --
    taken and reported as None while the sum is armed and empty, handed back (autograd then sums as usual) otherwise; the
    primary `take()`s whatever was offered and adds it in its store epilogue.  The elementwise sums autograd would launch
    (16 per C2 step, 41 us for the stem's output alone) become one more read in kernels that write the gradient anyway.
    An offered gradient that nobody took is a lost gradient: `assert_no_dangling_sums()` (Trainer, after every backward)."""
    __slots__ = ("addend", "armed", "__weakref__")
    _pending = weakref.WeakSet()

    def __init__(self):
        self.addend, self.armed = None, False

    def arm(self):
        self.armed = True

    def offer(self, g):
        if g is None or not self.armed or self.addend is not None:
            return g
        self.addend = g
        SkipSum._pending.add(self)
        return None

    def take(self):
        self.armed = False
        a, self.addend = self.addend, None
        SkipSum._pending.discard(self)
        return a


def skip_of(t):
    """The SkipSum a producer attached to tensor `t`, or None."""
    return getattr(t, "_dc_skip", None) if t is not None else None


def assert_no_dangling_sums():
    left = [s for s in SkipSum._pending if s.addend is not None]
    for s in left:
        s.take()
    forks = [f for f in _parked_forks if f.addend is not None]
    for f in forks:
        f.addend, f.armed = None, False
        _parked_forks.discard(f)
    if left:
        raise _lib.DepthcoreError("%d gradient(s) were handed to a SkipSum whose primary consumer never collected them" % len(left))
    if forks:
        raise _lib.DepthcoreError("%d gradient(s) were parked in a GradFork whose other reader never ran its backward (an output "
                                  "of the fused decoder left out of the loss?): the gradient below that point is incomplete"
                                  % len(forks))


# ----------------------------------------------------------------------------------------------
# a1 training-mode BatchNorm2d (+ residual add) (+ ReLU), fused  (ResNet BasicBlock / Bottleneck / stem)
# ----------------------------------------------------------------------------------------------
class _BNReLU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, res, gamma, beta, running_mean, running_var, eps, momentum, relu, groups, fork=None):
        L = _lib.lib()
        xx = _c(x.detach())
        rr = _c(res.detach()) if res is not None else None
        g, b = _c(gamma.detach()), _c(beta.detach())
        N, C, H, W = xx.shape
        y = torch.empty_like(xx)
        mean = torch.empty(groups * C, dtype=torch.float32, device=xx.device)
        invstd = torch.empty(groups * C, dtype=torch.float32, device=xx.device)
        ws = torch.empty(L.dc_bn_workspace(N, C, H * W), dtype=torch.uint8, device=xx.device)
        nmask = L.dc_bn_mask_bytes(N, C, H * W) if relu else 0
        mask = torch.empty(nmask, dtype=torch.uint8, device=xx.device) if nmask else None      # [y > 0] as bits
        check(L.dc_bn_relu_fwd(ptr(xx), ptr(rr), ptr(g), ptr(b), ptr(y), ptr(mean), ptr(invstd),
                               ptr(running_mean), ptr(running_var), ws.data_ptr(), mask.data_ptr() if nmask else None,
                               N, C, H * W, float(eps), float(momentum), int(relu), int(groups), stream(xx)), "dc_bn_relu_fwd")
        ctx.save_for_backward(xx, y, g, mean, invstd, mask)
        if relu:
            _record_kink("relu", y)
        ctx.cfg = (int(relu), res is not None, int(groups))
        ctx.slots = (_slot(gamma), _slot(beta))
        ctx.fork = fork if (fork is not None and res is not None and res.requires_grad) else None
        return y

    @staticmethod
    def backward(ctx, gy):
        L = _lib.lib()
        xx, y, g, mean, invstd, mask = ctx.saved_tensors
        relu, has_res, groups = ctx.cfg
        N, C, H, W = xx.shape
        g_c = _c(gy)
        dx = torch.empty_like(xx)
        dres = torch.empty_like(xx) if (has_res and ctx.needs_input_grad[1]) else None
        dgamma = _grad_dst(ctx.slots[0], g)
        dbeta = _grad_dst(ctx.slots[1], g)
        ws = torch.empty(L.dc_bn_workspace(N, C, H * W), dtype=torch.uint8, device=xx.device)
        check(L.dc_bn_relu_bwd(ptr(xx), ptr(y), ptr(g_c), ptr(g), ptr(mean), ptr(invstd), ptr(dx), ptr(dres), ptr(dgamma),
                               ptr(dbeta), ws.data_ptr(), mask.data_ptr() if mask is not None else None, N, C, H * W, relu,
                               groups, stream(xx)), "dc_bn_relu_bwd")
        if ctx.fork is not None and dres is not None:
            ctx.fork.park(dres)             # conv1's data-gradient kernel adds it (GradFork)
            dres = None
        return dx, dres, dgamma, dbeta, None, None, None, None, None, None, None


def bn_relu(x, bn, res=None, relu=True, groups=1, fork=None):
    """y = relu?(bn(x) [+ res]) with `bn` an nn.BatchNorm2d in training mode (batch statistics; updates its
    running_mean / running_var in place; `num_batches_tracked` is advanced by the caller).  `groups` > 1:
    the batch is that many independent sub-batches (statistics and running-stat updates per sub-batch)."""
    mom = 0.1 if bn.momentum is None else bn.momentum
    return _BNReLU.apply(x, res, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, mom, relu, groups, fork)


# ----------------------------------------------------------------------------------------------
# a1 nn.MaxPool2d(3, 2, 1) of the ResNet stem
# ----------------------------------------------------------------------------------------------
class _MaxPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, skip=None):
        L = _lib.lib()
        xx = _c(x.detach())
        N, C, H, W = xx.shape
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        y = torch.empty(N, C, Ho, Wo, dtype=torch.float32, device=xx.device)
        code = torch.empty(N, C, Ho, Wo, dtype=torch.uint8, device=xx.device)
        check(L.dc_maxpool3x3s2_fwd(ptr(xx), ptr(y), code.data_ptr(), N * C, H, W, stream(xx)), "dc_maxpool3x3s2_fwd")
        ctx.save_for_backward(code)
        _record_kink("maxpool", code)
        ctx.dims = (N, C, H, W)
        ctx.skip = skip if (skip is not None and x.requires_grad) else None
        if ctx.skip is not None:
            ctx.skip.arm()
        return y

    @staticmethod
    def backward(ctx, gy):
        L = _lib.lib()
        (code,) = ctx.saved_tensors
        N, C, H, W = ctx.dims
        g_c = _c(gy)
        dx = torch.empty(N, C, H, W, dtype=torch.float32, device=gy.device)
        add = ctx.skip.take() if ctx.skip is not None else None
        if add is not None and (tuple(add.shape) != (N, C, H, W) or not add.is_contiguous()):
            raise _lib.DepthcoreError("SkipSum: gradient %s does not match the pooled tensor %s" % (tuple(add.shape), (N, C, H, W)))
        check(L.dc_maxpool3x3s2_bwd_add(ptr(g_c), code.data_ptr(), ptr(dx), ptr(add), N * C, H, W, stream(g_c)), "dc_maxpool3x3s2_bwd_add")
        return dx, None


def maxpool3x3s2(x, skip=None):
    """nn.MaxPool2d(3, 2, 1).  `skip`: the SkipSum of x (its other consumers' gradients are added in the backward kernel)."""
    return _MaxPool.apply(x, skip)


# ----------------------------------------------------------------------------------------------
# a1 trunk conv3x3 (stride 1, zero pad 1, no bias): fused Winograd F(2x2,3x3) forward and data gradient
# ----------------------------------------------------------------------------------------------
class _WinoConv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, fork=None):
        L = _lib.lib()
        xx, ww = _c(x.detach()), _c(weight.detach())
        B, Ci, H, W = xx.shape
        Co = ww.shape[0]
        y = torch.empty(B, Co, H, W, dtype=torch.float32, device=xx.device)
        ctx.prec = _use_precision(_precision[0])
        ws = torch.empty(L.dc_wino3x3_workspace(B, Ci, Co, H, W), dtype=torch.uint8, device=xx.device)
        check(L.dc_wino3x3_fwd(ptr(xx), ptr(ww), ptr(y), ws.data_ptr(), B, Ci, Co, H, W, stream(xx)), "dc_wino3x3_fwd")
        ctx.save_for_backward(xx, ww)
        ctx.slot = _slot(weight)
        ctx.param = _lane_param(ctx, 1, weight)
        ctx.fork = fork if x.requires_grad else None
        return y

    @staticmethod
    def backward(ctx, gy):
        L = _lib.lib()
        xx, ww = ctx.saved_tensors
        B, Ci, H, W = xx.shape
        Co = ww.shape[0]
        g_c = _c(gy)
        gx = gw = None
        _use_precision(ctx.prec)
        add = _fork_addend(ctx, xx)
        if ctx.needs_input_grad[0]:
            gx = torch.empty_like(xx)
            ws = torch.empty(L.dc_wino3x3_workspace(B, Ci, Co, H, W), dtype=torch.uint8, device=xx.device)
            check(L.dc_wino3x3_dgrad_add(ptr(g_c), ptr(ww), ptr(gx), ptr(add), ws.data_ptr(), B, Ci, Co, H, W, stream(g_c)),
                  "dc_wino3x3_dgrad_add")
        if ctx.needs_input_grad[1]:
            with WgradLanes.lane(ctx.param, xx, g_c):
                gw = _grad_dst(ctx.slot, ww)
                ws = torch.empty(L.dc_wino3x3_wgrad_workspace(B, Ci, Co, H, W), dtype=torch.uint8, device=xx.device)
                check(L.dc_wino3x3_wgrad(ptr(xx), ptr(g_c), ptr(gw), ws.data_ptr(), B, Ci, Co, H, W, stream(xx)),
                      "dc_wino3x3_wgrad")
        return gx, gw, None


def _lane_param(ctx, idx, weight):
    """forward(): the parameter whose weight gradient may take a lane (WgradLanes), counted as used once more."""
    if not ctx.needs_input_grad[idx]:
        return None
    WgradLanes.count_use(weight)
    return weight


def _fork_addend(ctx, xx):
    """The skip gradient parked for this convolution's input (GradFork), or None."""
    if ctx.fork is None:
        return None
    add = ctx.fork.take()
    if add is not None and not ctx.needs_input_grad[0]:
        raise _lib.DepthcoreError("GradFork: a skip gradient is waiting but the convolution's input gradient was not requested")
    if add is not None and (add.shape != xx.shape or not add.is_contiguous()):
        raise _lib.DepthcoreError("GradFork: skip gradient %s does not match the block input %s" % (tuple(add.shape), tuple(xx.shape)))
    return add


def wino_conv3x3(x, weight, fork=None):
    """F.conv2d(x, weight, None, 1, 1) for 3x3 kernels on even-width maps (fused Winograd on the matrix cores).
    The Winograd kernels use 32-bit buffer offsets: a tensor of 2 GiB or more takes the direct implicit-GEMM kernels of
    the fused conv block (same arithmetic contract, size_t indexing) instead."""
    if max(x.numel(), x.numel() // x.shape[1] * weight.shape[0]) * 4 >= 0x7fffffff:
        if fork is not None:
            fork.armed = False          # (the direct kernels have no addend: the caller falls back to autograd's sum)
            raise _lib.DepthcoreError("GradFork is not available on the >= 2 GiB path; call without a fork")
        return conv3x3_block(x, None, weight, None, False, ACT_NONE, PAD_ZERO)
    return _WinoConv.apply(x, weight, fork)


# ---- transformed-weight cache of the Winograd kernels (include/depthcore.h: dc_wino_cache_*) ----------------------
def _wino_release(owner):
    """weakref.finalize target: runs when a WinoWeightCache is closed or collected (never touches the dead object)."""
    try:
        _lib.lib().dc_wino_cache_release_owner(owner)
    except Exception:
        pass


class WinoWeightCache:
    """A model's registrations in libdepthcore's Winograd weight cache (one owner per Trainer).

    `refresh()` at the start of a training step transforms every registered 3x3 weight in one launch; the step's
    convolutions (forward, data gradient, and every frame of the sequence models) then skip their per-launch transform;
    `invalidate()` once the backward is done, before the optimiser rewrites the weights.

    Every cache object is its own OWNER in the library: its refresh launch (and a hipGraph that captured it) reads this
    object's weights only, which `_keep` holds alive for as long as the owner exists.  Constructing, closing or garbage-
    collecting another cache can therefore neither free nor rewrite anything a captured graph of this one reads; `close()`
    (also run when the object is collected) releases this owner only, and its device buffers are parked until `clear_all()`."""

    def __init__(self, params):
        L = _lib.lib()
        self._keep = []
        self._owner = int(L.dc_wino_cache_new_owner())
        for p_ in params:
            # (3x3: Winograd U / prepared bf16 weights; 1x1: the split-operand GEMMs' three bf16 pieces, csrc/gemm1x1_x3.hip)
            if p_.dim() == 4 and tuple(p_.shape[2:]) in ((3, 3), (1, 1)) and p_.is_cuda and p_.dtype == torch.float32 and p_.is_contiguous():
                check(L.dc_wino_cache_register(self._owner, p_.data_ptr(), int(p_.shape[1]), int(p_.shape[0])),
                      "dc_wino_cache_register")
                self._keep.append(p_)           # the registry holds raw addresses: keep the tensors alive with it
        self._fin = weakref.finalize(self, _wino_release, self._owner)

    def refresh(self):
        if self._fin.alive and self._keep:
            check(_lib.lib().dc_wino_cache_refresh(self._owner, stream(self._keep[0])), "dc_wino_cache_refresh")

    def invalidate(self):
        if self._fin.alive:
            _lib.lib().dc_wino_cache_invalidate(self._owner)

    def variants(self):
        """cached (weight, pass, tile layout) variants in the whole registry (0 once every owner is closed)."""
        return int(_lib.lib().dc_wino_cache_variants()) if self._fin.alive else 0

    def close(self):
        self._fin()                             # unregister once (idempotent)
        self._keep = []

    @staticmethod
    def clear_all():
        """Drop every registration and free every parked buffer (device-synchronising).  Only when no captured graph that
        used the cache will be replayed again."""
        check(_lib.lib().dc_wino_cache_clear(), "dc_wino_cache_clear")


# ----------------------------------------------------------------------------------------------
# a1 / a4  1x1 convolutions (stride 1 / 2; optional bias + activation): MFMA GEMMs on the NCHW tensors
# ----------------------------------------------------------------------------------------------
class _Conv1x1(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, stride, act, fork=None, skip=None):
        L = _lib.lib()
        xx, ww = _c(x.detach()), _c(weight.detach())
        bs = _c(bias.detach()) if bias is not None else None
        B, Ci, Hi, Wi = xx.shape
        Co = ww.shape[0]
        if ww.numel() != Co * Ci or (bs is not None and bs.numel() != Co):
            raise _lib.DepthcoreError("1x1 weight %s / bias do not match %d input channels" % (tuple(ww.shape), Ci))
        y = torch.empty(B, Co, Hi // stride, Wi // stride, dtype=torch.float32, device=xx.device)
        # dc_set_gemm_split: the same fp32 GEMM through three bf16 pieces per operand on the bf16 matrix cores (csrc/gemm1x1_x3.hip)
        ctx.split = bool(L.dc_get_gemm_split())
        if ctx.split and L.dc_gemm1x1x3_fwd_ok(B, Ci, Co, Hi, Wi, int(stride)):
            ws = torch.empty(L.dc_gemm1x1x3_workspace(Ci, Co), dtype=torch.uint8, device=xx.device)
            check(L.dc_gemm1x1x3_fwd(ptr(xx), ptr(ww), ptr(bs), ptr(y), ws.data_ptr(), B, Ci, Co, Hi, Wi, int(stride), int(act), stream(xx)),
                  "dc_gemm1x1x3_fwd")
        else:
            check(L.dc_conv1x1_bias_act_fwd(ptr(xx), ptr(ww), ptr(bs), ptr(y), B, Ci, Co, Hi, Wi, int(stride), int(act), stream(xx)),
                  "dc_conv1x1_bias_act_fwd")
        plain = bias is None and act == ACT_NONE
        if act == ACT_RELU:
            _record_kink("relu", y)
        ctx.save_for_backward(xx, ww, None if plain else y)
        ctx.cfg = (int(stride), int(act), bias is not None)
        ctx.slots = (_slot(weight), _slot(bias) if bias is not None else None)
        ctx.param = _lane_param(ctx, 1, weight)
        ctx.fork = fork if x.requires_grad else None
        # x's SkipSum: this convolution is (one of) the primary consumer(s) -- the member of the pair fork that runs second
        # returns the complete gradient of x and collects the secondary consumers' gradients with it
        ctx.skip = skip if (skip is not None and ctx.fork is not None and ctx.fork.pair) else None
        if ctx.skip is not None:
            ctx.skip.arm()
        return y

    @staticmethod
    def backward(ctx, gy):
        L = _lib.lib()
        xx, ww, y = ctx.saved_tensors
        B, Ci, Hi, Wi = xx.shape
        s_, act, has_bias = ctx.cfg
        Co = ww.shape[0]
        g_c = _c(gy)
        gx = gw = gb = None
        if y is not None:      # gradient of the pre-activation (+ bias gradient), one pass
            P = (Hi // s_) * (Wi // s_)
            gpre = torch.empty_like(g_c) if act != ACT_NONE else None
            if has_bias and ctx.needs_input_grad[2]:
                gb = _grad_dst(ctx.slots[1], None)
                if gb is None:
                    gb = torch.empty(Co, dtype=torch.float32, device=xx.device)
            check(L.dc_bias_act_bwd(ptr(y), ptr(g_c), ptr(gpre), ptr(gb), B, Co, P, act, stream(xx)), "dc_bias_act_bwd")
            if gpre is not None:
                g_c = gpre
        first_of_pair = ctx.fork is not None and ctx.fork.pair and ctx.fork.arrive() == 0
        add = None if first_of_pair else _fork_addend(ctx, xx)
        if ctx.needs_input_grad[0]:
            gx = torch.empty_like(xx)
            add2 = ctx.skip.take() if (ctx.skip is not None and not first_of_pair) else None
            if add2 is not None and (add2.shape != xx.shape or not add2.is_contiguous()):
                raise _lib.DepthcoreError("SkipSum: gradient %s does not match the input %s" % (tuple(add2.shape), tuple(xx.shape)))
            if ctx.split and L.dc_gemm1x1x3_dgrad_ok(B, Ci, Co, Hi, Wi, s_):
                ws = torch.empty(L.dc_gemm1x1x3_workspace(Ci, Co), dtype=torch.uint8, device=xx.device)
                check(L.dc_gemm1x1x3_dgrad(ptr(g_c), ptr(ww), ptr(gx), ws.data_ptr(), ptr(add), ptr(add2), B, Ci, Co, Hi, Wi, s_, stream(xx)),
                      "dc_gemm1x1x3_dgrad")
            else:
                check(L.dc_conv1x1_dgrad_add2(ptr(g_c), ptr(ww), ptr(gx), ptr(add), ptr(add2), B, Ci, Co, Hi, Wi, s_, stream(xx)),
                      "dc_conv1x1_dgrad_add2")
            if first_of_pair:
                ctx.fork.park(gx)          # the other convolution of the pair adds it and returns the sum
                gx = None
        elif first_of_pair:
            raise _lib.DepthcoreError("GradFork: the shared input needs no gradient")
        if ctx.needs_input_grad[1]:
            with WgradLanes.lane(ctx.param, xx, g_c):
                gw = _grad_dst(ctx.slots[0], ww)
                if ctx.split and L.dc_gemm1x1x3_wgrad_ok(B, Ci, Co, Hi, Wi, s_):
                    ws = torch.empty(L.dc_gemm1x1x3_wgrad_workspace(B, Ci, Co, Hi, Wi, s_), dtype=torch.uint8, device=xx.device)
                    check(L.dc_gemm1x1x3_wgrad(ptr(xx), ptr(g_c), ptr(gw), ws.data_ptr(), B, Ci, Co, Hi, Wi, s_, stream(xx)),
                          "dc_gemm1x1x3_wgrad")
                else:
                    ws = torch.empty(L.dc_conv1x1_wgrad_workspace(B, Ci, Co, Hi, Wi, s_), dtype=torch.uint8, device=xx.device)
                    check(L.dc_conv1x1_wgrad(ptr(xx), ptr(g_c), ptr(gw), ws.data_ptr(), B, Ci, Co, Hi, Wi, s_, stream(xx)),
                          "dc_conv1x1_wgrad")
        return gx, gw, gb, None, None, None, None


def conv1x1(x, weight, stride=1, bias=None, act=ACT_NONE, fork=None, skip=None):
    """act(F.conv2d(x, weight, bias, stride)) for (Co,Ci,1,1) weights; stride 2 needs even H, W.  `fork`: GradFork; `skip`:
    the SkipSum of x (used when `fork` is a pair fork)."""
    return _Conv1x1.apply(x, weight, bias, stride, act, fork, skip)


# ----------------------------------------------------------------------------------------------
# a1 strided trunk convolutions: 7x7 / 2 stem and 3x3 / 2 (no bias), implicit GEMMs on the matrix cores
# ----------------------------------------------------------------------------------------------
def conv_s2_supported(x, weight):
    B, Ci, Hi, Wi = x.shape
    return bool(_lib.lib().dc_convs2_supported(B, Ci, weight.shape[0], Hi, Wi, weight.shape[-1]))


class _ConvS2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, fork=None):
        L = _lib.lib()
        xx, ww = _c(x.detach()), _c(weight.detach())
        B, Ci, Hi, Wi = xx.shape
        Co, ks = ww.shape[0], ww.shape[-1]
        if tuple(ww.shape) != (Co, Ci, ks, ks) or not L.dc_convs2_supported(B, Ci, Co, Hi, Wi, ks):
            raise _lib.DepthcoreError("strided convolution %s on input %s is outside dc_convs2_* (3x3 / 7x7, stride 2, even "
                                      "sizes, output width %% 4 == 0)" % (tuple(ww.shape), tuple(xx.shape)))
        y = torch.empty(B, Co, Hi // 2, Wi // 2, dtype=torch.float32, device=xx.device)
        ctx.prec = _use_precision(_precision[0])
        ws = torch.empty(L.dc_convs2_fwd_workspace(B, Ci, Co, Hi, Wi, ks), dtype=torch.uint8, device=xx.device)
        check(L.dc_convs2_fwd(ptr(xx), ptr(ww), ptr(y), ws.data_ptr(), B, Ci, Co, Hi, Wi, ks, stream(xx)), "dc_convs2_fwd")
        ctx.save_for_backward(xx, ww)
        ctx.slot = _slot(weight)
        ctx.param = _lane_param(ctx, 1, weight)
        ctx.fork = fork if (fork is not None and x.requires_grad) else None     # pair GradFork with the block's `downsample`
        return y

    @staticmethod
    def backward(ctx, gy):
        L = _lib.lib()
        xx, ww = ctx.saved_tensors
        B, Ci, Hi, Wi = xx.shape
        Co, ks = ww.shape[0], ww.shape[-1]
        g_c = _c(gy)
        gx = gw = None
        _use_precision(ctx.prec)
        first = ctx.fork is not None and ctx.fork.arrive() == 0
        if ctx.needs_input_grad[0]:
            n = L.dc_convs2_dgrad_workspace(B, Ci, Co, Hi, Wi, ks)
            if not n:
                raise _lib.DepthcoreError("no data gradient for this strided convolution (the 7x7 stem reads the image)")
            gx = torch.empty_like(xx)
            ws = torch.empty(n, dtype=torch.uint8, device=xx.device)
            check(L.dc_convs2_dgrad(ptr(g_c), ptr(ww), ptr(gx), ws.data_ptr(), B, Ci, Co, Hi, Wi, ks, stream(xx)), "dc_convs2_dgrad")
            if first:
                ctx.fork.park(gx)          # the 1x1 `downsample` adds it in its store epilogue and returns the sum
                gx = None
            elif ctx.fork is not None:     # (this kernel has no addend epilogue: arriving second, it adds in a pass of its own)
                gx = gx + ctx.fork.take()
        elif ctx.fork is not None:
            raise _lib.DepthcoreError("GradFork: the shared input needs no gradient")
        if ctx.needs_input_grad[1]:
            with WgradLanes.lane(ctx.param, xx, g_c):
                gw = _grad_dst(ctx.slot, ww)
                ws = torch.empty(L.dc_convs2_wgrad_workspace(B, Ci, Co, Hi, Wi, ks), dtype=torch.uint8, device=xx.device)
                check(L.dc_convs2_wgrad(ptr(xx), ptr(g_c), ptr(gw), ws.data_ptr(), B, Ci, Co, Hi, Wi, ks, stream(xx)), "dc_convs2_wgrad")
        return gx, gw, None


def conv_s2(x, weight, fork=None):
    """F.conv2d(x, weight, None, stride=2, padding=k // 2) for k = 3 or 7.  `fork`: pair GradFork shared with the block's
    1x1 `downsample` (the other consumer of x)."""
    return _ConvS2.apply(x, weight, fork)


# ----------------------------------------------------------------------------------------------
# a1 every other nn.Conv2d shape of the trunk (odd / tiny maps, a stem whose input needs a gradient): plain direct kernels
# ----------------------------------------------------------------------------------------------
class _ConvDirect(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, stride, pad):
        L = _lib.lib()
        xx, ww = _c(x.detach()), _c(weight.detach())
        bs = _c(bias.detach()) if bias is not None else None
        B, Ci, Hi, Wi = xx.shape
        Co, k = ww.shape[0], ww.shape[-1]
        if tuple(ww.shape) != (Co, Ci, k, k) or (bs is not None and bs.numel() != Co):
            raise _lib.DepthcoreError("weight %s / bias do not match an input of %d channels" % (tuple(ww.shape), Ci))
        Ho, Wo = (Hi + 2 * pad - k) // stride + 1, (Wi + 2 * pad - k) // stride + 1
        y = torch.empty(B, Co, max(Ho, 0), max(Wo, 0), dtype=torch.float32, device=xx.device)
        check(L.dc_conv2d_direct_fwd(ptr(xx), ptr(ww), ptr(bs), ptr(y), B, Ci, Co, Hi, Wi, k, int(stride), int(pad), stream(xx)),
              "dc_conv2d_direct_fwd")
        ctx.save_for_backward(xx, ww)
        ctx.cfg = (int(stride), int(pad), bias is not None)
        ctx.slots = (_slot(weight), _slot(bias) if bias is not None else None)
        if ctx.needs_input_grad[1]:
            WgradLanes.count_use(weight)     # (see _Conv3x3)
        return y

    @staticmethod
    def backward(ctx, gy):
        L = _lib.lib()
        xx, ww = ctx.saved_tensors
        s_, p_, has_bias = ctx.cfg
        B, Ci, Hi, Wi = xx.shape
        Co, k = ww.shape[0], ww.shape[-1]
        g_c = _c(gy)
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty_like(xx)
            check(L.dc_conv2d_direct_dgrad(ptr(g_c), ptr(ww), ptr(gx), B, Ci, Co, Hi, Wi, k, s_, p_, stream(xx)), "dc_conv2d_direct_dgrad")
        if ctx.needs_input_grad[1] or (has_bias and ctx.needs_input_grad[2]):
            gw = _grad_dst(ctx.slots[0], ww)
            if has_bias and ctx.needs_input_grad[2]:
                gb = _grad_dst(ctx.slots[1], None)
                if gb is None:
                    gb = torch.empty(Co, dtype=torch.float32, device=xx.device)
            check(L.dc_conv2d_direct_wgrad(ptr(xx), ptr(g_c), ptr(gw), ptr(gb), B, Ci, Co, Hi, Wi, k, s_, p_, stream(xx)),
                  "dc_conv2d_direct_wgrad")
        return gx, gw, gb, None, None


def conv2d_direct(x, weight, bias=None, stride=1, padding=0):
    """F.conv2d(x, weight, bias, stride, padding) for square kernels (k <= 11), stride <= 4, padding < k, no groups / dilation:
    the shapes outside the tiled kernels (slow, correct, deterministic)."""
    return _ConvDirect.apply(x, weight, bias, stride, padding)


# ----------------------------------------------------------------------------------------------
# a1 the stem on the raw frames: (image - mean) / std and the pose pairs' concat inside the kernels' patch loader
#                                   (reference networks/resnet_encoder.py:89-90, trainer.py:398-412)
# ----------------------------------------------------------------------------------------------
def stem_supported(frames, weight):
    f0 = frames[0]
    return (f0.is_cuda and f0.dtype == torch.float32 and len(frames) in (1, 3) and f0.shape[1] == 3
            and tuple(weight.shape[1:]) == (3 * (2 if len(frames) == 3 else 1), 7, 7)
            and all(t.shape == f0.shape and not t.requires_grad for t in frames)
            and bool(_lib.lib().dc_stem_supported(len(frames), f0.shape[0], weight.shape[0], f0.shape[2], f0.shape[3])))


class _StemConv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, weight, mean, std, *frames):
        L = _lib.lib()
        fr = [_c(t.detach()) for t in frames]
        ww = _c(weight.detach())
        nf = len(fr)
        Bf, _, Hi, Wi = fr[0].shape
        Co = ww.shape[0]
        B, Ci = (2 * Bf, 6) if nf == 3 else (Bf, 3)
        ptrs = (ctypes.c_void_p * nf)(*[ptr(t) for t in fr])
        y = torch.empty(B, Co, Hi // 2, Wi // 2, dtype=torch.float32, device=ww.device)
        _use_precision(_lib.PREC_F32)                      # (the stem has no reduced-precision kernel: 3 / 6 input channels)
        ws = torch.empty(L.dc_convs2_fwd_workspace(B, Ci, Co, Hi, Wi, 7), dtype=torch.uint8, device=ww.device)
        check(L.dc_stem_fwd(ptrs, nf, float(mean), float(std), ptr(ww), ptr(y), ws.data_ptr(), Bf, Hi, Wi, Co, stream(ww)),
              "dc_stem_fwd")
        ctx.save_for_backward(ww, *fr)
        ctx.cfg = (float(mean), float(std))
        ctx.slot = _slot(weight)
        ctx.param = _lane_param(ctx, 0, weight)
        return y

    @staticmethod
    def backward(ctx, gy):
        L = _lib.lib()
        ww, *fr = ctx.saved_tensors
        nf = len(fr)
        Bf, _, Hi, Wi = fr[0].shape
        Co = ww.shape[0]
        B, Ci = (2 * Bf, 6) if nf == 3 else (Bf, 3)
        g_c = _c(gy)
        gw = None
        if ctx.needs_input_grad[0]:
            with WgradLanes.lane(ctx.param, g_c):
                gw = _grad_dst(ctx.slot, ww)
                ptrs = (ctypes.c_void_p * nf)(*[ptr(t) for t in fr])
                ws = torch.empty(L.dc_convs2_wgrad_workspace(B, Ci, Co, Hi, Wi, 7), dtype=torch.uint8, device=ww.device)
                check(L.dc_stem_wgrad(ptrs, nf, ctx.cfg[0], ctx.cfg[1], ptr(g_c), ptr(gw), ws.data_ptr(), Bf, Hi, Wi, Co, stream(ww)),
                      "dc_stem_wgrad")
        return (gw, None, None) + (None,) * nf


def stem_conv(frames, weight, mean=0.45, std=0.225):
    """conv2d((x - mean) / std, weight, stride 2, padding 3) with x = frames[0] (one frame) or, for three frames (f-1, f0, f+1),
    the two temporal pairs cat(f-1, f0), cat(f0, f+1) stacked along the batch -- neither the normalised image nor the pair
    tensor is materialised.  Frames are inputs (no gradient)."""
    return _StemConv.apply(weight, mean, std, *frames)


# ----------------------------------------------------------------------------------------------
# f1 ResidualAttentionUnit of the Fusion_v3 front-end (reference networks/fusion_v2.py:46-137)
# ----------------------------------------------------------------------------------------------
PLAIN, PIXEL_SHUFFLE2 = "plain", "ps2"


def _attn_map(tensors, kinds, H, W):
    """dc_attn_map over a list of contiguous source tensors: `plain` (B,c,H,W) contributes c channels, `ps2`
    (B,4,H/2,W/2) one channel read through PixelShuffle(2)."""
    m = _lib.AttnMap()
    c = 0
    for t, k in zip(tensors, kinds):
        if k == PLAIN:
            if t.shape[2] != H or t.shape[3] != W:
                raise _lib.DepthcoreError("attention source %s does not match %dx%d" % (tuple(t.shape), H, W))
            for ch in range(t.shape[1]):
                m.ptr[c] = ptr(t) + ch * H * W * 4
                m.batch_stride[c] = t.shape[1] * H * W
                m.mode[c] = _lib.ATTN_PLAIN
                c += 1
        else:
            if tuple(t.shape[1:]) != (4, H // 2, W // 2) or (H & 1) or (W & 1):
                raise _lib.DepthcoreError("pixel-shuffle source %s does not match %dx%d" % (tuple(t.shape), H, W))
            m.ptr[c] = ptr(t)
            m.batch_stride[c] = 4 * (H // 2) * (W // 2)
            m.mode[c] = _lib.ATTN_PIXEL_SHUFFLE2
            c += 1
    return m, c


def _attn_params(ps):
    """ps: (rel_h, rel_w, wk, bk, wq, bq, wv, bv) in the reference's registration order -> dc_attn_params."""
    a = _lib.AttnParams()
    a.rel_h, a.rel_w, a.wk, a.bk, a.wq, a.bq, a.wv, a.bv = (ptr(t) for t in ps)
    return a


def _attn_param_grads(dp, C, like):
    """dparams vector [dWq, dbq, dWk, dbk, dWv, dbv, drel_h, drel_w] -> gradients in the order/shape of `like`
    (rel_h, rel_w, wk, bk, wq, bq, wv, bv)."""
    cc = C * C
    o = [0, cc, cc + C, 2 * cc + C, 2 * cc + 2 * C, 3 * cc + 2 * C, 3 * cc + 3 * C, 3 * cc + 3 * C + 3, 3 * cc + 3 * C + 6]
    dwq, dbq, dwk, dbk, dwv, dbv, drh, drw = (dp[o[i]:o[i + 1]] for i in range(8))
    out = (drh, drw, dwk, dbk, dwq, dbq, dwv, dbv)
    return tuple(g.reshape(t.shape) for g, t in zip(out, like))


class _ResidualAttentionUnit(torch.autograd.Function):
    """atten2(relu(atten1(relu(u)))) + relu(u) with u gathered from `srcs` (no cat / pixel-shuffle tensors): two fused
    launches forward, two backward (the unit's in-place ReLUs make the skip connection add relu(u), fusion_v2.py:130-136)."""

    @staticmethod
    def forward(ctx, kinds, nsrc, *args):
        L = _lib.lib()
        srcs = [_c(t.detach()) for t in args[:nsrc]]
        p1 = [_c(t.detach()) for t in args[nsrc:nsrc + 8]]
        p2 = [_c(t.detach()) for t in args[nsrc + 8:nsrc + 16]]
        B = srcs[0].shape[0]
        H, W = (srcs[0].shape[2], srcs[0].shape[3]) if kinds[0] == PLAIN else (srcs[0].shape[2] * 2, srcs[0].shape[3] * 2)
        xm, C = _attn_map(srcs, kinds, H, W)
        if C not in (2, 4) or p1[2].shape[0] != C:
            raise _lib.DepthcoreError("AttentionConv kernels cover 2 and 4 channels (got %d sources channels, weights %s)"
                                      % (C, tuple(p1[2].shape)))
        dev = srcs[0].device
        y1 = torch.empty(B, C, H, W, dtype=torch.float32, device=dev)
        y = torch.empty_like(y1)
        a1, a2 = _attn_params(p1), _attn_params(p2)
        st = stream(srcs[0])
        check(L.dc_attnconv_fwd(ctypes.byref(xm), ctypes.byref(a1), None, ptr(y1), B, C, H, W, 1, 0, st), "dc_attnconv_fwd")
        y1m, _ = _attn_map([y1], [PLAIN], H, W)
        check(L.dc_attnconv_fwd(ctypes.byref(y1m), ctypes.byref(a2), ctypes.byref(xm), ptr(y), B, C, H, W, 1, 1, st),
              "dc_attnconv_fwd")
        ctx.save_for_backward(y1, *srcs, *p1, *p2)
        ctx.cfg = (tuple(kinds), nsrc, B, C, H, W)
        return y

    @staticmethod
    def backward(ctx, gy):
        L = _lib.lib()
        kinds, nsrc, B, C, H, W = ctx.cfg
        saved = ctx.saved_tensors
        y1, srcs, p1, p2 = saved[0], list(saved[1:1 + nsrc]), list(saved[1 + nsrc:9 + nsrc]), list(saved[9 + nsrc:17 + nsrc])
        dev = y1.device
        g_c = _c(gy)
        npar = L.dc_attnconv_param_count(C)
        dp1 = torch.empty(npar, dtype=torch.float32, device=dev)
        dp2 = torch.empty(npar, dtype=torch.float32, device=dev)
        ws = torch.empty(L.dc_attnconv_bwd_workspace(B, C, H, W), dtype=torch.uint8, device=dev)
        dy1 = torch.empty_like(y1)
        dskip = torch.empty_like(y1)
        xm, _ = _attn_map(srcs, kinds, H, W)
        y1m, _ = _attn_map([y1], [PLAIN], H, W)
        dy1m, _ = _attn_map([dy1], [PLAIN], H, W)
        dskm, _ = _attn_map([dskip], [PLAIN], H, W)
        a1, a2 = _attn_params(p1), _attn_params(p2)
        st = stream(y1)
        # second AttentionConv: input relu(y1), skip relu(u)
        check(L.dc_attnconv_bwd(ctypes.byref(y1m), ctypes.byref(a2), ctypes.byref(xm), ptr(g_c), ctypes.byref(dy1m), None,
                                ctypes.byref(dskm), ptr(dp2), ws.data_ptr(), B, C, H, W, 1, 1, st), "dc_attnconv_bwd")
        # first AttentionConv: input relu(u); its input gradient + the skip gradient go straight to the sources' gradients
        dsrcs = [torch.empty_like(t) for t in srcs]
        dm, _ = _attn_map(dsrcs, kinds, H, W)
        check(L.dc_attnconv_bwd(ctypes.byref(xm), ctypes.byref(a1), None, ptr(dy1), ctypes.byref(dm), ptr(dskip), None,
                                ptr(dp1), ws.data_ptr(), B, C, H, W, 1, 0, st), "dc_attnconv_bwd")
        return (None, None, *dsrcs, *_attn_param_grads(dp1, C, p1), *_attn_param_grads(dp2, C, p2))


def residual_attention_unit(srcs, kinds, params1, params2):
    """srcs / kinds: the unit's input channels as a list of tensors (`PLAIN` (B,c,H,W) or `PIXEL_SHUFFLE2` (B,4,H/2,W/2));
    params1 / params2: (rel_h, rel_w, key w, key b, query w, query b, value w, value b) of atten1 / atten2."""
    return _ResidualAttentionUnit.apply(tuple(kinds), len(srcs), *srcs, *params1, *params2)


# ----------------------------------------------------------------------------------------------
# f2 ConvGRU cell gate arithmetic + sequence residual        (reference networks/rnn.py:101-143, trainer_gru.py:637-639)
# ----------------------------------------------------------------------------------------------
class _GruRH(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gates, h):
        L = _lib.lib()
        gt, hh = _c(gates.detach()), _c(h.detach())
        B, C = hh.shape[0], hh.shape[1]
        P = hh.shape[2] * hh.shape[3]
        if tuple(gt.shape) != (B, 2 * C) + tuple(hh.shape[2:]):
            raise _lib.DepthcoreError("gates %s do not match state %s" % (tuple(gt.shape), tuple(hh.shape)))
        rh = torch.empty_like(hh)
        check(L.dc_gru_rh_fwd(ptr(gt), ptr(hh), ptr(rh), B, C, P, stream(hh)), "dc_gru_rh_fwd")
        ctx.save_for_backward(gt, hh)
        return rh

    @staticmethod
    def backward(ctx, g):
        L = _lib.lib()
        gt, hh = ctx.saved_tensors
        B, C = hh.shape[0], hh.shape[1]
        P = hh.shape[2] * hh.shape[3]
        g_c = _c(g)
        dg, dh = torch.empty_like(gt), torch.empty_like(hh)
        check(L.dc_gru_rh_bwd(ptr(gt), ptr(hh), ptr(g_c), ptr(dg), ptr(dh), B, C, P, stream(hh)), "dc_gru_rh_bwd")
        return dg, dh


class _GruBlend(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gates, h, cnm):
        L = _lib.lib()
        gt, hh, cc = _c(gates.detach()), _c(h.detach()), _c(cnm.detach())
        B, C = hh.shape[0], hh.shape[1]
        P = hh.shape[2] * hh.shape[3]
        if tuple(gt.shape) != (B, 2 * C) + tuple(hh.shape[2:]) or cc.shape != hh.shape:
            raise _lib.DepthcoreError("gates %s / candidate %s do not match state %s" % (tuple(gt.shape), tuple(cc.shape), tuple(hh.shape)))
        out = torch.empty_like(hh)
        check(L.dc_gru_blend_fwd(ptr(gt), ptr(hh), ptr(cc), ptr(out), B, C, P, stream(hh)), "dc_gru_blend_fwd")
        ctx.save_for_backward(gt, hh, cc)
        return out

    @staticmethod
    def backward(ctx, g):
        L = _lib.lib()
        gt, hh, cc = ctx.saved_tensors
        B, C = hh.shape[0], hh.shape[1]
        P = hh.shape[2] * hh.shape[3]
        g_c = _c(g)
        dg, dh, dc = torch.empty_like(gt), torch.empty_like(hh), torch.empty_like(cc)
        check(L.dc_gru_blend_bwd(ptr(gt), ptr(hh), ptr(cc), ptr(g_c), ptr(dg), ptr(dh), ptr(dc), B, C, P, stream(hh)), "dc_gru_blend_bwd")
        return dg, dh, dc


class _GruResidual(torch.autograd.Function):
    @staticmethod
    def forward(ctx, f, H):
        L = _lib.lib()
        ff, HH = _c(f.detach()), _c(H.detach())
        n = ff.shape[0]
        M = ff[0].numel()
        if tuple(HH.shape) != (n + 1,) + tuple(ff.shape[1:]):
            raise _lib.DepthcoreError("hidden states %s must be (n+1, ...) for features %s" % (tuple(HH.shape), tuple(ff.shape)))
        out = torch.empty_like(ff)
        check(L.dc_gru_residual_fwd(ptr(ff), ptr(HH), ptr(out), n, M, stream(ff)), "dc_gru_residual_fwd")
        ctx.dims = (n, M, HH.shape)
        return out

    @staticmethod
    def backward(ctx, g):
        L = _lib.lib()
        n, M, hshape = ctx.dims
        g_c = _c(g)
        dH = torch.empty(hshape, dtype=torch.float32, device=g.device)
        check(L.dc_gru_residual_bwd(ptr(g_c), ptr(dH), n, M, stream(g_c)), "dc_gru_residual_bwd")
        return g_c, dH


def stack_frames(groups):
    """[[t_0, t_1, ...], ...] -> [torch.cat(group, 0) for group in groups] in ONE launch (dc_gather_copy): the sequence trainer's
    stacking of the frames of a sequence along the batch (trainer_gru.py:819-821, 886-896, 943-944).  float32 device tensors
    without a gradient (inputs of the step); a group's members share their trailing shape."""
    L = _lib.lib()
    outs, src, dst, n, keep = [], [], [], [], []
    for g in groups:
        ts = [_c(t.detach()) for t in g]
        if any(t.dtype != torch.float32 or not t.is_cuda or t.shape[1:] != ts[0].shape[1:] for t in ts):
            raise _lib.DepthcoreError("stack_frames takes float32 device tensors that agree in their trailing shape")
        out = torch.empty((sum(t.shape[0] for t in ts),) + tuple(ts[0].shape[1:]), dtype=torch.float32, device=ts[0].device)
        off = 0
        for t in ts:
            src.append(t.data_ptr()); dst.append(out.data_ptr() + 4 * off); n.append(t.numel())
            off += t.numel()
        keep.append(ts)
        outs.append(out)
    for i in range(0, len(src), 96):
        k = len(src[i:i + 96])
        check(L.dc_gather_copy((ctypes.c_void_p * k)(*src[i:i + 96]), (ctypes.c_void_p * k)(*dst[i:i + 96]),
                               (ctypes.c_size_t * k)(*n[i:i + 96]), k, stream(outs[0])), "dc_gather_copy")
    return outs


def gru_reset_times_state(gates, h):
    """r * h with r = gates[:, :C]."""
    return _GruRH.apply(gates, h)


def gru_blend(gates, h, cnm):
    """(1 - u) * h + u * cnm with u = gates[:, C:]."""
    return _GruBlend.apply(gates, h, cnm)


def gru_sequence_residual(features, hidden_states):
    """features (n,C,H,W) + (H[1:] + H[:-1]) / 2 with H (n+1,C,H,W)."""
    return _GruResidual.apply(features, hidden_states)


class _GruLevel(torch.autograd.Function):
    """One feature level of `run_gru_v5` for a sequence of n frames (batch_size 1, trainer_gru.py:607-639) as ONE autograd node:
    the cell runs over the frames from the learned initial state, every tensor of every step lives in one buffer per kind
    (gates / r*h / candidate / the n+1 hidden states -- a step reads and writes its slot by pointer, so there is neither a
    per-frame slice node nor a torch.cat of the trace), and the backward walks the frames in reverse itself: each step's
    gradient of h_i is ACCUMULATED into the slot that already holds the residual's and the later step's share
    (dc_gru_blend_bwd_acc, dc_gru_rh_bwd_acc, the gate convolution's data-gradient store with the slot as its own addend), the
    feature gradient leaves the gate convolution's store with the candidate convolution's and the residual's share added on the
    way, and the shared parameters get their gradients from ONE weight-gradient pass per convolution over the n frames as a
    batch once the walk is done (per-frame passes + a sum over frames otherwise).  Against the per-op
    graph (ConvGRUCell.forward, kept for single calls) that is ~9 elementwise sums, 3 fills / copies and 1.7 concatenations
    fewer per cell step: 137 + 47 + 25 launches of a C4 step."""

    @staticmethod
    def forward(ctx, feats, h0, wg, bg, wc, bc):
        L = _lib.lib()
        ff = _c(feats.detach())
        n, C, H, W = ff.shape
        P = H * W
        if tuple(h0.shape) != (1, C, H, W) or tuple(wg.shape) != (2 * C, 2 * C, 3, 3) or tuple(wc.shape) != (C, 2 * C, 3, 3):
            raise _lib.DepthcoreError("ConvGRU level: state %s / gate weights %s / candidate weights %s do not match features %s" % (
                tuple(h0.shape), tuple(wg.shape), tuple(wc.shape), tuple(ff.shape)))
        w_g, b_g, w_c, b_c = _c(wg.detach()), _c(bg.detach()), _c(wc.detach()), _c(bc.detach())
        dev = ff.device
        Hs = torch.empty(n + 1, C, H, W, dtype=torch.float32, device=dev)
        Hs[0].copy_(h0.detach()[0])
        gates = torch.empty(n, 2 * C, H, W, dtype=torch.float32, device=dev)
        rh = torch.empty(n, C, H, W, dtype=torch.float32, device=dev)
        cnm = torch.empty(n, C, H, W, dtype=torch.float32, device=dev)
        ctx.prec = _use_precision(_precision[0])
        ws = torch.empty(max(L.dc_conv3x3_fwd_workspace(C, C, 1, 2 * C, H, W), L.dc_conv3x3_fwd_workspace(C, C, 1, C, H, W)),
                         dtype=torch.uint8, device=dev)
        st = stream(ff)
        for i in range(n):
            x, h = ff[i:i + 1], Hs[i:i + 1]
            # [reset | update] = sigmoid(conv_gates(cat(x, h)))                                       rnn.py:125-130
            check(L.dc_conv3x3_fwd(ptr(x), C, 0, ptr(h), C, ptr(w_g), ptr(b_g), ptr(gates[i:i + 1]), ws.data_ptr(), 1, 2 * C, H, W,
                                   ACT_SIGMOID, PAD_ZERO, st), "dc_conv3x3_fwd")
            check(L.dc_gru_rh_fwd(ptr(gates[i:i + 1]), ptr(h), ptr(rh[i:i + 1]), 1, C, P, st), "dc_gru_rh_fwd")
            # cnm = tanh(conv_can(cat(x, reset * h)))                                                   rnn.py:132-134
            check(L.dc_conv3x3_fwd(ptr(x), C, 0, ptr(rh[i:i + 1]), C, ptr(w_c), ptr(b_c), ptr(cnm[i:i + 1]), ws.data_ptr(), 1, C, H, W,
                                   ACT_TANH, PAD_ZERO, st), "dc_conv3x3_fwd")
            check(L.dc_gru_blend_fwd(ptr(gates[i:i + 1]), ptr(h), ptr(cnm[i:i + 1]), ptr(Hs[i + 1:i + 2]), 1, C, P, st),
                  "dc_gru_blend_fwd")                                                                # rnn.py:136
        out = torch.empty_like(ff)
        check(L.dc_gru_residual_fwd(ptr(ff), ptr(Hs), ptr(out), n, C * P, st), "dc_gru_residual_fwd")   # trainer_gru.py:637-639
        ctx.save_for_backward(ff, Hs, gates, rh, cnm, w_g, w_c)
        ctx.slots = tuple(_slot(t) for t in (wg, bg, wc, bc))
        for t, k in ((wg, 2), (wc, 4)):
            if ctx.needs_input_grad[k]:
                WgradLanes.count_use(t)
        return out

    @staticmethod
    def backward(ctx, g):
        L = _lib.lib()
        ff, Hs, gates, rh, cnm, w_g, w_c = ctx.saved_tensors
        n, C, H, W = ff.shape
        P = H * W
        dev = ff.device
        need = ctx.needs_input_grad
        g_c = _c(g)
        st = stream(ff)
        f32 = dict(dtype=torch.float32, device=dev)
        dHs = torch.empty_like(Hs)
        check(L.dc_gru_residual_bwd(ptr(g_c), ptr(dHs), n, C * P, st), "dc_gru_residual_bwd")
        dfe = torch.empty_like(ff)
        dgates, dcn = torch.empty(n, 2 * C, H, W, **f32), torch.empty(n, C, H, W, **f32)     # g of the two convolutions' outputs, per frame
        dxc, drh = torch.empty(1, C, H, W, **f32), torch.empty(1, C, H, W, **f32)
        _use_precision(ctx.prec)
        ws = torch.empty(max(L.dc_conv3x3_bwd_workspace(C, C, b, co, H, W) for b in (1, n) for co in (C, 2 * C)),
                         dtype=torch.uint8, device=dev)
        for i in range(n - 1, -1, -1):
            x, h, gt, dh = ff[i:i + 1], Hs[i:i + 1], gates[i:i + 1], dHs[i:i + 1]
            dg, dc = dgates[i:i + 1], dcn[i:i + 1]
            # h_{i+1} = (1-u) h_i + u cnm: dHs[i+1] is complete here (residual + step i+1)
            check(L.dc_gru_blend_bwd_acc(ptr(gt), ptr(h), ptr(cnm[i:i + 1]), ptr(dHs[i + 1:i + 2]), ptr(dg), ptr(dh), ptr(dc),
                                         1, C, P, st), "dc_gru_blend_bwd_acc")
            # candidate convolution over cat(x, r*h), data gradients only: that of x takes the residual's share (g[i]) along
            check(L.dc_conv3x3_bwd_add(ptr(x), C, 0, ptr(rh[i:i + 1]), C, ptr(w_c), ptr(cnm[i:i + 1]), ptr(dc), ptr(dxc), ptr(drh),
                                       ptr(g_c[i:i + 1]), None, None, None, ws.data_ptr(), 1, C, H, W, ACT_TANH, PAD_ZERO, st),
                  "dc_conv3x3_bwd_add")
            check(L.dc_gru_rh_bwd_acc(ptr(gt), ptr(h), ptr(drh), ptr(dg), ptr(dh), 1, C, P, st), "dc_gru_rh_bwd_acc")
            # gate convolution over cat(x, h): d x = own + (candidate's + residual's), d h_i = own + what the slot holds
            check(L.dc_conv3x3_bwd_add(ptr(x), C, 0, ptr(h), C, ptr(w_g), ptr(gt), ptr(dg), ptr(dfe[i:i + 1]), ptr(dh),
                                       ptr(dxc), ptr(dh), None, None, ws.data_ptr(), 1, 2 * C, H, W, ACT_SIGMOID, PAD_ZERO, st),
                  "dc_conv3x3_bwd_add")
        # the parameters are shared by the n steps and nothing downstream waits for their gradients: ONE weight-gradient pass
        # per convolution with the n frames as its batch (sum over frames = sum over the batch), after the walk
        outs = [None] * 4
        for k, (like, x1, y, gy, co, act) in enumerate(((w_g, Hs, gates, dgates, 2 * C, ACT_SIGMOID), (w_c, rh, cnm, dcn, C, ACT_TANH))):
            if not (need[2 + 2 * k] or need[3 + 2 * k]):
                continue
            dw = _grad_dst(ctx.slots[2 * k], like) if need[2 + 2 * k] else None
            db = None
            if need[3 + 2 * k]:
                db = _grad_dst(ctx.slots[2 * k + 1], None)
                if db is None:
                    db = torch.empty(co, **f32)
            check(L.dc_conv3x3_bwd_add(ptr(ff), C, 0, ptr(x1), C, ptr(like), ptr(y), ptr(gy), None, None, None, None, ptr(dw), ptr(db),
                                       ws.data_ptr(), n, co, H, W, act, PAD_ZERO, st), "dc_conv3x3_bwd_add")
            outs[2 * k], outs[2 * k + 1] = dw, db
        return (dfe if need[0] else None, dHs[0:1] if need[1] else None) + tuple(outs)


def gru_level_sequence(features, h0, conv_gates, conv_can):
    """features (n,C,H,W) of one sequence, h0 (1,C,H,W) -> features + (H[1:] + H[:-1]) / 2, H the hidden states of the cell
    (conv_gates / conv_can: its two nn.Conv2d) run over the frames in order (trainer_gru.py:607-639 at one level)."""
    return _GruLevel.apply(features, h0, conv_gates.weight, conv_gates.bias, conv_can.weight, conv_can.bias)
