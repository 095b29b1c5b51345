"""torch.autograd bindings of the C ABI in include/depthcore.h.

Every op enqueues hand-written gfx950 kernels on the current HIP stream; tensors
are allocated by PyTorch's caching allocator and passed down as raw pointers.
"""
import ctypes

import torch

from . import _lib
from ._lib import PhotoDesc, check, ptr, stream


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


# ----------------------------------------------------------------------------------------------
# fused photometric loss  (reference trainer.py:465-622)
# ----------------------------------------------------------------------------------------------
class PhotoConfig:
    """Non-differentiable inputs + options of one fused photometric step.

    target/src: inputs[("color", 0|-1|+1, 0)];  color_s[s]: inputs[("color", 0, s)];
    K / inv_K: inputs[("K"|"inv_K", 0)];  noise[s]: the tie-break randn of trainer.py:594-595
    (None -> on-device counter RNG).  `materialize` asks for the log tensors of
    generate_images_pred (depth / sample / color / identity_selection).
    """

    def __init__(self, target, src_m1, src_p1, color_s, K, inv_K, noise=None, min_depth=0.1, max_depth=100.0,
                 smoothness=1e-3, disable_automasking=False, avg_reprojection=False, no_ssim=False,
                 align_corners=False, materialize=False, rng_seed=0):
        self.target, self.src = _c(target), (_c(src_m1), _c(src_p1))
        self.color_s = [_c(c) for c in color_s]
        self.K, self.inv_K = _c(K), _c(inv_K)
        self.noise = None if noise is None else [_c(n) for n in noise]
        self.min_depth, self.max_depth, self.smoothness = float(min_depth), float(max_depth), float(smoothness)
        self.flags = ((_lib.OPT_NO_AUTOMASK if disable_automasking else 0)
                      | (_lib.OPT_AVG_REPROJ if avg_reprojection else 0)
                      | (_lib.OPT_NO_SSIM if no_ssim else 0)
                      | (_lib.OPT_ALIGN_CORNERS if align_corners else 0))
        self.materialize = materialize
        self.rng_seed = int(rng_seed)
        self.extras = {}          # filled by forward: argmin maps + optional log tensors


def _fill_desc(cfg, T0, T1, disps):
    B, _, H, W = cfg.target.shape
    ns = len(disps)
    if ns < 1 or ns > _lib.MAX_SCALES:
        raise _lib.DepthcoreError("1..4 scales supported, got %d" % ns)
    if cfg.target.shape[1] != 3:
        raise _lib.DepthcoreError("images must be (B,3,H,W)")
    d = PhotoDesc()
    d.B, d.H, d.W, d.num_scales = B, H, W, ns
    d.flags = cfg.flags
    d.min_depth, d.max_depth, d.smoothness = cfg.min_depth, cfg.max_depth, cfg.smoothness
    d.target = ptr(cfg.target)
    for f in range(2):
        if cfg.src[f].shape != cfg.target.shape:
            raise _lib.DepthcoreError("source / target shape mismatch")
        d.source[f] = ptr(cfg.src[f])
    for t in (cfg.K, cfg.inv_K, T0, T1):
        if tuple(t.shape) != (B, 4, 4):
            raise _lib.DepthcoreError("K / inv_K / T must be (B,4,4), got %s" % (tuple(t.shape),))
    d.K, d.inv_K = ptr(cfg.K), ptr(cfg.inv_K)
    d.T[0], d.T[1] = ptr(T0), ptr(T1)
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
    return d


class _PhotoLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cfg, T0, T1, *disps):
        L = _lib.lib()
        T0, T1 = _c(T0.detach()), _c(T1.detach())
        disps = [_c(x.detach()) for x in disps]
        dev = cfg.target.device
        B, _, H, W = cfg.target.shape
        ns = len(disps)
        d = _fill_desc(cfg, T0, T1, disps)
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
        check(L.dc_photo_fwd(ctypes.byref(d), stream()), "dc_photo_fwd")
        cfg.extras = ex
        ctx.cfg, ctx.ws, ctx.argmin = cfg, ws, argmin
        ctx.save_for_backward(T0, T1, *disps)
        return losses

    @staticmethod
    def backward(ctx, g_losses):
        L = _lib.lib()
        cfg = ctx.cfg
        T0, T1, *disps = ctx.saved_tensors
        d = _fill_desc(cfg, T0, T1, disps)
        d.workspace, d.workspace_bytes = ctx.ws.data_ptr(), ctx.ws.numel()
        g = _c(g_losses.to(torch.float32))
        d.g_losses = ptr(g)
        d_disp = [torch.empty_like(x) for x in disps]
        dT = [torch.empty_like(T0), torch.empty_like(T1)]
        for s in range(len(disps)):
            d.argmin[s] = ctx.argmin[s].data_ptr()
            d.d_disp[s] = ptr(d_disp[s])
        d.d_T[0], d.d_T[1] = ptr(dT[0]), ptr(dT[1])
        check(L.dc_photo_bwd(ctypes.byref(d), stream()), "dc_photo_bwd")
        return (None, dT[0], dT[1], *d_disp)


def photometric_loss(cfg, T_m1, T_p1, disps):
    """-> losses tensor (num_scales+1,): [loss/0, ..., loss]  (trainer.py:618-621)."""
    return _PhotoLoss.apply(cfg, T_m1, T_p1, *disps)


def photo_algorithmic_bytes(cfg, T0, T1, disps, backward):
    d = _fill_desc(cfg, _c(T0.detach()), _c(T1.detach()), [_c(x.detach()) for x in disps])
    return _lib.lib().dc_photo_algorithmic_bytes(ctypes.byref(d), int(backward))
