"""BatchNorm folded into the neighbouring convolutions of the ResNet trunks (include/depthcore.h: dc_bn_fold).

torchvision's BasicBlock / Bottleneck (reference networks/resnet_encoder.py:87-98) put a training-mode BatchNorm2d (+ReLU, and
on the block output + the skip) behind every convolution.  As stand-alone kernels that is four streaming passes over HBM per
layer (statistics, apply, backward statistics, backward apply).  Here:

  * the PRODUCING convolution emits the per-channel partial sums of its raw output in its store epilogue (`BNStats`), a
    finalize launch turns them into mean / invstd / scale / shift (and the running statistics);
  * a BatchNorm + ReLU with ONE consumer (bn1 -> conv2, bn2 -> conv3) is never materialised: the consuming convolution reads
    relu(scale * x + shift) in its loader (forward and weight gradient), its data-gradient epilogue applies the ReLU decision
    and emits the backward partials, and one pass forms dx (`FoldedConv*`);
  * a block output relu(bn(x) + skip) keeps its apply pass (`bn_apply`), and when the next block's conv1 is its only consumer
    (`BNLink`) that convolution's data-gradient epilogue -- which already adds the skip's gradient -- does the masking and the
    backward partials, so the backward is one pass as well.

Same arithmetic as depthcore.ops.bn_relu up to the order of the sums; deterministic.  No fallback: CPU tensors raise.
"""
import ctypes

import torch

from . import _lib
from . import ops as _ops
from ._lib import BnFold, check, ptr, stream

_c = _ops._c


class BNStats:
    """Partial {sum, sum of squares} of a raw convolution output, per channel (dc_bn_fold.stat_part)."""
    __slots__ = ("part", "nparts", "ppg", "groups")

    def __init__(self, part, nparts, ppg, groups):
        self.part, self.nparts, self.ppg, self.groups = part, nparts, ppg, groups


class BNLink:
    """Left on a block output y = relu(bn(x) + skip) by `bn_apply`: what the output's ONLY consumer (the next block: conv1 +
    skip, joined by a GradFork) needs to do the first pass of this BatchNorm's backward in its data-gradient epilogue."""
    __slots__ = ("x", "mean", "mask", "groups", "bwd", "armed")

    def __init__(self, x, mean, mask, groups):
        self.x, self.mean, self.mask, self.groups = x, mean, mask, groups
        self.bwd = None          # (part, nparts, ppg): set by the consumer's backward -- the gradient it returned is g', masked
        self.armed = False       # a consumer has taken the link in its forward


def _ival():
    return ctypes.c_int(0)


def _finalize(L, xx, stats, gamma, beta, rm, rv, eps, momentum, groups):
    """BNStats (or a stand-alone statistics pass when the producer had no epilogue) -> tab (4, groups*C): mean, invstd, scale, shift."""
    N, C, H, W = xx.shape
    if stats is None:
        ppg = _ival()
        nparts = L.dc_bn_stat_parts(N, C, H * W, groups, ctypes.byref(ppg))
        if not nparts:
            raise _lib.DepthcoreError("BatchNorm fold: no statistics layout for %s with %d groups" % (tuple(xx.shape), groups))
        part = torch.empty(C * nparts * 2, dtype=torch.float32, device=xx.device)
        check(L.dc_bn_stats(ptr(xx), ptr(part), N, C, H * W, groups, stream(xx)), "dc_bn_stats")
        stats = BNStats(part, nparts, ppg.value, groups)
    if stats.groups != groups:
        raise _lib.DepthcoreError("BatchNorm fold: statistics were taken for %d groups, the layer has %d" % (stats.groups, groups))
    tab = torch.empty(4, groups * C, dtype=torch.float32, device=xx.device)
    check(L.dc_bn_finalize(ptr(stats.part), stats.nparts, stats.ppg, float(N // groups) * H * W, ptr(gamma), ptr(beta), ptr(rm), ptr(rv),
                           tab[0].data_ptr(), tab[1].data_ptr(), tab[2].data_ptr(), tab[3].data_ptr(), C, groups, float(eps),
                           float(momentum), stream(xx)), "dc_bn_finalize")
    return tab


def _bwd_finish(L, xx, gp, bwd, gamma, tab, groups, slots):
    """backward partials -> coefficients, dgamma, dbeta; dx = a g' + b (x - mean) + c0 in one pass."""
    N, C, H, W = xx.shape
    part, nparts, ppg = bwd
    coef = torch.empty(groups * C * 4, dtype=torch.float32, device=xx.device)
    dgamma = _ops._grad_dst(slots[0], gamma)
    dbeta = _ops._grad_dst(slots[1], gamma)
    check(L.dc_bn_bwd_finalize(ptr(part), nparts, ppg, float(N // groups) * H * W, ptr(gamma), tab[0].data_ptr(), tab[1].data_ptr(),
                               ptr(coef), ptr(dgamma), ptr(dbeta), C, groups, stream(xx)), "dc_bn_bwd_finalize")
    dx = torch.empty_like(xx)
    check(L.dc_bn_bwd_apply(ptr(xx), ptr(gp), ptr(coef), ptr(dx), N, C, H * W, groups, stream(xx)), "dc_bn_bwd_apply")
    return dx, dgamma, dbeta


class _Cfg:
    """non-tensor arguments of the folded ops (one object, so that the autograd signatures stay short)"""
    __slots__ = ("kind", "stride", "groups", "fork", "skip", "in_stats", "bn", "prev", "want_stats", "relu", "res_fork", "leave_link",
                 "out_stats", "out_link")      # the last two are written by the forward (read by the wrapper right after apply)

    def __init__(self, **kw):
        for k in self.__slots__:
            setattr(self, k, kw.get(k))


def _bn_buffers(bn):
    mom = 0.1 if bn.momentum is None else bn.momentum
    return bn.running_mean, bn.running_var, bn.eps, mom


# ----------------------------------------------------------------------------------------------
# convolution with the fold: [relu(bn(x)) in the loader] -> conv -> [statistics epilogue]; the backward's data gradient carries
# the BatchNorm-backward epilogue.  kind "g1": 1x1 on the tiled GEMMs (dc_conv1x1_*_bn); "wino": stride-1 3x3 (dc_wino3x3_*_bn)
# ----------------------------------------------------------------------------------------------
class _K1:
    """1x1 (stride 1 / 2) on the tiled GEMM kernels: fp32-MFMA (csrc/gemm1x1.hip) or, under dc_set_gemm_split, the split-operand
    kernels (csrc/gemm1x1_x3.hip) where they take the shape -- decided per pass, by the SAME predicate in the partial-count query
    and in the launch (the epilogues' partial layouts follow each kernel family's own tiles)"""
    @staticmethod
    def out_hw(Hi, Wi, s_):
        return Hi // s_, Wi // s_

    @staticmethod
    def _x3_fwd(L, B, Ci, Co, Hi, Wi, s_, groups):
        return bool(L.dc_get_gemm_split()) and L.dc_gemm1x1x3_stat_parts(B, Ci, Co, Hi, Wi, s_, groups, None) > 0

    @staticmethod
    def _x3_dgrad(L, B, Ci, Co, Hi, Wi, s_, groups):
        return bool(L.dc_get_gemm_split()) and s_ == 1 and L.dc_gemm1x1x3_bwd_parts(B, Ci, Co, Hi, Wi, groups, None) > 0

    @staticmethod
    def stat_parts(L, B, Ci, Co, Hi, Wi, s_, groups, ppg):
        if _K1._x3_fwd(L, B, Ci, Co, Hi, Wi, s_, groups):
            return L.dc_gemm1x1x3_stat_parts(B, Ci, Co, Hi, Wi, s_, groups, ppg)
        return L.dc_conv1x1_stat_parts(B, Ci, Co, Hi, Wi, s_, groups, ppg)

    @staticmethod
    def bwd_parts(L, B, Ci, Co, Hi, Wi, s_, groups, ppg):
        if _K1._x3_dgrad(L, B, Ci, Co, Hi, Wi, s_, groups):
            return L.dc_gemm1x1x3_bwd_parts(B, Ci, Co, Hi, Wi, groups, ppg)
        return L.dc_conv1x1_bwd_parts(B, Ci, Co, Hi, Wi, groups, ppg) if s_ == 1 else 0

    @staticmethod
    def fwd(L, xx, ww, y, B, Ci, Co, Hi, Wi, s_, f):
        if _K1._x3_fwd(L, B, Ci, Co, Hi, Wi, s_, f.groups):
            ws = torch.empty(L.dc_gemm1x1x3_workspace(Ci, Co), dtype=torch.uint8, device=xx.device)
            check(L.dc_gemm1x1x3_fwd_bn(ptr(xx), ptr(ww), None, ptr(y), ws.data_ptr(), B, Ci, Co, Hi, Wi, s_, 0, ctypes.byref(f), stream(xx)),
                  "dc_gemm1x1x3_fwd_bn")
            return
        check(L.dc_conv1x1_fwd_bn(ptr(xx), ptr(ww), ptr(y), B, Ci, Co, Hi, Wi, s_, ctypes.byref(f), stream(xx)), "dc_conv1x1_fwd_bn")

    @staticmethod
    def dgrad(L, g_c, ww, gx, add, B, Ci, Co, Hi, Wi, s_, f, add2=None):
        if f.bwd_part:            # BatchNorm-backward epilogue (stride 1): the kernel family the partials were sized for
            if add2 is not None:
                raise _lib.DepthcoreError("the BatchNorm epilogues take one addend")
            if _K1._x3_dgrad(L, B, Ci, Co, Hi, Wi, s_, f.groups):
                ws = torch.empty(L.dc_gemm1x1x3_workspace(Ci, Co), dtype=torch.uint8, device=gx.device)
                check(L.dc_gemm1x1x3_dgrad_bn(ptr(g_c), ptr(ww), ptr(gx), ws.data_ptr(), ptr(add), None, B, Ci, Co, Hi, Wi, s_, ctypes.byref(f),
                                              stream(gx)), "dc_gemm1x1x3_dgrad_bn")
            else:
                check(L.dc_conv1x1_dgrad_bn(ptr(g_c), ptr(ww), ptr(gx), ptr(add), B, Ci, Co, Hi, Wi, s_, ctypes.byref(f), stream(gx)),
                      "dc_conv1x1_dgrad_bn")
            return
        # plain data gradient (+ the addends of the input's other consumers; stride 2: the cell-block scatter)
        if L.dc_get_gemm_split() and L.dc_gemm1x1x3_dgrad_ok(B, Ci, Co, Hi, Wi, s_):
            ws = torch.empty(L.dc_gemm1x1x3_workspace(Ci, Co), dtype=torch.uint8, device=gx.device)
            check(L.dc_gemm1x1x3_dgrad(ptr(g_c), ptr(ww), ptr(gx), ws.data_ptr(), ptr(add), ptr(add2), B, Ci, Co, Hi, Wi, s_, stream(gx)),
                  "dc_gemm1x1x3_dgrad")
        else:
            check(L.dc_conv1x1_dgrad_add2(ptr(g_c), ptr(ww), ptr(gx), ptr(add), ptr(add2), B, Ci, Co, Hi, Wi, s_, stream(gx)),
                  "dc_conv1x1_dgrad_add2")

    @staticmethod
    def wgrad(L, xx, g_c, gw, B, Ci, Co, Hi, Wi, s_, f):
        if L.dc_get_gemm_split() and L.dc_gemm1x1x3_wgrad_ok(B, Ci, Co, Hi, Wi, s_):
            ws = torch.empty(L.dc_gemm1x1x3_wgrad_workspace(B, Ci, Co, Hi, Wi, s_), dtype=torch.uint8, device=xx.device)
            check(L.dc_gemm1x1x3_wgrad_bn(ptr(xx), ptr(g_c), ptr(gw), ws.data_ptr(), B, Ci, Co, Hi, Wi, s_, ctypes.byref(f), stream(xx)),
                  "dc_gemm1x1x3_wgrad_bn")
            return
        ws = torch.empty(L.dc_conv1x1_wgrad_workspace(B, Ci, Co, Hi, Wi, s_), dtype=torch.uint8, device=xx.device)
        check(L.dc_conv1x1_wgrad_bn(ptr(xx), ptr(g_c), ptr(gw), ws.data_ptr(), B, Ci, Co, Hi, Wi, s_, ctypes.byref(f), stream(xx)),
              "dc_conv1x1_wgrad_bn")


class _KW:
    """stride-1 3x3, zero padding 1: fused Winograd on the fp32 matrix cores"""
    @staticmethod
    def out_hw(Hi, Wi, s_):
        return Hi, Wi

    @staticmethod
    def stat_parts(L, B, Ci, Co, Hi, Wi, s_, groups, ppg):
        return L.dc_wino3x3_stat_parts(B, Ci, Co, Hi, Wi, groups, ppg)

    @staticmethod
    def bwd_parts(L, B, Ci, Co, Hi, Wi, s_, groups, ppg):
        return L.dc_wino3x3_bwd_parts(B, Ci, Co, Hi, Wi, groups, ppg)

    @staticmethod
    def fwd(L, xx, ww, y, B, Ci, Co, Hi, Wi, s_, f):
        ws = torch.empty(L.dc_wino3x3_workspace(B, Ci, Co, Hi, Wi), dtype=torch.uint8, device=xx.device)
        check(L.dc_wino3x3_fwd_bn(ptr(xx), ptr(ww), ptr(y), ws.data_ptr(), B, Ci, Co, Hi, Wi, ctypes.byref(f), stream(xx)), "dc_wino3x3_fwd_bn")

    @staticmethod
    def dgrad(L, g_c, ww, gx, add, B, Ci, Co, Hi, Wi, s_, f, add2=None):
        if add2 is not None:
            raise _lib.DepthcoreError("SkipSum on a 3x3 convolution")
        ws = torch.empty(L.dc_wino3x3_workspace(B, Ci, Co, Hi, Wi), dtype=torch.uint8, device=gx.device)
        check(L.dc_wino3x3_dgrad_bn(ptr(g_c), ptr(ww), ptr(gx), ptr(add), ws.data_ptr(), B, Ci, Co, Hi, Wi, ctypes.byref(f), stream(gx)),
              "dc_wino3x3_dgrad_bn")

    @staticmethod
    def wgrad(L, xx, g_c, gw, B, Ci, Co, Hi, Wi, s_, f):
        ws = torch.empty(L.dc_wino3x3_wgrad_workspace(B, Ci, Co, Hi, Wi), dtype=torch.uint8, device=xx.device)
        check(L.dc_wino3x3_wgrad_bn(ptr(xx), ptr(g_c), ptr(gw), ws.data_ptr(), B, Ci, Co, Hi, Wi, ctypes.byref(f), stream(xx)),
              "dc_wino3x3_wgrad_bn")


_KINDS = {"g1": _K1, "wino": _KW}


class _ConvF(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, gamma, beta, cfg):
        L = _lib.lib()
        K = _KINDS[cfg.kind]
        xx, ww = _c(x.detach()), _c(weight.detach())
        B, Ci, Hi, Wi = xx.shape
        Co, s_, groups = ww.shape[0], cfg.stride, cfg.groups
        if ww.shape[1] != Ci:
            raise _lib.DepthcoreError("weight %s does not match %d input channels" % (tuple(ww.shape), Ci))
        if cfg.kind == "wino":
            _ops._use_precision(_lib.PREC_F32)       # (the fold is an fp32-policy path; the callers check the policy)
        fold_in = gamma is not None
        f = BnFold()
        f.groups = groups
        tab = g = None
        if fold_in:
            g = _c(gamma.detach())
            tab = _finalize(L, xx, cfg.in_stats, g, _c(beta.detach()), *_bn_buffers(cfg.bn), groups)
            f.in_scale, f.in_shift = tab[2].data_ptr(), tab[3].data_ptr()
            _ops._record_kink("relu", lambda: _materialize(xx, tab, groups))
        Ho, Wo = K.out_hw(Hi, Wi, s_)
        y = torch.empty(B, Co, Ho, Wo, dtype=torch.float32, device=xx.device)
        if cfg.want_stats:
            ppg = _ival()
            nparts = K.stat_parts(L, B, Ci, Co, Hi, Wi, s_, groups, ctypes.byref(ppg))
            if nparts:
                part = torch.empty(Co * nparts * 2, dtype=torch.float32, device=xx.device)
                f.stat_part = part.data_ptr()
                cfg.out_stats = BNStats(part, nparts, ppg.value, groups)
        K.fwd(L, xx, ww, y, B, Ci, Co, Hi, Wi, s_, f)
        prev = cfg.prev if (cfg.prev is not None and x.requires_grad and cfg.fork is not None and not cfg.fork.pair) else None
        if prev is not None:
            prev.armed = True
        ctx.save_for_backward(xx, ww, g, tab, prev.x if prev is not None else None, prev.mean if prev is not None else None,
                              prev.mask if prev is not None else None)
        ctx.cfg = (s_, groups, fold_in, cfg.kind)
        ctx.prev = prev
        ctx.slots = (_ops._slot(weight), _ops._slot(gamma) if fold_in else None, _ops._slot(beta) if fold_in else None)
        ctx.param = _ops._lane_param(ctx, 1, weight)
        ctx.fork = cfg.fork if x.requires_grad else None
        # x's SkipSum (ops.SkipSum): the member of a pair fork that runs second collects the secondary consumers' gradients
        ctx.skip = cfg.skip if (cfg.skip is not None and cfg.kind == "g1" and ctx.fork is not None and ctx.fork.pair and not fold_in) else None
        if ctx.skip is not None:
            ctx.skip.arm()
        return y

    @staticmethod
    def backward(ctx, gy):
        L = _lib.lib()
        xx, ww, g, tab, px, pmean, pmask = ctx.saved_tensors
        B, Ci, Hi, Wi = xx.shape
        s_, groups, fold_in, kind = ctx.cfg
        K = _KINDS[kind]
        Co = ww.shape[0]
        g_c = _c(gy)
        if kind == "wino":
            _ops._use_precision(_lib.PREC_F32)
        gx = gw = dgamma = dbeta = None
        fork = ctx.fork
        first_of_pair = fork is not None and fork.pair and fork.arrive() == 0
        add = None if first_of_pair else _ops._fork_addend(ctx, xx)
        if ctx.prev is not None and add is None:
            raise _lib.DepthcoreError("BNLink: the block's skip gradient did not arrive -- the convolution's result would not be "
                                      "the complete gradient of the linked output")
        f = BnFold()
        f.groups = groups
        bwd = None
        # (the folded BatchNorm's dgamma / dbeta come out of the data-gradient epilogue's partials: that launch runs whenever
        # they are wanted, even if the raw input itself needs no gradient -- a frozen producer)
        bn_only = fold_in and not ctx.needs_input_grad[0] and (ctx.needs_input_grad[2] or ctx.needs_input_grad[3])
        if ctx.needs_input_grad[0] or bn_only:
            gx = torch.empty_like(xx)
            if fold_in or ctx.prev is not None:
                ppg = _ival()
                nparts = K.bwd_parts(L, B, Ci, Co, Hi, Wi, s_, groups, ctypes.byref(ppg))
                if not nparts:
                    raise _lib.DepthcoreError("BatchNorm fold: no backward epilogue for %s" % (tuple(xx.shape),))
                bpart = torch.empty(Ci * nparts * 2, dtype=torch.float32, device=xx.device)
                bwd = (bpart, nparts, ppg.value)
                f.bwd_part = bpart.data_ptr()
                if fold_in:
                    f.bn_x, f.bn_mean = xx.data_ptr(), tab[0].data_ptr()
                    f.in_scale, f.in_shift = tab[2].data_ptr(), tab[3].data_ptr()
                else:
                    f.bn_x, f.bn_mean, f.bn_mask = px.data_ptr(), pmean.data_ptr(), pmask.data_ptr()
            add2 = ctx.skip.take() if (ctx.skip is not None and not first_of_pair) else None
            if add2 is not None and (add2.shape != xx.shape or not add2.is_contiguous()):
                raise _lib.DepthcoreError("SkipSum: gradient %s does not match the input %s" % (tuple(add2.shape), tuple(xx.shape)))
            K.dgrad(L, g_c, ww, gx, add, B, Ci, Co, Hi, Wi, s_, f, add2)
            if first_of_pair:
                fork.park(gx)
                gx = None
            elif ctx.prev is not None:
                ctx.prev.bwd = bwd              # gx is g' of the producing block's last BatchNorm: its backward is one pass now
        elif first_of_pair:
            raise _lib.DepthcoreError("GradFork: the shared input needs no gradient")
        if ctx.needs_input_grad[1]:
            fw = BnFold()
            fw.groups = groups
            reads = [xx, g_c]
            if fold_in:
                fw.in_scale, fw.in_shift = tab[2].data_ptr(), tab[3].data_ptr()
                reads.append(tab)
            with _ops.WgradLanes.lane(ctx.param, *reads):
                gw = _ops._grad_dst(ctx.slots[0], ww)
                K.wgrad(L, xx, g_c, gw, B, Ci, Co, Hi, Wi, s_, fw)
        if fold_in and gx is not None:
            gx, dgamma, dbeta = _bwd_finish(L, xx, gx, bwd, g, tab, groups, ctx.slots[1:])
            if bn_only:
                gx = None
        return gx, gw, dgamma, dbeta, None


def _materialize(xx, tab, groups):
    """relu(scale x + shift) as a tensor -- decision observers (tests) only; the training path never forms it."""
    N, C = xx.shape[:2]
    sc = tab[2].view(groups, 1, C, 1, 1)
    sh = tab[3].view(groups, 1, C, 1, 1)
    return torch.relu(xx.view(groups, N // groups, C, *xx.shape[2:]) * sc + sh).view_as(xx)


def conv1x1_ok(conv, xshape, groups):
    """All three passes of this stride-1 1x1 convolution on an fp32 device tensor of shape `xshape` take the fold with a
    BatchNorm on its input (tiled kernels, 16-byte staging, backward epilogue for this group layout)."""
    B, Ci, Hi, Wi = xshape
    L = _lib.lib()
    return (conv.kernel_size == (1, 1) and conv.stride == (1, 1) and conv.padding == (0, 0) and conv.groups == 1 and conv.bias is None
            and conv.in_channels == Ci and B % groups == 0 and (Hi * Wi) % 4 == 0
            and bool(L.dc_conv1x1_bn_ok(B, Ci, conv.out_channels, Hi, Wi))
            and L.dc_conv1x1_bwd_parts(B, Ci, conv.out_channels, Hi, Wi, groups, None) > 0)


def wino_ok(conv, xshape, groups):
    """Forward and data gradient of this stride-1 3x3 convolution take the fold with a BatchNorm on its input."""
    B, Ci, Hi, Wi = xshape
    _ops._use_precision(_lib.PREC_F32)      # (the query below looks at the calling thread's policy; the callers run the fp32 one)
    return (conv.kernel_size == (3, 3) and conv.stride == (1, 1) and conv.padding == (1, 1) and conv.dilation == (1, 1)
            and conv.groups == 1 and conv.bias is None and conv.in_channels == Ci and Wi % 2 == 0 and (Hi * Wi) % 4 == 0
            and max(B * Ci * Hi * Wi, B * conv.out_channels * Hi * Wi) * 4 < 0x7fffffff
            and bool(_lib.lib().dc_wino3x3_bn_ok(B, Ci, conv.out_channels, Hi, Wi, groups)))


def _conv(kind, x, weight, stride, groups, fork, in_bn, in_stats, prev, want_stats, skip=None):
    cfg = _Cfg(kind=kind, stride=int(stride), groups=int(groups), fork=fork, skip=skip, in_stats=in_stats, bn=in_bn, prev=prev,
               want_stats=want_stats)
    y = _ConvF.apply(x, weight, in_bn.weight if in_bn is not None else None, in_bn.bias if in_bn is not None else None, cfg)
    return y, cfg.out_stats


def conv1x1(x, weight, stride=1, groups=1, fork=None, in_bn=None, in_stats=None, prev=None, want_stats=True, skip=None):
    """y_raw, BNStats-or-None = conv1x1([relu(in_bn(x))]).  `in_bn`: the nn.BatchNorm2d (training mode) in front of this convolution
    whose ReLU-ed output has no other consumer -- x is then that BatchNorm's RAW input and `in_stats` its statistics partials
    (None: a stand-alone statistics pass).  `prev`: BNLink of the block output x.  `fork`: GradFork."""
    return _conv("g1", x, weight, stride, groups, fork, in_bn, in_stats, prev, want_stats, skip)


def conv3x3(x, weight, groups=1, fork=None, in_bn=None, in_stats=None, prev=None, want_stats=True):
    """The same for F.conv2d(x, weight, None, 1, 1) on the Winograd kernels (even width)."""
    return _conv("wino", x, weight, 1, groups, fork, in_bn, in_stats, prev, want_stats)


# ----------------------------------------------------------------------------------------------
# the apply pass that remains: block outputs y = relu?(bn(x) [+ skip]) with several consumers
# ----------------------------------------------------------------------------------------------
class _BNApply(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, res, gamma, beta, cfg):
        L = _lib.lib()
        xx = _c(x.detach())
        rr = _c(res.detach()) if res is not None else None
        g = _c(gamma.detach())
        N, C, H, W = xx.shape
        groups, relu = cfg.groups, cfg.relu
        tab = _finalize(L, xx, cfg.in_stats, g, _c(beta.detach()), *_bn_buffers(cfg.bn), groups)
        y = torch.empty_like(xx)
        nmask = L.dc_bn_mask_bytes(N, C, H * W) if relu else 0
        mask = torch.empty(nmask, dtype=torch.uint8, device=xx.device) if nmask else None
        check(L.dc_bn_apply(ptr(xx), ptr(rr), tab[2].data_ptr(), tab[3].data_ptr(), ptr(y), mask.data_ptr() if nmask else None,
                            N, C, H * W, int(relu), groups, stream(xx)), "dc_bn_apply")
        if relu:
            _ops._record_kink("relu", y)
        ctx.save_for_backward(xx, y if (relu and mask is None) else None, g, tab, mask)
        ctx.cfg = (int(relu), res is not None, groups)
        ctx.slots = (_ops._slot(gamma), _ops._slot(beta))
        ctx.fork = cfg.res_fork if (cfg.res_fork is not None and res is not None and res.requires_grad) else None
        ctx.link = cfg.out_link = BNLink(xx, tab[0], mask, groups) if (cfg.leave_link and relu and mask is not None) else None
        return y

    @staticmethod
    def backward(ctx, gy):
        L = _lib.lib()
        xx, y, g, tab, mask = ctx.saved_tensors
        relu, has_res, groups = ctx.cfg
        N, C, H, W = xx.shape
        g_c = _c(gy)
        link = ctx.link
        want_res = has_res and ctx.needs_input_grad[1]
        if link is not None and link.bwd is not None:
            # the only consumer's data-gradient epilogue masked the gradient and took the partials: one pass left; the masked
            # gradient itself is the skip's gradient
            dx, dgamma, dbeta = _bwd_finish(L, xx, g_c, link.bwd, g, tab, groups, ctx.slots)
            dres = g_c if want_res else None
            link.bwd = None
        else:
            if link is not None and link.armed:
                raise _lib.DepthcoreError("BNLink: the consumer took the link but its backward has not run before this BatchNorm's")
            dx = torch.empty_like(xx)
            dres = torch.empty_like(xx) if want_res else None
            dgamma = _ops._grad_dst(ctx.slots[0], g)
            dbeta = _ops._grad_dst(ctx.slots[1], g)
            ws = torch.empty(L.dc_bn_workspace(N, C, H * W), dtype=torch.uint8, device=xx.device)
            check(L.dc_bn_relu_bwd(ptr(xx), ptr(y), ptr(g_c), ptr(g), tab[0].data_ptr(), tab[1].data_ptr(), ptr(dx), ptr(dres),
                                   ptr(dgamma), ptr(dbeta), ws.data_ptr(), mask.data_ptr() if mask is not None else None, N, C, H * W,
                                   relu, groups, stream(xx)), "dc_bn_relu_bwd")
        if ctx.fork is not None and dres is not None:
            ctx.fork.park(dres)
            dres = None
        return dx, dres, dgamma, dbeta, None


def bn_apply(x, bn, stats=None, res=None, relu=True, groups=1, fork=None, leave_link=False):
    """y = relu?(bn(x) [+ res]) from the producing convolution's statistics partials (`stats`; None: a stand-alone statistics
    pass).  `leave_link`: attach a BNLink to y for a consumer that is known to be the only one (ResNetTrunk sets it for block
    outputs inside a stage)."""
    cfg = _Cfg(groups=int(groups), in_stats=stats, bn=bn, relu=bool(relu), res_fork=fork, leave_link=bool(leave_link))
    y = _BNApply.apply(x, res, bn.weight, bn.bias, cfg)
    if cfg.out_link is not None and y.requires_grad:
        y._dc_bn_link = cfg.out_link
    return y


def take_link(x, conv, groups):
    """The BNLink of a block output for its only consumer `conv` (the caller vouches for "only"), or None when there is none
    or the convolution's data gradient has no BatchNorm epilogue on this shape."""
    link = getattr(x, "_dc_bn_link", None)
    if link is None or link.groups != groups:
        return None
    B, Ci, Hi, Wi = x.shape
    if (Hi * Wi) % 4:
        ok = False
    elif conv.kernel_size == (1, 1):
        ok = conv.stride == (1, 1) and _lib.lib().dc_conv1x1_bwd_parts(B, Ci, conv.out_channels, Hi, Wi, groups, None) > 0
    elif conv.kernel_size == (3, 3):
        ok = (conv.stride == (1, 1) and conv.padding == (1, 1) and Wi % 2 == 0
              and _lib.lib().dc_wino3x3_bwd_parts(B, Ci, conv.out_channels, Hi, Wi, groups, None) > 0)
    else:
        ok = False
    return link if ok else None
