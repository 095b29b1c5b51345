"""Drop-in for the reference's `layers.py` (same public names and call signatures), backed by
libdepthcore.so -- hand-written gfx950 kernels behind the C ABI of include/depthcore.h.

    from layers import *        # as trainer.py:24 and networks/depth_decoder.py:14 do

Every function / module cites the reference lines it mirrors.  There is no CPU or eager fallback:
a tensor that is not on the GPU, or a missing libdepthcore.so, raises.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from depthcore import ops as _ops


def disp_to_depth(disp, min_depth, max_depth):
    """layers.py:16-25 -> (scaled_disp, depth)."""
    return _ops.disp_to_depth(disp, min_depth, max_depth)


def transformation_from_parameters(axisangle, translation, invert=False):
    """layers.py:28-45: axisangle, translation (B,1,3) -> (B,4,4)."""
    return _ops.pose_matrix(axisangle, translation, invert)


def get_translation_matrix(translation_vector):
    """layers.py:48-61."""
    return _ops.pose_matrix(torch.zeros_like(translation_vector), translation_vector, False)


def rot_from_axisangle(vec):
    """layers.py:64-103 (input Bx1x3)."""
    return _ops.pose_matrix(vec, torch.zeros_like(vec), False)


class Conv3x3(nn.Module):
    """layers.py:119-136: pad (reflection or zero) + 3x3 conv, as one dc_conv3x3 launch.  Sub-modules `.pad`,
    `.conv` are kept as in the reference so state_dict keys match (`conv.weight`, `conv.bias`); `.conv` only
    holds the parameters.  (The same arithmetic as separate library launches, for timing comparisons, lives in
    tools/time_decoder.py -- not in this module.)"""

    def __init__(self, in_channels, out_channels, use_refl=True):
        super().__init__()
        self.pad = nn.ReflectionPad2d(1) if use_refl else nn.ZeroPad2d(1)
        self.conv = nn.Conv2d(int(in_channels), int(out_channels), 3)
        self._pad_mode = _ops.PAD_REFLECT if use_refl else _ops.PAD_ZERO

    def forward(self, x, skip=None, up=False, act=_ops.ACT_NONE, fork=None):
        """`fork`: a pair ops.GradFork shared with the other fused block that reads x (DepthDecoder)."""
        return _ops.conv3x3_block(x, skip, self.conv.weight, self.conv.bias, up, act, self._pad_mode, fork)


class ConvBlock(nn.Module):
    """layers.py:106-116: Conv3x3 + ELU (fused).  `forward(x, skip, up=True)` additionally fuses the decoder's
    `upsample(x)` + `torch.cat([.., skip], 1)` (networks/depth_decoder.py:56-60) into the same launch."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.conv = Conv3x3(in_channels, out_channels)
        self.nonlin = nn.ELU(inplace=True)

    def forward(self, x, skip=None, up=False, fork=None):
        return self.conv(x, skip, up, _ops.ACT_ELU, fork)


class BackprojectDepth(nn.Module):
    """layers.py:139-168.  Buffers `id_coords`, `ones`, `pix_coords` are kept (non-trainable
    Parameters, as in the reference) for API / checkpoint compatibility; `pix_coords` is written by
    the dc_pix_coords kernel (bit-exact integers) when the module is moved to the GPU."""

    def __init__(self, batch_size, height, width):
        super().__init__()
        self.batch_size, self.height, self.width = batch_size, height, width
        ys, xs = np.divmod(np.arange(height * width, dtype=np.int64), width)
        idc = np.stack([xs.reshape(height, width), ys.reshape(height, width)], 0).astype(np.float32)
        self.id_coords = nn.Parameter(torch.from_numpy(idc), requires_grad=False)
        self.ones = nn.Parameter(torch.ones(batch_size, 1, height * width), requires_grad=False)
        pc = torch.from_numpy(np.stack([xs, ys, np.ones_like(xs)], 0).astype(np.float32))
        self.pix_coords = nn.Parameter(pc.unsqueeze(0).repeat(batch_size, 1, 1), requires_grad=False)

    def _apply(self, fn, *a, **k):
        super()._apply(fn, *a, **k)
        if self.pix_coords.is_cuda:
            self.pix_coords.data = _ops.pix_coords(self.batch_size, self.height, self.width, self.pix_coords.device)
        return self

    def forward(self, depth, inv_K):
        return _ops.backproject(depth, inv_K)


class Project3D(nn.Module):
    """layers.py:171-193."""

    def __init__(self, batch_size, height, width, eps=1e-7):
        super().__init__()
        self.batch_size, self.height, self.width, self.eps = batch_size, height, width, eps

    def forward(self, points, K, T):
        return _ops.project3d(points, K, T, self.height, self.width, self.eps)


def upsample(x):
    """layers.py:196-199: nearest x2 (dc_upsample_nearest2x_*).  The depth decoder does not call it: its upsample is folded
    into the next convolution's loader (ConvBlock.forward(up=True))."""
    return _ops.upsample_nearest2x(x)


def get_smooth_loss(disp, img):
    """layers.py:202-215."""
    return _ops.smooth_loss(disp, img)


class SSIM(nn.Module):
    """layers.py:218-248."""

    def __init__(self):
        super().__init__()
        self.C1, self.C2 = 0.01 ** 2, 0.03 ** 2

    def forward(self, x, y):
        return _ops.ssim(x, y)


def grid_sample(img, grid, padding_mode="border", align_corners=False):
    """The `F.grid_sample(img, grid, padding_mode="border")` call of trainer.py:508-511."""
    if padding_mode != "border":
        raise NotImplementedError("only padding_mode='border' is on the hot path")
    return _ops.grid_sample_border(img, grid, align_corners)


def interpolate_bilinear(x, size):
    """The `F.interpolate(disp, [H, W], mode="bilinear", align_corners=False)` call of trainer.py:474."""
    return _ops.upsample_bilinear(x, int(size[0]), int(size[1]))


def compute_depth_errors(gt, pred):
    """layers.py:251-269 (monitoring only; plain torch on whatever device the inputs live)."""
    thresh = torch.max(gt / pred, pred / gt)
    a1 = (thresh < 1.25).float().mean()
    a2 = (thresh < 1.25 ** 2).float().mean()
    a3 = (thresh < 1.25 ** 3).float().mean()
    rmse = torch.sqrt(((gt - pred) ** 2).mean())
    rmse_log = torch.sqrt(((torch.log(gt) - torch.log(pred)) ** 2).mean())
    abs_rel = torch.mean(torch.abs(gt - pred) / gt)
    sq_rel = torch.mean((gt - pred) ** 2 / gt)
    return abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3
