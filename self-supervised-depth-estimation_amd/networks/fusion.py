"""Fusion_v3 attention-fusion front-end with the reference's constructor / forward / state_dict layout
(reference networks/fusion_v2.py:46-137, 226-236, 279-363; caller trainer_fusion_v3.py:319-330).

Module names follow the reference (`fusion_block_{1..4}.resConfUnit{1,2,3}.atten{1,2}.{rel_h,rel_w,key_conv,query_conv,
value_conv}`, `.conv_1`, `.conv3x3.conv`, `.upscale.conv`: 210 tensors, 1,672 parameters), so checkpoints interchange.
The arithmetic runs on depthcore kernels: each ResidualAttentionUnit is two fused AttentionConv launches whose input
channels are gathered straight from their producers (the `torch.cat`s and the PixelShuffle of the reference never exist in
memory), the three tiny 3x3 convolutions of a block run on the fused conv block (bias + tanh in the epilogue).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F
import torch.nn.init as init

from depthcore import ops as _ops
from depthcore._lib import DepthcoreError
from layers import Conv3x3


class AttentionConv(nn.Module):
    """Parameters of networks/fusion_v2.py:46-65 (kernel 3, stride 1, padding 1, groups 1); evaluated by its unit."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, padding=1, groups=1, bias=False):
        super().__init__()
        if (kernel_size, stride, padding, groups) != (3, 1, 1, 1) or in_channels != out_channels:
            raise NotImplementedError("AttentionConv kernels cover the Fusion_v3 instantiation (k3 s1 p1 g1, C -> C)")
        self.out_channels = out_channels
        self.rel_h = nn.Parameter(torch.randn(1, 1, 1, kernel_size, 1), requires_grad=True)
        self.rel_w = nn.Parameter(torch.randn(1, 1, 1, 1, kernel_size), requires_grad=True)
        self.key_conv = nn.Conv2d(in_channels, out_channels, kernel_size=1, bias=bias)
        self.query_conv = nn.Conv2d(in_channels, out_channels, kernel_size=1, bias=bias)
        self.value_conv = nn.Conv2d(in_channels, out_channels, kernel_size=1, bias=bias)
        for c in (self.key_conv, self.value_conv, self.query_conv):           # fusion_v2.py:92-98
            init.kaiming_normal_(c.weight, mode="fan_out", nonlinearity="relu")
        init.normal_(self.rel_h, 0, 1)
        init.normal_(self.rel_w, 0, 1)

    def params(self):
        return (self.rel_h, self.rel_w, self.key_conv.weight, self.key_conv.bias, self.query_conv.weight,
                self.query_conv.bias, self.value_conv.weight, self.value_conv.bias)


class ResidualAttentionUnit(nn.Module):
    """networks/fusion_v2.py:101-137: atten2(relu(atten1(relu(x)))) + relu(x) (the in-place ReLU rewrites the skip input)."""

    def __init__(self, features):
        super().__init__()
        self.atten1 = AttentionConv(features, features, kernel_size=3, stride=1, padding=1, bias=True)
        self.atten2 = AttentionConv(features, features, kernel_size=3, stride=1, padding=1, bias=True)
        self.relu = nn.ReLU(inplace=True)

    def forward(self, srcs, kinds=None):
        if isinstance(srcs, torch.Tensor):
            srcs = [srcs]
        if not srcs[0].is_cuda:
            raise DepthcoreError("Fusion_v3 runs on depthcore kernels only; there is no CPU path")
        kinds = kinds or [_ops.PLAIN] * len(srcs)
        return _ops.residual_attention_unit(srcs, kinds, self.atten1.params(), self.atten2.params())


class ResidualConvUnit(nn.Module):
    """networks/fusion_v2.py:11-43 (`Fusion_v3(attention=False)`, reference option --disable_attention): conv2(relu(conv1(
    relu(x)))) + relu(x) -- the in-place ReLU rewrites the unit's input here as well.  Both 3x3 convolutions (zero padding,
    bias, ReLU of the first in its epilogue) run on the fused conv block; the unit's input is assembled with stock tensor
    ops (this variant is not on a BASELINE configuration: its 2-4 channel concatenations are not worth a gather kernel)."""

    def __init__(self, features):
        super().__init__()
        self.conv1 = nn.Conv2d(features, features, kernel_size=3, stride=1, padding=1, bias=True)
        self.conv2 = nn.Conv2d(features, features, kernel_size=3, stride=1, padding=1, bias=True)
        self.relu = nn.ReLU(inplace=True)

    def forward(self, srcs, kinds=None):
        if isinstance(srcs, torch.Tensor):
            srcs = [srcs]
        if not srcs[0].is_cuda:
            raise DepthcoreError("Fusion_v3 runs on depthcore kernels only; there is no CPU path")
        kinds = kinds or [_ops.PLAIN] * len(srcs)
        parts = [t if k == _ops.PLAIN else F.pixel_shuffle(t, 2) for t, k in zip(srcs, kinds)]
        r = torch.relu(parts[0] if len(parts) == 1 else torch.cat(parts, 1))
        c1, c2 = self.conv1, self.conv2
        h = _ops.conv3x3_block(r, None, c1.weight, c1.bias, False, _ops.ACT_RELU, _ops.PAD_ZERO)
        return _ops.conv3x3_block(h, None, c2.weight, c2.bias, False, _ops.ACT_NONE, _ops.PAD_ZERO) + r


class UpscalePS(nn.Module):
    """networks/fusion_v2.py:226-236.  `forward` returns the PRE-shuffle tensor tanh(conv(x)) (B, out*scale^2, h, w); its
    consumer reads it through the PixelShuffle index map (depthcore.ops.PIXEL_SHUFFLE2)."""

    def __init__(self, input_ch, output_ch, scale):
        super().__init__()
        if scale != 2 or output_ch != 1:
            raise NotImplementedError("UpscalePS kernels cover the Fusion_v3 instantiation (1 output channel, scale 2)")
        self.conv = nn.Conv2d(input_ch, output_ch * scale ** 2, kernel_size=3, stride=1, padding=1, bias=True)
        self.ps = nn.PixelShuffle(scale)

    def forward(self, x):
        return _ops.conv3x3_block(x, None, self.conv.weight, self.conv.bias, False, _ops.ACT_TANH, _ops.PAD_ZERO)


class FeatureFusionBlock_v3(nn.Module):
    """networks/fusion_v2.py:279-320."""

    def __init__(self, features, attention=True, init_scale=False):
        super().__init__()
        self.init_scale = init_scale
        if self.init_scale:
            self.conv_1 = nn.Conv2d(1, 2, kernel_size=3, stride=1, padding=1, bias=True)
        unit = ResidualAttentionUnit if attention else ResidualConvUnit           # fusion_v2.py:294-302
        self.resConfUnit1 = unit(features)
        self.resConfUnit2 = unit(features)
        self.resConfUnit3 = unit(features * 2)
        self.conv3x3 = Conv3x3(features * 2, 1)
        self.upscale = UpscalePS(features * 2, 1, 2)

    def forward(self, dt, upt, dt_1, dt_2, need_up=True):
        """upt: the previous block's PRE-shuffle upscale output (B,4,h/2,w/2), or None for the first block."""
        if self.init_scale:
            c = self.conv_1
            first = _ops.conv3x3_block(dt, None, c.weight, c.bias, False, _ops.ACT_NONE, _ops.PAD_ZERO)
            a = self.resConfUnit1([first])
        else:
            a = self.resConfUnit1([dt, upt], [_ops.PLAIN, _ops.PIXEL_SHUFFLE2])
        b = self.resConfUnit2([dt_1, dt_2])
        out = self.resConfUnit3([a, b])
        output_depth = self.conv3x3(out)
        output_up = self.upscale(out) if need_up else None
        return output_depth, output_up


class Fusion_v3(nn.Module):
    """networks/fusion_v2.py:323-363.  Input: the depth decoder's outputs on the 3B stacked frames; chunk 0 of every
    `("disp", s)` is fused with chunks 1 and 2 as context (the caller stacks frames [-2, -1, 0]: chunk 0 is frame -2 --
    reproduced as is).  Outputs `("disp", s)` at batch B, no sigmoid."""

    def __init__(self, attention=True):
        super().__init__()
        self.fusion_block_1 = FeatureFusionBlock_v3(features=2, attention=attention, init_scale=True)
        self.fusion_block_2 = FeatureFusionBlock_v3(features=2, attention=attention)
        self.fusion_block_3 = FeatureFusionBlock_v3(features=2, attention=attention)
        self.fusion_block_4 = FeatureFusionBlock_v3(features=2, attention=attention)

    def forward(self, depth_dec_outputs):
        outputs, up = {}, None
        blocks = (self.fusion_block_1, self.fusion_block_2, self.fusion_block_3, self.fusion_block_4)
        for blk, s in zip(blocks, (3, 2, 1, 0)):
            v = depth_dec_outputs[("disp", s)]
            cur, t1, t2 = v.split(len(v) // 3)
            # block 4's upscale output feeds nothing (the reference computes and drops it): skipped
            outputs[("disp", s)], up = blk(cur, up, t1, t2, need_up=(s != 0))
        return outputs
