"""ResnetEncoder with the reference's constructor / forward / state_dict layout
(reference networks/resnet_encoder.py:17-98), without torchvision.

The reference delegates the trunk to `torchvision.models.resnet*`; torchvision is not a
dependency here, so the ResNet v1.5 trunk is stated directly (same module names, hence the same
state_dict keys: `encoder.conv1.weight`, `encoder.layer1.0.bn1.running_mean`, ..., `encoder.fc.*`).
Convolutions go through `conv_impl` so that the MFMA implicit-GEMM kernels can be swapped in
without touching the module tree.
"""
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from depthcore import bnfold as _bnf
from depthcore import ops as _ops


def _bn_act(x, bn, res=None, relu=True, groups=1, fork=None):
    """BatchNorm2d (+ residual) (+ ReLU).  Training mode (the hot path) is one fused depthcore launch chain and has no
    fallback: a CPU tensor raises DepthcoreError.  Eval mode (running statistics; validation / export, not on the
    training path) uses the stock functional ops.
    `groups`: number of independent sub-batches stacked along dim 0 (statistics per sub-batch)."""
    if bn.training:
        return _ops.bn_relu(x, bn, res, relu, groups, fork)
    y = F.batch_norm(x, bn.running_mean, bn.running_var, bn.weight, bn.bias, False, bn.momentum or 0.1, bn.eps)
    if res is not None:
        y = y + res
    return F.relu(y) if relu else y


CONV_S2 = True        # 7x7 / 2 stem and 3x3 / 2 convolutions on depthcore's implicit-GEMM kernels (GPU)
GEMM_1X1 = True       # 1x1 stride-2 `downsample` convolutions on depthcore's NCHW MFMA GEMM (GPU)
WINO_TRUNK = True     # stride-1 3x3 trunk convolutions on depthcore's fused Winograd kernel (GPU, even widths)
STEM_FUSED = True     # input normalisation (and the pose pairs' concat) inside the stem kernels' loader (dc_stem_*)


# BatchNorm passes folded into the neighbouring convolutions (depthcore.bnfold; training mode on the GPU).  DC_BN_FOLD (same-box
# A/Bs): 0 = the stand-alone BatchNorm kernels; 1 = statistics from the producing convolution's epilogue only; 2 = + the block
# outputs' backward statistics in the next block's data-gradient epilogue (BNLink); 3 = + BatchNorm + ReLU with one consumer
# folded into that convolution's loader.  Unset = "auto" (-1): level 3 for the Bottleneck trunks (resnet50+: -3.0 % of a C3 step,
# 49.28 -> 47.81 ms same box), level 0 for the BasicBlock trunks (resnet18 at C2: 11.63 / 11.64 ms without, 11.60-11.69 ms at
# levels 1-3 -- the BatchNorm passes there are 6-11 us each and hide behind the other stream's convolutions)
BN_FOLD = int(os.environ.get("DC_BN_FOLD", "-1"))


def _fold_level(bottleneck):
    return BN_FOLD if BN_FOLD >= 0 else (3 if bottleneck else 0)


GRAD_FORK = True      # residual blocks without a downsample branch: the skip's gradient is added inside conv1's data-gradient kernel
SKIP_SUMS = os.environ.get("DC_GRAD_SUMS", "1") != "0"   # (DC_GRAD_SUMS=0: autograd's elementwise sums, same-box A/Bs) feature maps carry an ops.SkipSum: the decoder's skip-connection gradient is added by the map's primary consumer


def _fork_for(block, x):
    """A GradFork for this call of a residual block, or None: training on the GPU, identity skip, an input that needs a
    gradient, conv1 on a kernel with the addend epilogue (stride-1 3x3 Winograd or 1x1 GEMM) and a map below 2 GiB."""
    c = block.conv1
    if not (GRAD_FORK and block.downsample is None and block.training and x.is_cuda and x.requires_grad and x.dtype == torch.float32
            and torch.is_grad_enabled() and c.stride == (1, 1) and c.groups == 1 and c.bias is None and x.numel() * 4 < 0x7fffffff):
        return None
    if c.kernel_size == (3, 3) and WINO_TRUNK and c.padding == (1, 1) and c.dilation == (1, 1) and x.shape[-1] % 2 == 0:
        return _ops.GradFork()
    if c.kernel_size == (1, 1) and GEMM_1X1 and c.padding == (0, 0):
        return _ops.GradFork()
    return None


def _pair_fork_for(block, x):
    """A pair GradFork for a Bottleneck with a downsample branch (both consumers of x are 1x1 GEMM convolutions), or None."""
    c, d = block.conv1, block.downsample[0]
    ok = (GRAD_FORK and GEMM_1X1 and block.training and x.is_cuda and x.requires_grad and x.dtype == torch.float32
          and torch.is_grad_enabled() and x.numel() * 4 < 0x7fffffff
          and all(m.kernel_size == (1, 1) and m.padding == (0, 0) and m.groups == 1 and m.bias is None for m in (c, d))
          and c.stride == (1, 1) and d.stride in ((1, 1), (2, 2)) and (d.stride == (1, 1) or (x.shape[-1] % 2 == 0 and x.shape[-2] % 2 == 0)))
    return _ops.GradFork(pair=True) if ok else None


def _basic_pair_fork(block, x):
    """A pair GradFork for a BasicBlock with a downsample branch (3x3 / 2 conv1 on dc_convs2_*, 1x1 / 2 `downsample` on the tiled
    GEMM whose data gradient takes addends), or None."""
    c, d = block.conv1, block.downsample[0]
    B, Ci, H, W = x.shape
    ok = (GRAD_FORK and GEMM_1X1 and CONV_S2 and block.training and x.is_cuda and x.requires_grad and x.dtype == torch.float32
          and torch.is_grad_enabled() and x.numel() * 4 < 0x7fffffff and _ops._precision[0] == _ops.PRECISIONS["f32"]
          and c.kernel_size == (3, 3) and c.stride == (2, 2) and c.padding == (1, 1) and c.groups == 1 and c.bias is None
          and d.kernel_size == (1, 1) and d.stride == (2, 2) and d.padding == (0, 0) and d.groups == 1 and d.bias is None
          and H % 2 == 0 and W % 2 == 0 and c.in_channels % 4 == 0 and c.out_channels % 32 == 0 and _ops.conv_s2_supported(x, c.weight))
    return _ops.GradFork(pair=True) if ok else None


def _conv(conv, x, fork=None, skip=None):
    """nn.Conv2d call of the trunk: stride-1 3x3 on dc_wino3x3_* (84 % of a ResNet-18 trunk's multiplies), 1x1 on dc_conv1x1_*,
    the 7x7 / 2 stem and the 3x3 / 2 convolutions on dc_convs2_*; shapes outside those kernels' 16-byte staging (odd or
    tiny maps in tests) take dc_conv2d_direct_*.  On the GPU nothing reaches the framework's convolution."""
    if (WINO_TRUNK and x.is_cuda and conv.kernel_size == (3, 3) and conv.stride == (1, 1) and conv.padding == (1, 1)
            and conv.dilation == (1, 1) and conv.groups == 1 and conv.bias is None and x.shape[-1] % 2 == 0
            and x.dtype == torch.float32):
        return _ops.wino_conv3x3(x, conv.weight, fork)
    if (GEMM_1X1 and x.is_cuda and conv.kernel_size == (1, 1) and conv.padding == (0, 0) and conv.groups == 1
            and conv.bias is None and conv.stride in ((1, 1), (2, 2)) and x.dtype == torch.float32
            and (conv.stride == (1, 1) or (x.shape[-1] % 2 == 0 and x.shape[-2] % 2 == 0))):
        # Bottleneck conv1 / conv3 and every `downsample` branch: NCHW fp32-MFMA GEMMs (dc_conv1x1_*), no layout transposes
        return _ops.conv1x1(x, conv.weight, conv.stride[0], fork=fork, skip=skip)
    if (CONV_S2 and x.is_cuda and conv.stride == (2, 2) and conv.kernel_size in ((3, 3), (7, 7)) and conv.groups == 1
            and conv.padding == (conv.kernel_size[0] // 2,) * 2 and conv.dilation == (1, 1) and conv.bias is None
            and x.dtype == torch.float32 and _ops.conv_s2_supported(x, conv.weight)
            and ((conv.kernel_size == (7, 7) and not x.requires_grad)        # (the stem kernels have no data gradient)
                 or (conv.kernel_size == (3, 3) and conv.in_channels % 4 == 0 and conv.out_channels % 32 == 0))):
        # 7x7 / 2 stem and the 3x3 / 2 convolutions: implicit GEMMs on the matrix cores (dc_convs2_*)
        if fork is not None and (not fork.pair or conv.kernel_size != (3, 3)):
            raise _ops.DepthcoreError("GradFork handed to a strided convolution")
        return _ops.conv_s2(x, conv.weight, fork)
    if fork is not None:  # (_fork_for mirrors the two conditions above; a fork nobody collects would lose the skip's gradient)
        raise _ops.DepthcoreError("GradFork handed to a convolution that does not run on a kernel with the addend epilogue")
    if not x.is_cuda:       # no CPU fallback anywhere in this package (layers.py): the oracle under oracle/ is the CPU statement
        raise _ops.DepthcoreError("convolution on a %s tensor: depthcore's modules compute on the GPU only" % x.device)
    # shapes outside the tiled kernels' 16-byte staging (odd or tiny maps, a stem whose input needs a gradient): depthcore's
    # plain direct kernels -- on the GPU no shape reaches the framework's convolution
    k, s_, p_ = conv.kernel_size, conv.stride, conv.padding
    if (x.dtype != torch.float32 or conv.groups != 1 or conv.dilation != (1, 1) or k[0] != k[1] or s_[0] != s_[1] or p_[0] != p_[1]
            or isinstance(p_, str) or conv.padding_mode != "zeros"):
        raise _ops.DepthcoreError("convolution %r on a %s %s input is outside depthcore's kernels" % (conv, x.dtype, tuple(x.shape)))
    return _ops.conv2d_direct(x, conv.weight, conv.bias, s_[0], p_[0])


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, groups_ref=None):
        super().__init__()
        self._g = groups_ref if groups_ref is not None else [1]
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample
        self.link_out = False      # set by ResNetTrunk: the NEXT block of the stage is this block output's only consumer

    def _fold_ok(self, x, g):
        """The folded chain applies: fp32 training step on the GPU, conv2 on the Winograd kernels with both epilogues."""
        if not (_fold_level(False) and WINO_TRUNK and self.training and x.is_cuda and x.dtype == torch.float32 and torch.is_grad_enabled()
                and _ops._precision[0] == _ops.PRECISIONS["f32"]):
            return False
        B, _, H, W = x.shape
        s = self.conv1.stride[0]
        if self.conv1.stride not in ((1, 1), (2, 2)) or H % s or W % s or B % g:
            return False
        return _bnf.wino_ok(self.conv2, (B, self.conv2.in_channels, H // s, W // s), g)

    def _forward_fold(self, x, g):
        """conv1 (statistics epilogue; the previous block's BatchNorm backward in its data-gradient epilogue) -> [bn1 + ReLU
        inside conv2's loader] conv2 (statistics epilogue) -> relu(bn2 + skip), one apply pass."""
        fork = _fork_for(self, x)
        if fork is not None and _bnf.wino_ok(self.conv1, tuple(x.shape), g):
            y1, s1 = _bnf.conv3x3(x, self.conv1.weight, g, fork=fork, prev=_bnf.take_link(x, self.conv1, g))
        elif fork is None and self.conv1.stride == (1, 1) and _bnf.wino_ok(self.conv1, tuple(x.shape), g):
            y1, s1 = _bnf.conv3x3(x, self.conv1.weight, g)
        else:
            y1, s1 = _conv(self.conv1, x, fork), None           # (3x3 / 2: no statistics epilogue yet -- a stand-alone pass)
        lvl = _fold_level(False)
        if lvl >= 3:
            y2, s2 = _bnf.conv3x3(y1, self.conv2.weight, g, in_bn=self.bn1, in_stats=s1)
        else:
            y2, s2 = _bnf.conv3x3(_bnf.bn_apply(y1, self.bn1, s1, groups=g), self.conv2.weight, g)
        link = self.link_out and lvl >= 2
        if self.downsample is None:
            return _bnf.bn_apply(y2, self.bn2, s2, res=x, groups=g, fork=fork, leave_link=link)
        d = self.downsample[0]
        if (GEMM_1X1 and d.kernel_size == (1, 1) and d.padding == (0, 0) and d.groups == 1 and d.bias is None and d.stride in ((1, 1), (2, 2))
                and (d.stride == (1, 1) or (x.shape[-1] % 2 == 0 and x.shape[-2] % 2 == 0))):
            yd, sd = _bnf.conv1x1(x, d.weight, d.stride[0], g)            # statistics epilogue on the tiled shapes
        else:
            yd, sd = _conv(d, x), None
        idt = _bnf.bn_apply(yd, self.downsample[1], sd, relu=False, groups=g)
        return _bnf.bn_apply(y2, self.bn2, s2, res=idt, groups=g, leave_link=link)

    def forward(self, x):
        g = self._g[0]
        if self._fold_ok(x, g):
            return self._forward_fold(x, g)
        if self.downsample is None:
            fork = _fork_for(self, x)
            out = _bn_act(_conv(self.conv1, x, fork), self.bn1, groups=g)
            return _bn_act(_conv(self.conv2, out), self.bn2, res=x, groups=g, fork=fork)   # relu(bn2(conv2) + identity), one pass
        # x feeds the 3x3 / 2 conv1 AND the 1x1 / 2 `downsample` (and, as a feature map, the decoder's skip connection): a pair
        # fork -- conv1's backward runs first (the skip branch is evaluated first here) and parks its data gradient, the
        # downsample's data-gradient kernel adds it, and whatever the SkipSum of x collected, in its store epilogue
        pair = _basic_pair_fork(self, x)
        idt = _bn_act(_conv(self.downsample[0], x, pair, _ops.skip_of(x) if pair is not None else None), self.downsample[1], relu=False, groups=g)
        out = _bn_act(_conv(self.conv1, x, pair), self.bn1, groups=g)
        return _bn_act(_conv(self.conv2, out), self.bn2, res=idt, groups=g)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, groups_ref=None):
        super().__init__()
        self._g = groups_ref if groups_ref is not None else [1]
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)   # v1.5: stride on the 3x3
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.link_out = False      # set by ResNetTrunk: the NEXT block of the stage is this block output's only consumer

    def _fold_ok(self, x, g):
        """The folded chain applies: training step on the GPU, every 1x1 of the block on the tiled GEMM kernels with both
        epilogues, fp32 matrix precision."""
        if not (_fold_level(True) and GEMM_1X1 and self.training and x.is_cuda and x.dtype == torch.float32 and torch.is_grad_enabled()
                and _ops._precision[0] == _ops.PRECISIONS["f32"] and x.numel() * 4 < 0x7fffffff):
            return False
        B, _, H, W = x.shape
        s = self.conv2.stride[0]
        if self.conv2.stride not in ((1, 1), (2, 2)) or H % s or W % s or (H * W) % 4 or ((H // s) * (W // s)) % 4 or B % g:
            return False
        return _bnf.conv1x1_ok(self.conv3, (B, self.conv2.out_channels, H // s, W // s), g)

    def _forward_fold(self, x, g):
        """conv1 (statistics epilogue; the previous block's BatchNorm backward in its data-gradient epilogue) -> bn1 + ReLU ->
        conv2 (3x3) -> [bn2 + ReLU inside conv3's loader] conv3 (statistics epilogue) -> relu(bn3 + skip), one apply pass."""
        if self.downsample is None:
            fork = _fork_for(self, x)
            prev = _bnf.take_link(x, self.conv1, g) if fork is not None else None
        else:
            fork, prev = _pair_fork_for(self, x), None
        sk = _ops.skip_of(x) if (fork is not None and fork.pair) else None
        y1, s1 = _bnf.conv1x1(x, self.conv1.weight, 1, g, fork=fork, prev=prev, skip=sk)
        lvl = _fold_level(True)
        wino2 = WINO_TRUNK and _bnf.wino_ok(self.conv2, tuple(y1.shape), g)
        if wino2 and lvl >= 3:
            y2, s2 = _bnf.conv3x3(y1, self.conv2.weight, g, in_bn=self.bn1, in_stats=s1)     # bn1 + ReLU inside conv2's loader
        elif wino2:
            y2, s2 = _bnf.conv3x3(_bnf.bn_apply(y1, self.bn1, s1, groups=g), self.conv2.weight, g)
        else:
            y2, s2 = _conv(self.conv2, _bnf.bn_apply(y1, self.bn1, s1, groups=g)), None      # (3x3 / 2: apply pass, stand-alone statistics)
        if lvl >= 3:
            y3, s3 = _bnf.conv1x1(y2, self.conv3.weight, 1, g, in_bn=self.bn2, in_stats=s2)
        else:
            y3, s3 = _bnf.conv1x1(_bnf.bn_apply(y2, self.bn2, s2, groups=g), self.conv3.weight, 1, g)
        link = self.link_out and lvl >= 2
        if self.downsample is None:
            return _bnf.bn_apply(y3, self.bn3, s3, res=x, groups=g, fork=fork, leave_link=link)
        d = self.downsample[0]
        yd, sd = _bnf.conv1x1(x, d.weight, d.stride[0], g, fork=fork, skip=sk)
        idt = _bnf.bn_apply(yd, self.downsample[1], sd, relu=False, groups=g)
        return _bnf.bn_apply(y3, self.bn3, s3, res=idt, groups=g, leave_link=link)

    def forward(self, x):
        g = self._g[0]
        if self._fold_ok(x, g):
            return self._forward_fold(x, g)
        fork = _fork_for(self, x)
        if self.downsample is None:
            out = _bn_act(_conv(self.conv1, x, fork), self.bn1, groups=g)
            out = _bn_act(_conv(self.conv2, out), self.bn2, groups=g)
            return _bn_act(_conv(self.conv3, out), self.bn3, res=x, groups=g, fork=fork)
        # conv1 and the 1x1 `downsample` both read x: a pair fork sums their two data gradients in the second one's epilogue
        pair = _pair_fork_for(self, x)
        sk = _ops.skip_of(x) if pair is not None else None
        out = _bn_act(_conv(self.conv1, x, pair, sk), self.bn1, groups=g)
        out = _bn_act(_conv(self.conv2, out), self.bn2, groups=g)
        out = _conv(self.conv3, out)
        # (the skip branch is evaluated last so that its backward runs first: the stride-1 conv1 -- whose kernel adds in its
        # store epilogue -- is then the second of the pair; either order is correct)
        idt = _bn_act(_conv(self.downsample[0], x, pair, sk), self.downsample[1], relu=False, groups=g)
        return _bn_act(out, self.bn3, res=idt, groups=g)


_CFG = {18: (BasicBlock, (2, 2, 2, 2)), 34: (BasicBlock, (3, 4, 6, 3)), 50: (Bottleneck, (3, 4, 6, 3)),
        101: (Bottleneck, (3, 4, 23, 3)), 152: (Bottleneck, (3, 8, 36, 3))}


class ResNetTrunk(nn.Module):
    """Module names follow torchvision.models.ResNet (conv1, bn1, relu, maxpool, layer1-4, fc)."""

    def __init__(self, num_layers, num_input_images=1):
        super().__init__()
        block, layers = _CFG[num_layers]
        self.inplanes = 64
        self._g = [1]       # number of independent sub-batches in the current forward (shared with the blocks)
        self.conv1 = nn.Conv2d(num_input_images * 3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], 2)
        self.layer3 = self._make_layer(block, 256, layers[2], 2)
        self.layer4 = self._make_layer(block, 512, layers[3], 2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512 * block.expansion, 1000)      # kept for checkpoint compatibility; never used
        for m in self.modules():                               # networks/resnet_encoder.py:34-39
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _make_layer(self, block, planes, blocks, stride=1):
        down = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            down = nn.Sequential(nn.Conv2d(self.inplanes, planes * block.expansion, 1, stride, bias=False),
                                 nn.BatchNorm2d(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, down, groups_ref=self._g)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes, groups_ref=self._g))
        for b in layers[:-1]:
            b.link_out = True       # (a stage's last output is a feature map with several consumers: no link)
        return nn.Sequential(*layers)


def load_imagenet_weights(trunk, weights, num_input_images=1):
    """networks/resnet_encoder.py:52-57: a torchvision ResNet state dict goes into the trunk with `conv1.weight` tiled over
    the stacked input frames and divided by their number (so the stem's response to identical frames is unchanged).
    `weights`: a state dict or the path of a torchvision `resnet{N}-*.pth` file."""
    loaded = torch.load(weights, map_location="cpu") if isinstance(weights, (str, bytes)) or hasattr(weights, "__fspath__") \
        else dict(weights)
    loaded = {k: v for k, v in loaded.items()}
    loaded["conv1.weight"] = torch.cat([loaded["conv1.weight"]] * num_input_images, 1) / num_input_images
    trunk.load_state_dict(loaded)
    return trunk


class ResnetEncoder(nn.Module):
    """networks/resnet_encoder.py:62-98.  `pretrained`: False (scratch init), or the torchvision ImageNet weights as a
    state dict / the path of their file (applied with the reference's tile-and-divide rule for `num_input_images` > 1).
    `pretrained=True` would need the reference's download (model_zoo.load_url) and is refused: there is no network."""

    def __init__(self, num_layers, pretrained, num_input_images=1):
        super().__init__()
        if num_layers not in _CFG:
            raise ValueError("{} is not a valid number of resnet layers".format(num_layers))
        if pretrained is True:
            raise RuntimeError("pretrained=True downloads the ImageNet weights (networks/resnet_encoder.py:53); pass the "
                               "torchvision state dict or the path of its .pth file as `pretrained` instead")
        self.num_ch_enc = np.array([64, 64, 128, 256, 512])
        self.encoder = ResNetTrunk(num_layers, num_input_images)
        if pretrained:
            load_imagenet_weights(self.encoder, pretrained, num_input_images)
        if num_layers > 34:
            self.num_ch_enc[1:] *= 4
        self._nbt = None

    def _apply(self, fn, *a, **k):
        self._nbt = None            # tensors may be re-created by .to() / .cuda()
        return super()._apply(fn, *a, **k)

    def forward_pairs(self, f_prev, f_cur, f_next):
        """Both temporally ordered pairs of the pose network in ONE pass: equals `forward(cat([cat([f_prev, f_cur], 1),
        cat([f_cur, f_next], 1)], 0), bn_groups=2)` (trainer.py:398-419: pairs (-1, 0) and (0, +1)), with the concatenations
        and the input normalisation done by the stem kernel's loader when the shapes allow."""
        frames = (f_prev, f_cur, f_next)
        if STEM_FUSED and _ops.stem_supported(frames, self.encoder.conv1.weight):
            return self._trunk(_ops.stem_conv(frames, self.encoder.conv1.weight), 2)
        pairs = torch.cat([torch.cat([f_prev, f_cur], 1), torch.cat([f_cur, f_next], 1)], 0)
        return self.forward(pairs, bn_groups=2)

    def forward(self, input_image, bn_groups=1):
        """networks/resnet_encoder.py:87-98.  `bn_groups` > 1: `input_image` stacks that many independent
        sub-batches along dim 0; BatchNorm statistics (and running-stat updates) are kept per sub-batch, so the
        result equals `bn_groups` separate calls -- used to push both pose pairs through the trunk at once."""
        e = self.encoder
        if STEM_FUSED and input_image.shape[1] == 3 and _ops.stem_supported((input_image,), e.conv1.weight):
            return self._trunk(_ops.stem_conv((input_image,), e.conv1.weight), bn_groups)     # normalisation inside the loader
        x = (input_image - 0.45) / 0.225
        return self._trunk(_conv(e.conv1, x), bn_groups)

    def _trunk(self, x, bn_groups):
        """everything behind conv1; x = conv1((input - 0.45) / 0.225)"""
        e = self.encoder
        e._g[0] = int(bn_groups)
        self.features = []
        x = _bn_act(x, e.bn1, groups=e._g[0])
        tag = SKIP_SUMS and self.training and x.is_cuda and x.requires_grad and torch.is_grad_enabled()

        def feature(t):
            # a feature map has a primary consumer in the trunk (the max-pool, the next stage's first block) and, in the depth
            # network, the decoder's skip connection: the SkipSum lets the primary's backward kernel add the latter's gradient
            if tag:
                t._dc_skip = _ops.SkipSum()
            self.features.append(t)
            return t
        feature(x)
        feature(e.layer1(_ops.maxpool3x3s2(x, _ops.skip_of(x)) if x.is_cuda else e.maxpool(x)))
        feature(e.layer2(self.features[-1]))
        feature(e.layer3(self.features[-1]))
        self.features.append(e.layer4(self.features[-1]))
        if self.training:   # nn.BatchNorm2d bookkeeping, one multi-tensor launch instead of one per layer
            if self._nbt is None:
                self._nbt = [m.num_batches_tracked for m in e.modules() if isinstance(m, nn.BatchNorm2d)]
            torch._foreach_add_(self._nbt, int(bn_groups))
        e._g[0] = 1
        return self.features
