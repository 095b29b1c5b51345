"""DepthDecoder with the reference's constructor / forward / state_dict layout
(reference networks/depth_decoder.py:17-67): state_dict keys `decoder.{0..9}.conv.conv.*` for the
ten ConvBlocks (upconv(4,0), (4,1), (3,0) ... (0,1)) and `decoder.{10..}.conv.*` for the dispconvs."""
import os
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn

import layers as _layers
from layers import ConvBlock, Conv3x3
from depthcore import ops as _ops
from depthcore.ops import ACT_SIGMOID as _ACT_SIGMOID


X_FORK = os.environ.get("DC_GRAD_SUMS", "1") != "0"     # level i's x feeds dispconv(i) and upconv(i-1, 0): their two gradients are summed inside the second one's backward pass


class DepthDecoder(nn.Module):
    def __init__(self, num_ch_enc, scales=range(4), num_output_channels=1, use_skips=True):
        super().__init__()
        self.num_output_channels = num_output_channels
        self.use_skips = use_skips
        self.upsample_mode = "nearest"
        self.scales = scales
        self.num_ch_enc = num_ch_enc
        self.num_ch_dec = np.array([16, 32, 64, 128, 256])

        self.convs = OrderedDict()
        for i in range(4, -1, -1):
            cin = self.num_ch_enc[-1] if i == 4 else self.num_ch_dec[i + 1]
            self.convs[("upconv", i, 0)] = ConvBlock(cin, self.num_ch_dec[i])
            cin = self.num_ch_dec[i] + (self.num_ch_enc[i - 1] if (self.use_skips and i > 0) else 0)
            self.convs[("upconv", i, 1)] = ConvBlock(cin, self.num_ch_dec[i])
        for s in self.scales:
            self.convs[("dispconv", s)] = Conv3x3(self.num_ch_dec[s], self.num_output_channels)
        self.decoder = nn.ModuleList(list(self.convs.values()))
        self.sigmoid = nn.Sigmoid()

    def forward(self, input_features, pre_disp=False):
        """depth_decoder.py:50-67.  Each level is two launches: upconv(i,0), then nearest-x2 + skip concat +
        upconv(i,1) fused; the dispconv + sigmoid is a third."""
        self.outputs = {}
        x = input_features[-1]
        fork = None
        for i in range(4, -1, -1):
            x = self.convs[("upconv", i, 0)](x, fork=fork)
            skip = input_features[i - 1] if (self.use_skips and i > 0) else None
            x = self.convs[("upconv", i, 1)](x, skip, up=True)
            fork = None
            if i in self.scales:
                if pre_disp:
                    self.outputs[("disp", i)] = x
                else:
                    # level i's x feeds dispconv(i) AND upconv(i-1, 0): a pair GradFork -- upconv's backward runs first and
                    # parks its gradient of x, dispconv's data-gradient pass adds it (no elementwise sum by autograd)
                    if X_FORK and i > 0 and self.training and x.is_cuda and x.requires_grad and torch.is_grad_enabled():
                        fork = _ops.GradFork(pair=True)
                    self.outputs[("disp", i)] = self.convs[("dispconv", i)](x, act=_ACT_SIGMOID, fork=fork)
        return self.outputs
