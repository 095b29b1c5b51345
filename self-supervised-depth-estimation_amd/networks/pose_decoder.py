"""PoseDecoder with the reference's constructor / forward / state_dict layout
(reference networks/pose_decoder.py:14-54): keys `net.{0..3}.{weight,bias}`."""
from collections import OrderedDict

import torch
import torch.nn as nn

from depthcore import ops as _ops


class PoseDecoder(nn.Module):
    def __init__(self, num_ch_enc, num_input_features, num_frames_to_predict_for=None, stride=1):
        super().__init__()
        self.num_ch_enc = num_ch_enc
        self.num_input_features = num_input_features
        if num_frames_to_predict_for is None:
            num_frames_to_predict_for = num_input_features - 1
        self.num_frames_to_predict_for = num_frames_to_predict_for
        self.convs = OrderedDict()
        self.convs[("squeeze")] = nn.Conv2d(int(self.num_ch_enc[-1]), 256, 1)
        self.convs[("pose", 0)] = nn.Conv2d(num_input_features * 256, 256, 3, stride, 1)
        self.convs[("pose", 1)] = nn.Conv2d(256, 256, 3, stride, 1)
        self.convs[("pose", 2)] = nn.Conv2d(256, 6 * num_frames_to_predict_for, 1)
        self.relu = nn.ReLU()
        self.net = nn.ModuleList(list(self.convs.values()))

    def _trunk(self, input_features):
        """networks/pose_decoder.py:40-50: squeeze -> pose 0..2.  On the GPU every convolution is a depthcore launch with its
        bias and ReLU in the epilogue (dc_conv1x1_bias_act_fwd, dc_conv3x3_fwd); there is no library convolution on this path."""
        last = [f[-1] for f in input_features]
        if last[0].is_cuda:
            sq = self.convs["squeeze"]
            out = [_ops.conv1x1(f, sq.weight, 1, sq.bias, _ops.ACT_RELU) for f in last]
            out = out[0] if len(out) == 1 else torch.cat(out, 1)
            for i in range(2):
                conv = self.convs[("pose", i)]
                if conv.stride != (1, 1):
                    raise NotImplementedError("PoseDecoder stride != 1 is not on the hot path (the reference never sets it)")
                out = _ops.conv3x3_block(out, None, conv.weight, conv.bias, False, _ops.ACT_RELU, _ops.PAD_ZERO)
            fin = self.convs[("pose", 2)]
            out = _ops.conv1x1(out, fin.weight, 1, fin.bias, _ops.ACT_NONE)
        else:               # CPU: module bookkeeping / export only, not a compute path of this package
            out = torch.cat([self.relu(self.convs["squeeze"](f)) for f in last], 1)
            for i in range(3):
                out = self.convs[("pose", i)](out)
                if i != 2:
                    out = self.relu(out)
        return out

    def forward(self, input_features):
        """networks/pose_decoder.py:40-54."""
        out = self._trunk(input_features)
        out = out.mean(3).mean(2)
        out = 0.01 * out.view(-1, self.num_frames_to_predict_for, 1, 6)
        return out[..., :3], out[..., 3:]

    def forward_poses(self, input_features, groups):
        """forward() + the callers' transformation_from_parameters (trainer.py:416-419, 436-440) with the tail -- pixel mean,
        0.01, the axisangle / translation split, the frame selection and the 4x4 matrices -- as one launch each way
        (depthcore.ops.pose_head): -> axisangle, translation (records, not differentiable), [cam_T_cam per group];
        groups: (row0, rows, predicted frame, invert)."""
        return _ops.pose_head(self._trunk(input_features), self.num_frames_to_predict_for, groups)
