"""PoseCNN with the reference's constructor / forward / state_dict layout (reference networks/pose_cnn.py:14-53): keys
`pose_conv.{weight,bias}`, `net.{0..6}.{weight,bias}`.  Used with `pose_model_type = "posecnn"` (trainer.py:106-108)."""
import torch
import torch.nn as nn

from depthcore import ops as _ops


class PoseCNN(nn.Module):
    def __init__(self, num_input_frames):
        super().__init__()
        self.num_input_frames = num_input_frames
        self.convs = {}
        self.convs[0] = nn.Conv2d(3 * num_input_frames, 16, 7, 2, 3)
        self.convs[1] = nn.Conv2d(16, 32, 5, 2, 2)
        self.convs[2] = nn.Conv2d(32, 64, 3, 2, 1)
        self.convs[3] = nn.Conv2d(64, 128, 3, 2, 1)
        self.convs[4] = nn.Conv2d(128, 256, 3, 2, 1)
        self.convs[5] = nn.Conv2d(256, 256, 3, 2, 1)
        self.convs[6] = nn.Conv2d(256, 256, 3, 2, 1)
        self.pose_conv = nn.Conv2d(256, 6 * (num_input_frames - 1), 1)
        self.num_convs = len(self.convs)
        self.relu = nn.ReLU(True)
        self.net = nn.ModuleList(list(self.convs.values()))

    def _trunk(self, out):
        """networks/pose_cnn.py:40-48.  On the GPU the seven strided convolutions (7x7, 5x5, 3x3 with bias) run on depthcore's
        direct kernels (dc_conv2d_direct_*: this network is outside the BASELINE configurations, the plain kernels are the
        honest cost) and the 1x1 head on dc_conv1x1_bias_act_fwd; no library convolution."""
        if not out.is_cuda:     # no CPU fallback anywhere in this package (layers.py)
            raise _ops.DepthcoreError("PoseCNN on a %s tensor: depthcore's modules compute on the GPU only" % out.device)
        for i in range(self.num_convs):
            c = self.convs[i]
            out = torch.relu(_ops.conv2d_direct(out, c.weight, c.bias, 2, c.padding[0]))
        return _ops.conv1x1(out, self.pose_conv.weight, 1, self.pose_conv.bias, _ops.ACT_NONE)

    def forward(self, out):
        """networks/pose_cnn.py:40-53."""
        out = self._trunk(out).mean(3).mean(2)
        out = 0.01 * out.view(-1, self.num_input_frames - 1, 1, 6)
        return out[..., :3], out[..., 3:]

    def forward_poses(self, out, groups):
        """forward() + the callers' cam_T_cam with the tail as one launch each way (as PoseDecoder.forward_poses)."""
        return _ops.pose_head(self._trunk(out), self.num_input_frames - 1, groups)
