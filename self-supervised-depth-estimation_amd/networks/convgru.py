"""ConvGRU temporal fusion, `gru_version = v5` (reference networks/rnn.py:101-158 `ConvGRUCell` / `ConvGRUModel_v1`,
:960-1028 `ConvGRUBlocks_v5`; caller trainer_gru.py:595-644 `run_gru_v5`) with the reference's constructors and
state_dict layout: `cgru_{0..4}.cgru_1.conv_gates.{weight,bias}`, `cgru_{k}.cgru_1.conv_can.{weight,bias}`,
`cgru_{k}.h0_layer1` (21.9 M parameters, 2.9 M of them learned initial states).

A cell is four depthcore launches: the gate convolution over cat(x, h) and the candidate convolution over cat(x, r*h) are
fused conv blocks (the concatenation is index arithmetic in their staging; bias + sigmoid / tanh in the epilogue), the two
gate products are dc_gru_rh / dc_gru_blend."""
import os

import torch
import torch.nn as nn

from depthcore import ops as _ops
from depthcore._lib import DepthcoreError

# the levels' nodes on a stream each: measured SLOWER (C4 --graph 332 vs 359 frames/s, eager 264 vs 303 on the same box: the
# cross-queue dependencies cost more than the overlap of ~10 us kernels gains) -- off, kept for the A/B
LEVEL_STREAMS = os.environ.get("DC_GRU_LEVEL_STREAMS", "0") == "1"
_streams = {}


def _level_streams(device, cur, n):
    """n side streams per (device, stream the step runs on): the same ones every step (a captured graph forks into them)."""
    key = (device.index, cur.cuda_stream)
    if key not in _streams or len(_streams[key]) < n:
        _streams[key] = [torch.cuda.Stream(device) for _ in range(n)]
    return _streams[key][:n]


LEVEL_NODES = os.environ.get("DC_GRU_LEVEL_NODES", "1") != "0"    # 0: the per-op graph (one node per gate product / convolution), for A/Bs


class ConvGRUCell(nn.Module):
    """networks/rnn.py:101-143."""

    def __init__(self, input_size, input_dim, hidden_dim, kernel_size, bias):
        super().__init__()
        if tuple(kernel_size) != (3, 3) or not bias:
            raise NotImplementedError("ConvGRUCell kernels cover kernel_size (3, 3) with bias (trainer_gru.py:128)")
        self.height, self.width = input_size
        self.padding = kernel_size[0] // 2, kernel_size[1] // 2
        self.hidden_dim = hidden_dim
        self.bias = bias
        self.conv_gates = nn.Conv2d(input_dim + hidden_dim, 2 * hidden_dim, kernel_size, padding=self.padding, bias=bias)
        self.conv_can = nn.Conv2d(input_dim + hidden_dim, hidden_dim, kernel_size, padding=self.padding, bias=bias)

    def init_hidden(self, batch_size):
        return torch.zeros(batch_size, self.hidden_dim, self.height, self.width)

    def forward(self, input_tensor, h_cur):
        if not input_tensor.is_cuda:
            raise DepthcoreError("ConvGRUCell runs on depthcore kernels only; there is no CPU path")
        g, c = self.conv_gates, self.conv_can
        # [reset | update] = sigmoid(conv_gates(cat(x, h)))                                         rnn.py:125-130
        gates = _ops.conv3x3_block(input_tensor, h_cur, g.weight, g.bias, False, _ops.ACT_SIGMOID, _ops.PAD_ZERO)
        rh = _ops.gru_reset_times_state(gates, h_cur)
        # cnm = tanh(conv_can(cat(x, reset * h)))                                                     rnn.py:132-134
        cnm = _ops.conv3x3_block(input_tensor, rh, c.weight, c.bias, False, _ops.ACT_TANH, _ops.PAD_ZERO)
        return _ops.gru_blend(gates, h_cur, cnm)                                                    # rnn.py:136


class ConvGRUModel_v1(nn.Module):
    """networks/rnn.py:146-161: one cell + its learned initial state `h0_layer1` (1, hidden, H, W), zeros at init."""

    def __init__(self, dim_dict, kernel_size, bias, device):
        super().__init__()
        self.cgru_1 = ConvGRUCell((dim_dict["height"], dim_dict["width"]), dim_dict["input_dim"], dim_dict["hidden_dim_1"],
                                  kernel_size, bias)
        self.h0_layer1 = self.init_hidden(dim_dict["hidden_dim_1"], dim_dict["height"], dim_dict["width"], device)
        self.height, self.width, self.hidden_dim = dim_dict["height"], dim_dict["width"], dim_dict["hidden_dim_1"]

    def init_hidden(self, hidden_dim, height, width, device):
        return nn.Parameter(torch.zeros(1, hidden_dim, height, width, device=device), requires_grad=True)

    def forward(self, x, hidden_state_1):
        return self.cgru_1(x, hidden_state_1)


class ConvGRUBlocks_v5(nn.Module):
    """networks/rnn.py:960-1028: one ConvGRU per encoder feature level ("GRU inside skip connections").  The reference
    hard-codes the 192 x 640 feature sizes of a ResNet-18/34 encoder; `height` / `width` / `num_ch_enc` reproduce those
    numbers by default and let other input sizes (tests, 320 x 1024) be built."""

    def __init__(self, kernel_size, bias, device, fuse=True, height=192, width=640, num_ch_enc=(64, 64, 128, 256, 512)):
        super().__init__()
        self.fuse = fuse
        for k, ch in enumerate(num_ch_enc):
            dims = {"input_dim": int(ch), "hidden_dim_1": int(ch), "height": height >> (k + 1), "width": width >> (k + 1)}
            setattr(self, "cgru_%d" % k, ConvGRUModel_v1(dims, kernel_size, bias, device))

    def cells(self):
        return [getattr(self, "cgru_%d" % k) for k in range(5)]

    def forward(self, encoder_features, hidden_states):
        return [cell(f, h) for cell, f, h in zip(self.cells(), encoder_features, hidden_states)]

    def run_sequence(self, features):
        """trainer_gru.py:607-639 for batch_size 1: `features[k]` (n, C_k, h_k, w_k) holds the encoder features of the n
        frames of one sequence; the cells run over the frames in order from the learned initial states; returns
        features[k] + (H[1:] + H[:-1]) / 2 with H the n+1 hidden states."""
        if LEVEL_NODES and features[0].is_cuda:
            # one autograd node per level: the frame loop, the trace and the reverse walk live inside it (ops._GruLevel)
            if not LEVEL_STREAMS:
                return [_ops.gru_level_sequence(f, cell.h0_layer1, cell.cgru_1.conv_gates, cell.cgru_1.conv_can)
                        for cell, f in zip(self.cells(), features)]
            # the five levels do not depend on each other and each is a chain of small dependent launches: every level on its
            # own stream (autograd runs a node's backward on the stream of its forward, so the reverse walks overlap too)
            cur = torch.cuda.current_stream(features[0].device)
            side = _level_streams(features[0].device, cur, len(features))
            outs = []
            for cell, f, st in zip(self.cells(), features, side):
                st.wait_stream(cur)
                with torch.cuda.stream(st):
                    o = _ops.gru_level_sequence(f, cell.h0_layer1, cell.cgru_1.conv_gates, cell.cgru_1.conv_can)
                f.record_stream(st)
                o.record_stream(cur)
                outs.append(o)
            for st in side:
                cur.wait_stream(st)
            return outs
        n = features[0].shape[0]
        hidden = [cell.h0_layer1 for cell in self.cells()]
        trace = [[h] for h in hidden]
        for i in range(n):
            hidden = self([f[i:i + 1] for f in features], hidden)
            for k in range(5):
                trace[k].append(hidden[k])
        return [_ops.gru_sequence_residual(features[k], torch.cat(trace[k], 0)) for k in range(5)]
