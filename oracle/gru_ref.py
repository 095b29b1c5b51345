"""CPU restatement of the reference's ConvGRU temporal fusion, `gru_version = v5` (BASELINE configs[3]).

TEST INFRASTRUCTURE (see oracle/__init__.py).  Functional over a plain state dict with the reference's key names
(`cgru_{k}.cgru_1.conv_gates.weight`, ..., `cgru_{k}.h0_layer1`).  Pinned by tests/golden/convgru.npz, which
tests/golden/make_golden_r2.py generates by running the reference's networks/rnn.py (loaded by file path).

Reference lines followed: networks/rnn.py:101-143 (ConvGRUCell), :146-161 (ConvGRUModel_v1), :960-1028
(ConvGRUBlocks_v5); trainer_gru.py:595-644 (run_gru_v5, batch size 1).
"""
import torch
import torch.nn.functional as F


def conv_gru_cell(x, h, st, p):
    """networks/rnn.py:122-143."""
    C = h.shape[1]
    comb = torch.cat([x, h], 1)
    cc = F.conv2d(comb, st[p + "conv_gates.weight"], st[p + "conv_gates.bias"], padding=1)
    reset, update = torch.sigmoid(cc[:, :C]), torch.sigmoid(cc[:, C:])      # gamma -> reset, beta -> update
    comb2 = torch.cat([x, reset * h], 1)
    cnm = torch.tanh(F.conv2d(comb2, st[p + "conv_can.weight"], st[p + "conv_can.bias"], padding=1))
    return (1 - update) * h + update * cnm


def run_gru_v5(features, st):
    """trainer_gru.py:607-639 at batch size 1: per level, the cell runs over the n frames from the learned state
    `h0_layer1`; returns features[k] + (H[1:] + H[:-1]) / 2."""
    n = features[0].shape[0]
    out = []
    for k in range(5):
        p = "cgru_%d." % k
        h = st[p + "h0_layer1"]
        trace = [h]
        for i in range(n):
            h = conv_gru_cell(features[k][i:i + 1], h, st, p + "cgru_1.")
            trace.append(h)
        H = torch.cat(trace, 0)
        out.append(features[k] + (H[1:] + H[:-1]) / 2)
    return out


def gru_v5_layout(height=192, width=640, num_ch_enc=(64, 64, 128, 256, 512)):
    """(key, shape) of ConvGRUBlocks_v5's state dict in registration order."""
    lay = []
    for k, c in enumerate(num_ch_enc):
        p = "cgru_%d." % k
        lay += [(p + "h0_layer1", (1, c, height >> (k + 1), width >> (k + 1))),
                (p + "cgru_1.conv_gates.weight", (2 * c, 2 * c, 3, 3)), (p + "cgru_1.conv_gates.bias", (2 * c,)),
                (p + "cgru_1.conv_can.weight", (c, 2 * c, 3, 3)), (p + "cgru_1.conv_can.bias", (c,))]
    return lay
