"""Functional CPU ResNet encoder over a torchvision-layout state_dict.

TEST INFRASTRUCTURE (see oracle/__init__.py).  PARITY UNPINNED: the reference
(`networks/resnet_encoder.py:62-98`) delegates the trunk to
`torchvision.models.resnet*`, which is neither vendored nor installed here.
This restates torchvision's published ResNet v1.5 (BasicBlock for 18/34,
Bottleneck with the stride on the 3x3 for >=50) and is pinned by key/shape
equality with that layout and by the analytic parameter count.
"""
import torch
import torch.nn.functional as F

from .kinks import Kinks

BLOCKS = {18: (2, 2, 2, 2), 34: (3, 4, 6, 3), 50: (3, 4, 6, 3), 101: (3, 4, 23, 3), 152: (3, 8, 36, 3)}


def _bn(x, st, prefix, training, eps=1e-5):
    return F.batch_norm(x, st[prefix + ".running_mean"].clone(), st[prefix + ".running_var"].clone(),
                        st[prefix + ".weight"], st[prefix + ".bias"], training, 0.1, eps)


def _basic(x, st, p, stride, training, kn):
    out = F.conv2d(x, st[p + ".conv1.weight"], None, stride, 1)
    out = kn.relu(_bn(out, st, p + ".bn1", training))
    out = F.conv2d(out, st[p + ".conv2.weight"], None, 1, 1)
    out = _bn(out, st, p + ".bn2", training)
    if (p + ".downsample.0.weight") in st:
        x = _bn(F.conv2d(x, st[p + ".downsample.0.weight"], None, stride), st, p + ".downsample.1", training)
    return kn.relu(out + x)


def _bottleneck(x, st, p, stride, training, kn):
    out = kn.relu(_bn(F.conv2d(x, st[p + ".conv1.weight"]), st, p + ".bn1", training))
    out = kn.relu(_bn(F.conv2d(out, st[p + ".conv2.weight"], None, stride, 1), st, p + ".bn2", training))
    out = _bn(F.conv2d(out, st[p + ".conv3.weight"]), st, p + ".bn3", training)
    if (p + ".downsample.0.weight") in st:
        x = _bn(F.conv2d(x, st[p + ".downsample.0.weight"], None, stride), st, p + ".downsample.1", training)
    return kn.relu(out + x)


def resnet_encoder_forward(state, img, num_layers=18, training=True, kinks=None):
    """networks/resnet_encoder.py:87-98: 5 feature maps; keys prefixed `encoder.`.
    `kinks`: oracle.kinks.ForcedKinks to impose recorded ReLU / max-pool decisions (default: the plain functions)."""
    kn = kinks if kinks is not None else Kinks()
    st = {k[len("encoder."):]: v for k, v in state.items() if k.startswith("encoder.")}
    block = _basic if num_layers <= 34 else _bottleneck
    feats = []
    x = (img - 0.45) / 0.225
    x = F.conv2d(x, st["conv1.weight"], None, 2, 3)
    x = kn.relu(_bn(x, st, "bn1", training))
    feats.append(x)
    x = kn.max_pool(x)
    for li, n in enumerate(BLOCKS[num_layers], start=1):
        for j in range(n):
            stride = 2 if (li > 1 and j == 0) else 1
            x = block(x, st, "layer%d.%d" % (li, j), stride, training, kn)
        feats.append(x)
    return feats
