#!/usr/bin/env python3
"""Debug probe (GPU): data gradients of the stride-2 convolutions at the small shapes of the encoder test."""
import os, sys
import torch, torch.nn.functional as F
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "self-supervised-depth-estimation_amd")); sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from depthcore import ops
from helpers import rel_l2
g = torch.Generator().manual_seed(0)
for (B, Ci, Co, H, W) in [(4, 256, 512, 4, 8), (4, 128, 256, 8, 16), (4, 64, 128, 16, 32), (2, 64, 128, 48, 160)]:
    x = torch.randn(B, Ci, H, W, generator=g); w3 = torch.randn(Co, Ci, 3, 3, generator=g) * 0.05; w1 = torch.randn(Co, Ci, 1, 1, generator=g) * 0.05
    gy = torch.randn(B, Co, H // 2, W // 2, generator=g)
    for name, w, fn, ref in (("3x3s2", w3, lambda a, b: ops.conv_s2(a, b), lambda a, b: F.conv2d(a, b, None, 2, 1)),
                             ("1x1s2", w1, lambda a, b: ops.conv1x1(a, b, 2), lambda a, b: F.conv2d(a, b, None, 2, 0))):
        xh, wh = x.cuda().requires_grad_(), w.cuda().requires_grad_()
        y = fn(xh, wh); y.backward(gy.cuda())
        xr, wr = x.double().requires_grad_(), w.double().requires_grad_()
        yr = ref(xr, wr); yr.backward(gy.double())
        print(name, (B, Ci, Co, H, W), "y %.2e dx %.2e dw %.2e" % (rel_l2(y, yr), rel_l2(xh.grad, xr.grad), rel_l2(wh.grad, wr.grad)))
