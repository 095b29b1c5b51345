#!/usr/bin/env python3
"""Debug probe (GPU): gradient accumulation at a tensor with two depthcore consumers."""
import os, sys
import torch, torch.nn.functional as F
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "self-supervised-depth-estimation_amd")); sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from depthcore import ops
from helpers import rel_l2
g = torch.Generator().manual_seed(0)
B, Ci, Co, H, W = 4, 256, 512, 4, 8
x = torch.randn(B, Ci, H, W, generator=g); w3 = torch.randn(Co, Ci, 3, 3, generator=g) * 0.05; w1 = torch.randn(Co, Ci, 1, 1, generator=g) * 0.05
c1 = torch.randn(B, Co, H // 2, W // 2, generator=g); c2 = torch.randn(B, Co, H // 2, W // 2, generator=g)
xr = x.double().requires_grad_()
(F.conv2d(xr, w3.double(), None, 2, 1) * c1.double()).sum().backward(); ga = xr.grad.clone(); xr.grad = None
(F.conv2d(xr, w1.double(), None, 2, 0) * c2.double()).sum().backward(); gb = xr.grad.clone()
for order in ("3x3 first", "1x1 first"):
    xh = x.cuda().requires_grad_()
    xm = xh * 1.0                         # non-leaf, like a feature map
    if order == "3x3 first":
        y1 = ops.conv_s2(xm, w3.cuda()); y2 = ops.conv1x1(xm, w1.cuda(), 2)
    else:
        y2 = ops.conv1x1(xm, w1.cuda(), 2); y1 = ops.conv_s2(xm, w3.cuda())
    ((y1 * c1.cuda()).sum() + (y2 * c2.cuda()).sum()).backward()
    print(order, "sum %.3e   vs 3x3 only %.3e   vs 1x1 only %.3e" % (rel_l2(xh.grad, ga + gb), rel_l2(xh.grad, ga), rel_l2(xh.grad, gb)))
