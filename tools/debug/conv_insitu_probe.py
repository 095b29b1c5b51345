#!/usr/bin/env python3
"""Debug probe (GPU): every convolution of the encoder re-evaluated in isolation (torch fp64) on its recorded input, weight
and output gradient: weight gradient, and input gradient where the input has this convolution as its only consumer."""
import os, sys
import torch, torch.nn.functional as F
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "self-supervised-depth-estimation_amd")); sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import networks
from networks import resnet_encoder as RE
from helpers import rel_l2
torch.manual_seed(0)
enc = networks.ResnetEncoder(18, False).cuda(); enc.train()
rec = []
orig = RE._conv
def wrap(conv, x):
    if x.requires_grad: x.retain_grad()
    y = orig(conv, x); y.retain_grad(); rec.append((conv, x, y)); return y
RE._conv = wrap
g = torch.Generator().manual_seed(1)
x = torch.rand(4, 3, 64, 128, generator=g)
got = enc(x.cuda())
cots = [torch.randn(f.shape, generator=g) / f[0].numel() ** 0.5 for f in got]
sum((f * c.cuda()).sum() for f, c in zip(got, cots)).backward()
uses = {}
for conv, xx, y in rec: uses[id(xx)] = uses.get(id(xx), 0) + 1
for i, (conv, xx, y) in enumerate(rec):
    xr = xx.detach().double().cpu().requires_grad_(); wr = conv.weight.detach().double().cpu().requires_grad_()
    yr = F.conv2d(xr, wr, None, conv.stride, conv.padding); yr.backward(y.grad.double().cpu())
    single = uses[id(xx)] == 1 and xx.grad is not None and i > 0
    print("conv %2d k%d s%d %-18s y %.1e  dW %.3e  dX %s" % (i, conv.kernel_size[0], conv.stride[0], tuple(xx.shape), rel_l2(y, yr),
          rel_l2(conv.weight.grad, wr.grad), ("%.3e" % rel_l2(xx.grad, xr.grad)) if single else "(shared input)"))
