#!/usr/bin/env python3
"""Debug probe (GPU): every fused BN(+res)(+ReLU) call of the encoder re-evaluated in isolation by torch fp64 on the
recorded inputs and output gradient."""
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
orig = RE._ops.bn_relu
def wrap(x, bn, res=None, relu=True, groups=1):
    x.retain_grad()
    if res is not None: res.retain_grad()
    y = orig(x, bn, res, relu, groups); y.retain_grad()
    rec.append((x, res, y, bn, relu)); return y
RE._ops.bn_relu = wrap
g = torch.Generator().manual_seed(1)
x = torch.rand(4, 3, 64, 128, generator=g)
got = enc(x.cuda())
cots = [torch.randn(f.shape, generator=g) / f[0].numel() ** 0.5 for f in got]
sum((f * c.cuda()).sum() for f, c in zip(got, cots)).backward()
for i, (xx, res, y, bn, relu) in enumerate(rec):
    xr = xx.detach().double().cpu().requires_grad_()
    rr = res.detach().double().cpu().requires_grad_() if res is not None else None
    w, b = bn.weight.detach().double().cpu().requires_grad_(), bn.bias.detach().double().cpu().requires_grad_()
    yr = F.batch_norm(xr, None, None, w, b, True, 0.1, bn.eps)
    if rr is not None: yr = yr + rr
    if relu: yr = F.relu(yr)
    yr.backward(y.grad.double().cpu())
    print("bn %2d %-18s res %d relu %d: y %.1e dx %.3e dres %s dgamma %.1e dbeta %.1e" % (
        i, tuple(xx.shape), res is not None, relu, rel_l2(y, yr), rel_l2(xx.grad, xr.grad),
        "%.3e" % rel_l2(res.grad, rr.grad) if res is not None and res.grad is not None else "-",
        rel_l2(bn.weight.grad, w.grad), rel_l2(bn.bias.grad, b.grad)))
