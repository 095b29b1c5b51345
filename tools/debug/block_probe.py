#!/usr/bin/env python3
"""Debug probe (GPU): one BasicBlock (with / without stride-2 downsample) forward + input gradient vs the fp64 oracle."""
import os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "self-supervised-depth-estimation_amd")); sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import torch.nn as nn
from networks import resnet_encoder as RE
from oracle import resnet_ref as RR
from helpers import rel_l2
g = torch.Generator().manual_seed(0)
for (B, Cin, planes, H, W, stride) in [(4, 256, 512, 4, 8, 2), (4, 256, 256, 4, 8, 1), (4, 512, 512, 2, 4, 1), (4, 128, 256, 8, 16, 2), (4, 64, 64, 16, 32, 1)]:
    torch.manual_seed(1)
    down = None
    if stride != 1 or Cin != planes:
        down = nn.Sequential(nn.Conv2d(Cin, planes, 1, stride, bias=False), nn.BatchNorm2d(planes))
    blk = RE.BasicBlock(Cin, planes, stride, down).cuda().train()
    x = torch.relu(torch.randn(B, Cin, H, W, generator=g))
    cot = torch.randn(B, planes, H // stride, W // stride, generator=g)
    xh = x.cuda().requires_grad_()
    yh = blk(xh); yh.backward(cot.cuda())
    st = {"layer1.0." + k: (v.detach().cpu().double().requires_grad_() if v.is_floating_point() and "running" not in k else v.detach().cpu()) for k, v in blk.state_dict().items()}
    for k in list(st):
        if st[k].is_floating_point() and not st[k].requires_grad: st[k] = st[k].double()
    xr = x.double().requires_grad_()
    yr = RR._basic(xr, st, "layer1.0", stride, True); yr.backward(cot.double())
    errs = {n: rel_l2(p.grad, st["layer1.0." + n].grad) for n, p in blk.named_parameters()}
    print((B, Cin, planes, H, W, stride), "y %.2e dx %.3e" % (rel_l2(yh, yr), rel_l2(xh.grad, xr.grad)), "worst param", max(errs.items(), key=lambda t: t[1]))
