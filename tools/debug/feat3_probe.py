#!/usr/bin/env python3
"""Debug probe (GPU): d(feature 3) of the encoder rebuilt from its pieces."""
import os, sys, types
import torch, torch.nn.functional as F
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "self-supervised-depth-estimation_amd")); sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import networks
from networks import resnet_encoder as RE
from oracle import resnet_ref as RR
from helpers import rel_l2
torch.manual_seed(0)
enc = networks.ResnetEncoder(18, False).cuda(); enc.train()
state = {k: v.detach().cpu().clone() for k, v in enc.state_dict().items()}
rec = []
orig = RE._conv
def wrap(conv, x):
    y = orig(conv, x); y.retain_grad(); rec.append((conv, x, y)); return y
RE._conv = wrap
g = torch.Generator().manual_seed(1)
x = torch.rand(4, 3, 64, 128, generator=g)
got = enc(x.cuda())
for f in got: f.retain_grad()
cots = [torch.randn(f.shape, generator=g) / f[0].numel() ** 0.5 for f in got]
(got[4] * cots[4].cuda()).sum().backward()
st = {k: (v.double().requires_grad_() if v.is_floating_point() and "running" not in k else (v.double() if v.is_floating_point() else v)) for k, v in state.items()}
feats = RR.resnet_encoder_forward(st, x.double(), 18, True)
for f in feats: f.retain_grad()
(feats[4] * cots[4].double()).sum().backward()
print("d(feature4) %.2e  d(feature3) %.3e  d(feature2) %.3e" % (rel_l2(got[4].grad, feats[4].grad), rel_l2(got[3].grad, feats[3].grad), rel_l2(got[2].grad, feats[2].grad)))
# rebuild d(feature3) from the HIP run's own output gradients of layer4.0's two convolutions, with torch fp64
c15, x15, y15 = rec[15]; c16, x16, y16 = rec[16]
assert x15 is got[3] and x16 is got[3], "conv order changed"
xr = got[3].detach().double().cpu().requires_grad_()
(F.conv2d(xr, c15.weight.detach().double().cpu(), None, c15.stride, c15.padding) * y15.grad.double().cpu()).sum().backward()
(F.conv2d(xr, c16.weight.detach().double().cpu(), None, c16.stride, c16.padding) * y16.grad.double().cpu()).sum().backward()
print("rebuilt-from-HIP-pieces vs HIP accumulated: %.3e   vs oracle: %.3e" % (rel_l2(got[3].grad, xr.grad), rel_l2(xr.grad, feats[3].grad)))
print("|d f3| hip %.6e oracle %.6e  ratio %.6f" % (float(got[3].grad.norm()), float(feats[3].grad.norm()), float(got[3].grad.norm()) / float(feats[3].grad.norm())))
d = (got[3].grad.double().cpu() - feats[3].grad)
print("error: rel %.3e; fraction of elements with |err| > 1e-3 |max|: %.4f; max err %.3e at %s" % (float(d.norm() / feats[3].grad.norm()),
      float((d.abs() > 1e-3 * feats[3].grad.abs().max()).double().mean()), float(d.abs().max()), tuple(int(v) for v in (d.abs() == d.abs().max()).nonzero()[0])))
# ---- layer3 convolutions re-run in isolation on the recorded tensors
from depthcore import ops
for i in (10, 11, 12, 13, 14):
    conv, xx, yy = rec[i]
    xh = xx.detach().clone().requires_grad_(); wh = conv.weight.detach().clone().requires_grad_()
    yh = RE._conv.__wrapped__(conv, xh) if hasattr(RE._conv, "__wrapped__") else orig(types.SimpleNamespace(kernel_size=conv.kernel_size, stride=conv.stride, padding=conv.padding, dilation=conv.dilation, groups=conv.groups, bias=None, weight=wh, in_channels=conv.in_channels, out_channels=conv.out_channels), xh)
    yh.backward(yy.grad)
    xr = xx.detach().double().cpu().requires_grad_(); wr = conv.weight.detach().double().cpu().requires_grad_()
    F.conv2d(xr, wr, None, conv.stride, conv.padding).backward(yy.grad.double().cpu())
    print("conv %d k%d s%d isolated re-run: dX %.3e dW %.3e" % (i, conv.kernel_size[0], conv.stride[0], rel_l2(xh.grad, xr.grad), rel_l2(wh.grad, wr.grad)))
# and the two residual BN calls of layer3: dres
