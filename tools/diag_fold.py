import sys, os, torch, torch.nn as nn, torch.nn.functional as F
sys.path[:0] = ["/root/repo/self-supervised-depth-estimation_amd", "/root/repo/tests"]
from depthcore import bnfold
DEV = "cuda:0"
B, Ci, Cm, Co, H, W, groups = 2, 128, 128, 512, 24, 40, 1
g = torch.Generator().manual_seed(B * 1000 + Ci + Co + H)
x = torch.randn(B, Ci, H, W, generator=g)
wa = torch.randn(Cm, Ci, 1, 1, generator=g) / Ci ** 0.5
wb = torch.randn(Co, Cm, 1, 1, generator=g) / Cm ** 0.5
cot = torch.randn(B, Co, H, W, generator=g)
def _bn(C, g):
    bn = nn.BatchNorm2d(C)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(C, generator=g) + 0.5); bn.bias.copy_(torch.randn(C, generator=g) * 0.3)
    return bn
bn1r, bn2r = _bn(Cm, g), _bn(Co, g)
bn1h, bn2h = nn.BatchNorm2d(Cm).to(DEV), nn.BatchNorm2d(Co).to(DEV)
bn1h.load_state_dict(bn1r.state_dict()); bn2h.load_state_dict(bn2r.state_dict())
xr, war, wbr = x.double().requires_grad_(), wa.double().requires_grad_(), wb.double().requires_grad_()
bn1r.double(); bn2r.double()
pre1 = bn1r(F.conv2d(xr, war))
yr = F.relu(bn2r(F.conv2d(F.relu(pre1), wbr)))
gr = torch.autograd.grad((yr * cot.double()).sum(), [xr, war, wbr])
xh = x.to(DEV).requires_grad_(); wah, wbh = wa.to(DEV).requires_grad_(), wb.to(DEV).requires_grad_()
ya, sa = bnfold.conv1x1(xh, wah, 1, groups)
yb, sb = bnfold.conv1x1(ya, wbh, 1, groups, in_bn=bn1h, in_stats=sa)
yh = bnfold.bn_apply(yb, bn2h, sb, groups=groups)
gh = torch.autograd.grad((yh * cot.to(DEV)).sum(), [xh, wah, wbh])
d = (gh[0].cpu().double() - gr[0]).abs()
per_pix = d.amax(1)   # (B,H,W)
thr = 1e-3 * gr[0].abs().max()
print("x3 =", os.environ.get("DC_G1_X3"), "rel_l2 dx", float((gh[0].cpu().double()-gr[0]).norm()/gr[0].norm()), "pixels with error:", int((per_pix > thr).sum()), "of", per_pix.numel())
print("smallest |pre-activation bn1| :", float(pre1.abs().min()), " count < 1e-5:", int((pre1.abs() < 1e-5).sum()))
bad = (per_pix > thr).nonzero()
for b_, y_, x_ in bad[:5].tolist():
    print("  pixel", b_, y_, x_, "min |pre1| there:", float(pre1[b_, :, y_, x_].abs().min()))
