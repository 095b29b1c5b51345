#!/usr/bin/env python3
"""Debug probe (GPU): gradients at the input and output of every BatchNorm call, HIP vs fp64 oracle (aligned by value)."""
import os, sys
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
hrec = []
orig = RE._ops.bn_relu
def wrap(x, bn, res=None, relu=True, groups=1):
    x.retain_grad(); y = orig(x, bn, res, relu, groups); y.retain_grad(); hrec.append((x, y)); return y
RE._ops.bn_relu = wrap
g = torch.Generator().manual_seed(1)
x = torch.rand(4, 3, 64, 128, generator=g)
got = enc(x.cuda())
cots = [torch.randn(f.shape, generator=g) / f[0].numel() ** 0.5 for f in got]
(got[4] * cots[4].cuda()).sum().backward()
orec = []
obn = RR._bn
def owrap(xx, st, prefix, training, eps=1e-5):
    xx.retain_grad(); y = obn(xx, st, prefix, training, eps); y.retain_grad(); orec.append((xx, y, prefix)); return y
RR._bn = owrap
st = {k: (v.double().requires_grad_() if v.is_floating_point() and "running" not in k else (v.double() if v.is_floating_point() else v)) for k, v in state.items()}
feats = RR.resnet_encoder_forward(st, x.double(), 18, True)
(feats[4] * cots[4].double()).sum().backward()
used = set()
for i, (hx, hy) in enumerate(hrec):
    j = min((j for j in range(len(orec)) if j not in used and orec[j][0].shape == hx.shape), key=lambda j: rel_l2(hx, orec[j][0]))
    used.add(j)
    ox, oy, name = orec[j]
    print("%-22s d(bn in) %.3e   (bn-out grads are pre-relu in the oracle, post in HIP: not comparable)" % (name, rel_l2(hx.grad, ox.grad)))
# ---- conditioning: the same fp64 torch BN backward on HIP's input x vs on the oracle's input x (same gy)
print("--- BN-backward sensitivity to the 1e-6 forward differences")
used = set()
for i, (hx, hy) in enumerate(hrec):
    j = min((j for j in range(len(orec)) if j not in used and orec[j][0].shape == hx.shape), key=lambda j: rel_l2(hx, orec[j][0]))
    used.add(j)
    ox, oy, name = orec[j]
    if not name.startswith("layer3"): continue
    gy = torch.randn(ox.shape, dtype=torch.float64, generator=torch.Generator().manual_seed(i))
    res = []
    for xin in (hx.detach().double().cpu(), ox.detach()):
        xr = xin.clone().requires_grad_()
        F.batch_norm(xr, None, None, None, None, True, 0.1, 1e-5).backward(gy)
        res.append(xr.grad)
    v = ox.detach().var((0, 2, 3), unbiased=False); m = ox.detach().mean((0, 2, 3))
    print("%-22s x err %.2e  -> dx differs %.3e   min var %.3e  max m^2/var %.1f" % (name, rel_l2(hx, ox), rel_l2(res[0], res[1]), float(v.min()), float((m * m / v).max())))
print("--- structure of the dx error at layer3.1.bn2")
for i, (hx, hy) in enumerate(hrec):
    pass
hx, hy = hrec[14]
ox = [o for o in orec if o[2] == "layer3.1.bn2"][0][0]
d = hx.grad.double().cpu() - ox.grad
ref = ox.grad
print("rel err %.3e" % float(d.norm() / ref.norm()))
per_c = d.pow(2).sum((0, 2, 3)).sqrt() / ref.pow(2).sum((0, 2, 3)).sqrt()
print("per-channel rel err: median %.2e  max %.2e  #channels > 1e-3: %d of %d" % (float(per_c.median()), float(per_c.max()), int((per_c > 1e-3).sum()), per_c.numel()))
per_n = d.pow(2).sum((1, 2, 3)).sqrt() / ref.pow(2).sum((1, 2, 3)).sqrt()
print("per-sample rel err:", [float("%.2e" % v) for v in per_n])
c = int(per_c.argmax())
dd, rr = d[:, c].flatten(), ref[:, c].flatten()
xh = (ox.detach()[:, c] - ox.detach()[:, c].mean()) / ox.detach()[:, c].std(unbiased=False)
A = torch.stack([torch.ones_like(rr), xh.flatten(), rr], 1)
sol = torch.linalg.lstsq(A, dd.unsqueeze(1)).solution.flatten()
print("worst channel %d: err ~ %.3e * 1 + %.3e * xhat + %.3e * ref ; residual %.2e of %.2e" % (c, sol[0], sol[1], sol[2], float((A @ sol - dd).norm()), float(dd.norm())))
