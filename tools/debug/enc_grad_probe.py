#!/usr/bin/env python3
"""Debug probe (GPU): per-convolution output gradients of the HIP ResnetEncoder vs the fp64 oracle, to locate a gradient error."""
import os, sys, types
import torch
import torch.nn.functional as F
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "self-supervised-depth-estimation_amd")); sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import networks
from networks import resnet_encoder as RE
from oracle import resnet_ref as RR
from helpers import rel_l2

L, B, H, W = int(sys.argv[1]) if len(sys.argv) > 1 else 18, 4, 64, 128
torch.manual_seed(0)
enc = networks.ResnetEncoder(L, False).cuda(); enc.train()
state = {k: v.detach().cpu().clone() for k, v in enc.state_dict().items()}
g = torch.Generator().manual_seed(1)
x = torch.rand(B, 3, H, W, generator=g)
hip_outs = []
orig_conv = RE._conv
def wrap(conv, xx):
    y = orig_conv(conv, xx); y.retain_grad(); hip_outs.append(y); return y
RE._conv = wrap
from depthcore import ops as _o
hip_mp = []
omp = _o.maxpool3x3s2
def mpw(t):
    t.retain_grad(); y = omp(t); y.retain_grad(); hip_mp.extend([t, y]); return y
RE._ops.maxpool3x3s2 = mpw
got = enc(x.cuda())
cots = [torch.randn(f.shape, generator=g) / f[0].numel() ** 0.5 for f in got]
loss = sum((f * c.cuda()).sum() for f, c in zip(got, cots))
loss.backward()
ref_outs = []
oconv = F.conv2d
def rwrap(*a, **k):
    y = oconv(*a, **k); y.retain_grad(); ref_outs.append(y); return y
RR.F = types.SimpleNamespace(**{n: getattr(F, n) for n in dir(F) if not n.startswith("__")}); RR.F.conv2d = rwrap
ref_mp = []
def rmp(t, *a, **k):
    t.retain_grad(); y = F.max_pool2d(t, *a, **k); y.retain_grad(); ref_mp.extend([t, y]); return y
RR.F.max_pool2d = rmp
st = {k: (v.double().requires_grad_() if v.is_floating_point() and "running" not in k else (v.double() if v.is_floating_point() else v)) for k, v in state.items()}
feats = RR.resnet_encoder_forward(st, x.double(), L, True)
l2 = sum((f * c.double()).sum() for f, c in zip(feats, cots)); l2.backward()
print("convs", len(hip_outs), len(ref_outs))
print("d(bn1 out) %.3e   d(maxpool out) %.3e   maxpool out %.2e" % (rel_l2(hip_mp[0].grad, ref_mp[0].grad), rel_l2(hip_mp[1].grad, ref_mp[1].grad), rel_l2(hip_mp[1], ref_mp[1])))
# maxpool backward alone: HIP kernel on the ORACLE's d(maxpool out)
xin = hip_mp[0].detach().clone().requires_grad_()
yy = omp(xin); yy.backward(ref_mp[1].grad.float().cuda())
d_mp_only = ref_mp[0].grad - cots[0].double()          # oracle: gradient that reaches bn1 out through the maxpool
print("HIP maxpool bwd on oracle grad vs oracle: %.3e" % rel_l2(xin.grad, d_mp_only))
t = ref_mp[0].detach().clone().requires_grad_(); F.max_pool2d(t, 3, 2, 1).backward(ref_mp[1].grad)
print("torch-CPU maxpool bwd (fp64) vs oracle path: %.3e;  ties in windows: input zeros %.3f" % (rel_l2(t.grad, d_mp_only), float((ref_mp[0] == 0).double().mean())))
used = set()
for i, a in enumerate(hip_outs):          # align by forward value (the two sides call a block's convolutions in different orders)
    j = min((j for j in range(len(ref_outs)) if j not in used and ref_outs[j].shape == a.shape), key=lambda j: rel_l2(a, ref_outs[j]))
    used.add(j)
    b = ref_outs[j]
    print("conv %2d <-> %2d" % (i, j), tuple(a.shape), "out %.2e  dOut %.3e  |dOut| %.3e" % (rel_l2(a, b), rel_l2(a.grad, b.grad), float(b.grad.norm())))
names = [n for n, p in enc.named_parameters() if p.grad is not None]
errs = sorted(((rel_l2(dict(enc.named_parameters())[n].grad, st[n].grad), n) for n in names if n in st), reverse=True)[:8]
print("worst parameter gradients:", [("%.2e" % e, n) for e, n in errs])
for i in range(5):
    print("feature %d: value err %.2e" % (i, rel_l2(got[i], feats[i])))
wg, wr = enc.encoder.conv1.weight.grad, st["encoder.conv1.weight"].grad
print("conv1.weight grad err %.3e" % rel_l2(wg, wr))
# wgrad of the stem from the ORACLE's dOut through the HIP kernel and through torch fp64
from depthcore import ops
x0 = ((x - 0.45) / 0.225)
xw = x0.cuda(); ww = enc.encoder.conv1.weight.detach().clone().requires_grad_()
y = ops.conv_s2(xw, ww); y.backward(ref_outs[0].grad.float().cuda())
print("HIP stem wgrad with oracle dOut vs oracle: %.3e" % rel_l2(ww.grad, wr))
y2 = ops.conv_s2(xw, ww); ww.grad = None; y2.backward(hip_outs[0].grad)
print("HIP stem wgrad with HIP dOut vs model's: %.3e" % rel_l2(ww.grad, wg))
