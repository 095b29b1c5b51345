"""a1 at network level: networks.ResnetEncoder (resnet18 and resnet50, bn_groups 1 and 2) forward AND backward on the HIP
path against the oracle's functional ResNet (oracle/resnet_ref.py) evaluated in fp64.

Every feature map and every parameter gradient are compared by relative L2 norm (the gradient of `conv1.weight` sits
behind the data gradients of all other layers; the input image itself never needs a gradient on this path -- the stem's
kernels have none, see dc_convs2_dgrad).  The bound is
stated against the fp64 result and calibrated in the test itself: the HIP path may be at most 4x as far from fp64 as
the oracle's own fp32 evaluation is (training-mode BatchNorm on small maps amplifies rounding differences -- the fp32
oracle itself is 0.4 % off fp64 on some layer3 BatchNorm gradients at these sizes -- so a fixed number would either be
loose for the stem or flaky for layer4), with a floor of 5e-5.

A ReLU network is only piecewise smooth: when one pre-activation lies within fp32 rounding of zero, the fp32 and the fp64
evaluation route that element's gradient differently, and on these small test tensors a single such element moves every
upstream gradient by ~0.5 % (found with tools/debug/bisect_probe.py: one channel of one BatchNorm input carried the whole
difference while every kernel, re-run in isolation on the recorded tensors, agreed with torch to 1e-7).  Each
configuration is therefore evaluated on three inputs: at least two must meet the calibrated bound everywhere, and on the
third no entry that misses it may exceed 3e-2."""
import pytest
import torch

from helpers import rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _oracle(state, x, cots, num_layers, groups, dtype):
    from oracle.resnet_ref import resnet_encoder_forward
    st = {k: (v.to(dtype).requires_grad_() if v.is_floating_point() and "running" not in k else
              (v.to(dtype) if v.is_floating_point() else v)) for k, v in state.items()}
    xr = x.to(dtype).requires_grad_()
    n = x.shape[0] // groups
    parts = [resnet_encoder_forward(st, xr[g * n:(g + 1) * n], num_layers, training=True) for g in range(groups)]
    feats = [torch.cat([p[i] for p in parts], 0) for i in range(5)]
    loss = sum((f * c.to(dtype)).sum() for f, c in zip(feats, cots))
    names = [k for k, v in st.items() if v.requires_grad and ".fc." not in k]
    grads = torch.autograd.grad(loss, [st[k] for k in names] + [xr])
    return feats, dict(zip(names, grads[:-1])), grads[-1]


@pytest.mark.parametrize("num_layers,groups,nimg,B,H,W", [
    (18, 1, 1, 4, 64, 128),
    (18, 2, 2, 4, 64, 128),          # the pose encoder's stacked pairs (6 channels, per-pair BN statistics)
    (50, 1, 1, 4, 64, 128),
    (50, 2, 2, 4, 96, 128),
    (18, 1, 1, 1, 192, 640),         # BASELINE configs[0] shape (C1: a single 192x640 frame)
])
def test_encoder_forward_and_all_gradients_vs_fp64_oracle(num_layers, groups, nimg, B, H, W):
    results = [_one_input(num_layers, groups, nimg, B, H, W, seed) for seed in (1, 2, 3)]
    clean = sum(1 for bad, worst in results if not bad)
    assert clean >= 2, [(len(bad), sorted(bad, key=lambda t: -t[1])[:4]) for bad, worst in results]
    # the cap applies to entries that MISS the calibrated bound (an entry inside it is as close to fp64 as torch's own fp32)
    assert max([e for bad, worst in results for _, e, _ in bad] or [0.0]) < 3e-2, [bad for bad, worst in results]


def _one_input(num_layers, groups, nimg, B, H, W, seed):
    import networks
    torch.manual_seed(0)
    enc = networks.ResnetEncoder(num_layers, False, num_input_images=nimg).to(DEV)
    enc.train()
    state = {k: v.detach().cpu().clone() for k, v in enc.state_dict().items()}
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(B, 3 * nimg, H, W, generator=g)
    xh = x.to(DEV)
    got = enc(xh, bn_groups=groups)
    cots = [torch.randn(f.shape, generator=g) / f[0].numel() ** 0.5 for f in got]
    loss = sum((f * c.to(DEV)).sum() for f, c in zip(got, cots))
    params = {"encoder." + n: p for n, p in enc.encoder.named_parameters() if not n.startswith("fc.")}
    gh_p = dict(zip(params.keys(), torch.autograd.grad(loss, list(params.values()))))

    f64, g64, gx64 = _oracle(state, x, cots, num_layers, groups, torch.float64)
    f32, g32, gx32 = _oracle(state, x, cots, num_layers, groups, torch.float32)

    def bound(e32):
        return max(4.0 * e32, 5e-5)

    worst = worst32 = 0.0
    bad = []
    for i in range(5):
        e, e32 = rel_l2(got[i], f64[i]), rel_l2(f32[i], f64[i])
        if e > bound(e32):
            bad.append(("feature %d" % i, e, e32))
    assert set(g64) == set(gh_p)
    for k in g64:
        e, e32 = rel_l2(gh_p[k], g64[k]), rel_l2(g32[k], g64[k])
        worst, worst32 = max(worst, e), max(worst32, e32)
        if e > bound(e32):
            bad.append((k, e, e32))
    return bad, worst
