"""Fused training-mode BatchNorm (+ residual) (+ ReLU) kernels (dc_bn_relu_fwd/bwd) vs plain torch on the CPU."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from helpers import close, rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("shape,with_res,relu", [((3, 8, 10, 12), False, True), ((2, 64, 24, 40), True, True),
                                                 ((4, 16, 7, 9), True, False), ((12, 64, 96, 320), False, True),
                                                 ((2, 5, 3, 5), False, False)])
def test_bn_relu_vs_torch(shape, with_res, relu):
    from depthcore import ops
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(shape, generator=g) * 1.7 + 0.3
    res = torch.randn(shape, generator=g) if with_res else None
    cot = torch.randn(shape, generator=g)
    C = shape[1]
    bn_ref, bn_hip = nn.BatchNorm2d(C), nn.BatchNorm2d(C).to(DEV)
    with torch.no_grad():
        bn_ref.weight.copy_(torch.rand(C, generator=g) + 0.5); bn_ref.bias.copy_(torch.randn(C, generator=g))
        bn_hip.weight.copy_(bn_ref.weight); bn_hip.bias.copy_(bn_ref.bias)
    xr = x.clone().requires_grad_()
    rr = res.clone().requires_grad_() if with_res else None
    y = bn_ref(xr)
    if with_res:
        y = y + rr
    if relu:
        y = F.relu(y)
    leaves = [xr, bn_ref.weight, bn_ref.bias] + ([rr] if with_res else [])
    gr = torch.autograd.grad((y * cot).sum(), leaves)
    xh = x.to(DEV).requires_grad_()
    rh = res.to(DEV).requires_grad_() if with_res else None
    yh = ops.bn_relu(xh, bn_hip, rh, relu)
    lh = [xh, bn_hip.weight, bn_hip.bias] + ([rh] if with_res else [])
    gh = torch.autograd.grad((yh * cot.to(DEV)).sum(), lh)
    close(yh, y, rtol=1e-4, atol=1e-5)
    close(bn_hip.running_mean, bn_ref.running_mean, rtol=1e-5, atol=1e-6)
    close(bn_hip.running_var, bn_ref.running_var, rtol=1e-4, atol=1e-6)
    for a, b, name in zip(gh, gr, ["dx", "dgamma", "dbeta", "dres"]):
        assert rel_l2(a, b) < 2e-4, (name, rel_l2(a, b))


def test_encoder_matches_oracle_resnet():
    """networks.ResnetEncoder (fused BN/ReLU kernels + MIOpen convs) vs the functional CPU ResNet of the oracle."""
    import networks
    from oracle.resnet_ref import resnet_encoder_forward
    torch.manual_seed(0)
    enc = networks.ResnetEncoder(18, False).to(DEV)
    enc.train()
    state = {k: v.detach().cpu().clone() for k, v in enc.state_dict().items()}
    x = torch.rand(2, 3, 64, 96, generator=torch.Generator().manual_seed(1))
    want = resnet_encoder_forward(state, x, 18, training=True)
    got = enc(x.to(DEV))
    for a, b in zip(got, want):
        close(a, b, rtol=2e-3, atol=2e-4)
    assert int(enc.encoder.bn1.num_batches_tracked) == 1 and int(enc.encoder.layer4[1].bn2.num_batches_tracked) == 1


@pytest.mark.parametrize("shape", [(2, 5, 12, 16), (1, 3, 9, 11), (12, 64, 96, 320)])
def test_maxpool_vs_torch(shape):
    from depthcore import ops
    g = torch.Generator().manual_seed(7)
    x = F.relu(torch.randn(shape, generator=g))          # post-ReLU input: many exact ties at 0
    xr = x.clone().requires_grad_()
    y = F.max_pool2d(xr, 3, 2, 1)
    cot = torch.randn(y.shape, generator=g)
    (gr,) = torch.autograd.grad((y * cot).sum(), [xr])
    xh = x.to(DEV).requires_grad_()
    yh = ops.maxpool3x3s2(xh)
    (gh,) = torch.autograd.grad((yh * cot.to(DEV)).sum(), [xh])
    assert torch.equal(yh.cpu(), y.detach())
    assert torch.equal(gh.cpu(), gr)                       # same tie-break as ATen -> identical routing


def test_grouped_bn_equals_separate_calls():
    """bn_groups=2 on a stacked batch == two sequential calls (outputs, grads, running statistics)."""
    import networks
    torch.manual_seed(0)
    a = networks.ResnetEncoder(18, False, num_input_images=2).to(DEV)
    b = networks.ResnetEncoder(18, False, num_input_images=2).to(DEV)
    b.load_state_dict(a.state_dict())
    a.train(); b.train()
    g = torch.Generator(device=DEV).manual_seed(3)
    x1 = torch.rand(2, 6, 64, 96, device=DEV, generator=g)
    x2 = torch.rand(2, 6, 64, 96, device=DEV, generator=g)
    fa = a(torch.cat([x1, x2], 0), bn_groups=2)
    f1 = [t.clone() for t in b(x1)]
    f2 = b(x2)
    for s, u, v in zip(fa, f1, f2):
        close(s[:2], u, rtol=1e-4, atol=3e-5)       # (MIOpen may pick another conv algorithm at batch 4 vs 2)
        close(s[2:], v, rtol=1e-4, atol=3e-5)
    (fa[-1].square().sum()).backward()
    (f1[-1].square().sum() + f2[-1].square().sum()).backward()
    for (n, p), (_, q) in zip(a.named_parameters(), b.named_parameters()):
        if p.grad is not None:
            assert rel_l2(p.grad, q.grad) < 1e-3, n
    sa, sb = a.state_dict(), b.state_dict()
    for k in sa:
        if "running" in k:
            close(sa[k], sb[k], rtol=1e-4, atol=1e-6)
        if "num_batches_tracked" in k:
            assert int(sa[k]) == int(sb[k]) == 2
