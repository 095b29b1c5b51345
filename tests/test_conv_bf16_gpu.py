"""The bf16 matrix-core convolutions (csrc/conv_bf16.hip; BASELINE configs[4] "reduced-precision networks, fp32 loss").

Their contract is exact and checked exactly: operands rounded to bf16 (round to nearest even), products and sums in fp32.
The oracle therefore rounds the operands the same way and convolves in fp64; what is left is fp32 summation order
(rel. 1e-6), so a wrong tap, channel, padding or transposed-read mapping cannot hide behind a "bf16 tolerance".  A second
check puts the bf16 results next to the fp32 kernels' (the difference is the operand rounding, ~2^-9 per factor)."""
import pytest
import torch
import torch.nn.functional as F

from helpers import rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rb(t):
    return t.to(torch.bfloat16).to(torch.float64)


def _act(v, act):
    from depthcore import ops
    return {ops.ACT_NONE: lambda z: z, ops.ACT_ELU: F.elu, ops.ACT_SIGMOID: torch.sigmoid, ops.ACT_RELU: F.relu,
            ops.ACT_TANH: torch.tanh}[act](v)


def _act_bwd(y, act):
    from depthcore import ops
    if act == ops.ACT_ELU:
        return torch.where(y > 0, torch.ones_like(y), y + 1)
    if act == ops.ACT_SIGMOID:
        return y * (1 - y)
    if act == ops.ACT_RELU:
        return (y > 0).to(y.dtype)
    if act == ops.ACT_TANH:
        return 1 - y * y
    return torch.ones_like(y)


CASES = [
    # B, C0, up0, C1, Co, H, W, act, pad, bias
    (2, 64, 0, 0, 64, 24, 32, "none", "zero", False),       # ResNet trunk shape: MR = 4, two chunks
    (1, 128, 0, 0, 32, 16, 48, "none", "zero", False),      # MR = 2, four chunks
    (2, 16, 0, 0, 16, 32, 64, "elu", "reflect", True),      # half-filled chunk, MR = 1 (decoder level 0)
    (2, 32, 1, 64, 32, 32, 48, "elu", "reflect", True),     # upsample + concat (decoder level 1: 32 + 64 channels)
    (1, 64, 1, 64, 64, 20, 32, "elu", "reflect", True),     # H not a multiple of the tile
    (2, 32, 0, 0, 80, 16, 16, "relu", "zero", True),        # Co not a multiple of the channel block (pose-decoder style)
    (1, 4, 0, 0, 4, 16, 32, "tanh", "zero", True),          # Fusion_v3's tiny convolutions
    (2, 256, 0, 0, 128, 12, 40, "elu", "reflect", True),    # W % 16 != 0: the last tile hangs over the border (192x640 pyramid)
    (2, 128, 1, 128, 64, 12, 40, "elu", "reflect", True),   # ... with the upsampled half read at W / 2 = 20
    (1, 512, 0, 0, 64, 6, 20, "none", "zero", False),       # the deepest trunk map
]


@pytest.mark.parametrize("case", CASES, ids=[str(c) for c in CASES])
def test_block_forward_and_backward_exact_vs_rounded_operand_oracle(case):
    from depthcore import ops
    B, C0, up0, C1, Co, H, W, act, pad, bias = case
    act_, pad_ = getattr(ops, "ACT_" + act.upper()), getattr(ops, "PAD_" + pad.upper())
    g = torch.Generator().manual_seed(hash(case) % 1000)
    x0 = torch.randn(B, C0, H >> up0, W >> up0, generator=g)
    x1 = torch.randn(B, C1, H, W, generator=g) if C1 else None
    w = torch.randn(Co, C0 + C1, 3, 3, generator=g) / (3.0 * (C0 + C1) ** 0.5)
    b = 0.1 * torch.randn(Co, generator=g) if bias else None
    gy = torch.randn(B, Co, H, W, generator=g)
    dv = lambda t: None if t is None else t.to(DEV).requires_grad_()
    hx0, hx1, hw, hb = dv(x0), dv(x1), dv(w), dv(b)
    with ops.matrix_precision("bf16"):
        y = ops.conv3x3_block(hx0, hx1, hw, hb, bool(up0), act_, pad_)
    leaves = [t for t in (hx0, hx1, hw, hb) if t is not None]
    grads = dict(zip([n for n, t in zip(("x0", "x1", "w", "b"), (hx0, hx1, hw, hb)) if t is not None],
                     torch.autograd.grad(y, leaves, gy.to(DEV))))

    # oracle on the rounded operands, fp64
    xc = F.interpolate(x0.double(), scale_factor=2, mode="nearest") if up0 else x0.double()
    if C1:
        xc = torch.cat([xc, x1.double()], 1)
    xp = F.pad(rb(xc), (1, 1, 1, 1), mode="reflect" if pad == "reflect" else "constant")
    pre = F.conv2d(xp, rb(w)) + (b.double().view(1, -1, 1, 1) if bias else 0)
    want = _act(pre, act_)
    assert rel_l2(y, want) < 2e-6, rel_l2(y, want)
    gp = gy.double() * _act_bwd(y.detach().cpu().double(), act_)          # the kernels form g' in fp32 from their own y
    gpf = (gy.to(DEV) * _act_bwd(y.detach(), act_)).cpu()                  # ... exactly this fp32 tensor
    gpr = rb(gpf)
    # data gradient: g' (rounded) against the rounded filter, over the padded domain, folded back by autograd
    xpad = torch.zeros_like(xp).requires_grad_()
    (dxp,) = torch.autograd.grad(F.conv2d(xpad, rb(w)), xpad, gpr)
    xc_leaf = xc.clone().requires_grad_()
    (dxc,) = torch.autograd.grad(F.pad(xc_leaf, (1, 1, 1, 1), mode="reflect" if pad == "reflect" else "constant"), xc_leaf, dxp)
    dx0 = dxc[:, :C0]
    if up0:
        dx0 = dx0.reshape(B, C0, H // 2, 2, W // 2, 2).sum((3, 5))
    assert rel_l2(grads["x0"], dx0) < 5e-6, rel_l2(grads["x0"], dx0)
    if C1:
        assert rel_l2(grads["x1"], dxc[:, C0:]) < 5e-6
    # weight gradient: rounded input patch against rounded g' -- from 16 output channels (round 5: the thin 16 / 32-channel
    # levels take the bf16 weight-gradient kernel too, a partly filled 64-channel tile still beats the fp32 direct kernel 2.7x);
    # below that the fp32 direct kernel, i.e. unrounded operands
    wl = torch.zeros(Co, C0 + C1, 3, 3, dtype=torch.float64, requires_grad=True)
    if Co >= 16:
        (dw,) = torch.autograd.grad(F.conv2d(xp, wl), wl, gpr)
    else:
        xp32 = F.pad(xc, (1, 1, 1, 1), mode="reflect" if pad == "reflect" else "constant")
        (dw,) = torch.autograd.grad(F.conv2d(xp32, wl), wl, gpf.double())
    assert rel_l2(grads["w"], dw) < 5e-6, rel_l2(grads["w"], dw)
    if bias:
        assert rel_l2(grads["b"], gp.sum((0, 2, 3))) < 5e-6          # the bias gradient sums the unrounded fp32 g'

    # next to the fp32 kernels: the difference is the operand rounding
    y32 = ops.conv3x3_block(hx0, hx1, hw, hb, bool(up0), act_, pad_)
    g32 = torch.autograd.grad(y32, leaves, gy.to(DEV))
    assert 1e-5 < rel_l2(y, y32) < 1e-2
    for a, c in zip(grads.values(), g32):      # (ReLU: the two outputs disagree on the sign of a few pre-activations)
        assert rel_l2(a, c) < (1e-1 if act == "relu" else 2e-2)


@pytest.mark.parametrize("shape", [(2, 64, 64, 24, 32), (1, 128, 256, 12, 16), (2, 512, 512, 6, 16), (2, 256, 256, 12, 40), (1, 512, 512, 6, 20)])
def test_trunk_entry_points_under_bf16(shape):
    """dc_wino3x3_fwd / _dgrad / _wgrad (the ResNet trunks' stride-1 3x3) route to the bf16 kernels under the policy."""
    from depthcore import ops
    B, Ci, Co, H, W = shape
    g = torch.Generator().manual_seed(Ci + Co)
    x = torch.randn(B, Ci, H, W, generator=g)
    w = torch.randn(Co, Ci, 3, 3, generator=g) / (3.0 * Ci ** 0.5)
    gy = torch.randn(B, Co, H, W, generator=g)
    hx, hw = x.to(DEV).requires_grad_(), w.to(DEV).requires_grad_()
    with ops.matrix_precision("bf16"):
        y = ops.wino_conv3x3(hx, hw)
    gx, gw = torch.autograd.grad(y, [hx, hw], gy.to(DEV))      # (backward outside the context: the op remembers its precision)
    xr, wr = rb(x).requires_grad_(), rb(w).requires_grad_()
    want = F.conv2d(xr, wr, padding=1)
    assert rel_l2(y, want) < 2e-6
    (wx,) = torch.autograd.grad(F.conv2d(xr, rb(w), padding=1), xr, rb(gy))
    (ww,) = torch.autograd.grad(F.conv2d(rb(x), wr, padding=1), wr, rb(gy))
    assert rel_l2(gx, wx) < 5e-6 and rel_l2(gw, ww) < 5e-6, (rel_l2(gx, wx), rel_l2(gw, ww))
    y32 = ops.wino_conv3x3(hx, hw)
    assert 1e-5 < rel_l2(y, y32) < 1e-2


def test_trunk_data_gradient_addend_under_bf16():
    """The residual fork's addend rides in the bf16 kernel's store epilogue (c3b_conv_kernel: acc + addend, one rounding, as the
    separate add pass it replaces): with a GradFork the gradient of x is bitwise conv-gradient + parked gradient."""
    from depthcore import ops
    g = torch.Generator().manual_seed(11)
    for B, C, H, W in ((2, 64, 24, 32), (1, 128, 12, 40)):
        x = torch.randn(B, C, H, W, generator=g).to(DEV).requires_grad_()
        w = (torch.randn(C, C, 3, 3, generator=g) / (3.0 * C ** 0.5)).to(DEV)
        gy, skip = torch.randn(B, C, H, W, generator=g).to(DEV), torch.randn(B, C, H, W, generator=g).to(DEV)
        with ops.matrix_precision("bf16"):
            (plain,) = torch.autograd.grad(ops.wino_conv3x3(x, w), x, gy)
            fork = ops.GradFork()
            y = ops.wino_conv3x3(x, w, fork)
        fork.park(skip)                       # what the block's last BatchNorm would leave for conv1
        (summed,) = torch.autograd.grad(y, x, gy)
        assert torch.equal(summed, plain + skip)


def test_bf16_block_is_deterministic_at_full_size():
    """Decoder level 1 at the BASELINE size (B = 12, 32 + 64 channels, 96 x 320): bitwise reproducible forward and gradients."""
    from depthcore import ops
    g = torch.Generator(device=DEV).manual_seed(0)
    x0 = torch.randn(12, 32, 48, 160, device=DEV, generator=g).requires_grad_()
    x1 = torch.randn(12, 64, 96, 320, device=DEV, generator=g).requires_grad_()
    w = (0.05 * torch.randn(32, 96, 3, 3, device=DEV, generator=g)).requires_grad_()
    b = torch.zeros(32, device=DEV).requires_grad_()
    gy = torch.randn(12, 32, 96, 320, device=DEV, generator=g)
    res = []
    for _ in range(2):
        with ops.matrix_precision("bf16"):
            y = ops.conv3x3_block(x0, x1, w, b, True, ops.ACT_ELU, ops.PAD_REFLECT)
        res.append([y.detach().clone()] + [t.clone() for t in torch.autograd.grad(y, [x0, x1, w, b], gy)])
    for a, c in zip(*res):
        assert torch.equal(a, c) and torch.isfinite(a).all()


@pytest.mark.parametrize("shape", [(2, 64, 128, 32, 64), (1, 128, 256, 24, 32), (2, 256, 512, 8, 32), (2, 128, 256, 24, 80), (1, 256, 512, 12, 40)])
def test_stride2_3x3_under_bf16(shape):
    """The trunk's 3x3 / 2 convolutions (dc_convs2_*): forward and weight gradient are the stride-2 instantiation of the
    bf16 kernels, the data gradient is the stride-1 kernel over the DILATED output gradient (zeros between the samples)."""
    from depthcore import ops
    B, Ci, Co, H, W = shape
    g = torch.Generator().manual_seed(Ci * 3 + Co)
    x = torch.randn(B, Ci, H, W, generator=g)
    w = torch.randn(Co, Ci, 3, 3, generator=g) / (3.0 * Ci ** 0.5)
    gy = torch.randn(B, Co, H // 2, W // 2, generator=g)
    hx, hw = x.to(DEV).requires_grad_(), w.to(DEV).requires_grad_()
    with ops.matrix_precision("bf16"):
        y = ops.conv_s2(hx, hw)
    gx, gw = torch.autograd.grad(y, [hx, hw], gy.to(DEV))
    xr, wr = rb(x).requires_grad_(), rb(w).requires_grad_()
    assert rel_l2(y, F.conv2d(rb(x), rb(w), stride=2, padding=1)) < 2e-6
    (wx,) = torch.autograd.grad(F.conv2d(xr, rb(w), stride=2, padding=1), xr, rb(gy))
    (ww,) = torch.autograd.grad(F.conv2d(rb(x), wr, stride=2, padding=1), wr, rb(gy))
    assert rel_l2(gx, wx) < 5e-6 and rel_l2(gw, ww) < 5e-6, (rel_l2(gx, wx), rel_l2(gw, ww))
    y32 = ops.conv_s2(hx, hw)
    assert 1e-5 < rel_l2(y, y32) < 1e-2
