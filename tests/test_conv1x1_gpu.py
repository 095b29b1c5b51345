"""1x1 trunk convolutions (dc_conv1x1_*) against torch's fp64 convolution: forward, data gradient (incl. the zeros a
stride-2 gradient must write) and the split-reduced weight gradient."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

CASES = [
    # B, Ci, Co, H, W, stride
    (12, 64, 128, 48, 160, 2),     # layer2.0.downsample at the bench batch
    (4, 128, 256, 24, 80, 2),
    (4, 256, 512, 12, 40, 2),
    (2, 64, 256, 20, 36, 1),       # bottleneck conv3-style
    (3, 7, 5, 6, 10, 2),           # ragged channel counts
    (1, 20, 70, 5, 9, 1),          # pixels not a multiple of the tile
    (2, 3, 3, 2, 2, 2),
]


@pytest.mark.parametrize("B,Ci,Co,H,W,s", CASES)
def test_conv1x1_vs_torch(B, Ci, Co, H, W, s):
    from depthcore import ops
    g = torch.Generator().manual_seed(B * 100 + Ci + s)
    x = torch.randn(B, Ci, H, W, generator=g).cuda().requires_grad_(True)
    w = (torch.randn(Co, Ci, 1, 1, generator=g) * (1.0 / Ci) ** 0.5).cuda().requires_grad_(True)
    y = ops.conv1x1(x, w, s)
    gy = torch.randn(y.shape, generator=g).cuda()
    # poison the gradient buffer path: dgrad must overwrite every element, including the skipped positions
    y.backward(gy)
    xr, wr = x.detach().double().requires_grad_(True), w.detach().double().requires_grad_(True)
    yr = F.conv2d(xr, wr, None, s)
    yr.backward(gy.double())
    for name, got, ref in (("y", y, yr), ("dx", x.grad, xr.grad), ("dw", w.grad, wr.grad)):
        err = (got.double() - ref).abs().max().item()
        assert err <= 5e-6 * max(ref.abs().max().item(), 1e-6), "%s: %.3e" % (name, err)
    if s == 2:
        assert torch.count_nonzero(x.grad[:, :, 1::2, :]) == 0 and torch.count_nonzero(x.grad[:, :, :, 1::2]) == 0


def test_conv1x1_deterministic_and_rejects_odd_stride2():
    from depthcore import ops
    x = torch.randn(4, 64, 24, 80).cuda().requires_grad_(True)
    w = torch.randn(128, 64, 1, 1).cuda().requires_grad_(True)
    gy = torch.randn(4, 128, 12, 40).cuda()
    outs = []
    for _ in range(2):
        x.grad = w.grad = None
        y = ops.conv1x1(x, w, 2)
        y.backward(gy)
        outs.append((y.detach().clone(), x.grad.clone(), w.grad.clone()))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    with pytest.raises(RuntimeError):
        ops.conv1x1(torch.randn(1, 4, 5, 6).cuda(), torch.randn(4, 4, 1, 1).cuda(), 2)
