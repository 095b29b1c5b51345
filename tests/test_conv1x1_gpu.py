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
    # ResNet-50 Bottleneck / downsample shapes of BASELINE configs[2] (320x1024 -> 80x256 ... 10x32), reduced batch:
    # 128x128, 64x128 and 64x64 tiles of the tiled kernels, flattened batch*pixels, split weight gradient
    (2, 256, 64, 80, 256, 1),      # layer1.x.conv1
    (2, 64, 256, 80, 256, 1),      # layer1.x.conv3 / downsample
    (2, 256, 128, 80, 256, 1),     # layer2.0.conv1 (runs at the input resolution)
    (2, 256, 512, 80, 256, 2),     # layer2.0.downsample
    (2, 128, 512, 40, 128, 1),     # layer2.x.conv3
    (2, 1024, 256, 20, 64, 1),     # layer3.x.conv1
    (2, 512, 1024, 40, 128, 2),    # layer3.0.downsample
    (3, 2048, 512, 10, 32, 1),     # layer4.x.conv1: tiles span images (P = 320)
    (3, 512, 2048, 10, 32, 1),     # layer4.x.conv3
    (3, 1024, 2048, 20, 64, 2),    # layer4.0.downsample
    (24, 512, 256, 6, 20, 1),      # pose decoder squeeze at configs[1] (P = 120: not a multiple of the 32-pixel chunk)
    (5, 260, 12, 6, 20, 1),        # 12 output channels (pose_2), channel count not a multiple of the chunk
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
    for shape in ((4, 64, 128, 24, 80, 2), (8, 1024, 256, 20, 64, 1)):
        B, Ci, Co, H, W, s = shape
        x = torch.randn(B, Ci, H, W).cuda().requires_grad_(True)
        w = torch.randn(Co, Ci, 1, 1).cuda().requires_grad_(True)
        gy = torch.randn(B, Co, H // s, W // s).cuda()
        outs = []
        for _ in range(2):
            x.grad = w.grad = None
            y = ops.conv1x1(x, w, s)
            y.backward(gy)
            outs.append((y.detach().clone(), x.grad.clone(), w.grad.clone()))
        for a, b in zip(*outs):
            assert torch.equal(a, b)
    with pytest.raises(RuntimeError):
        ops.conv1x1(torch.randn(1, 4, 5, 6).cuda(), torch.randn(4, 4, 1, 1).cuda(), 2)


@pytest.mark.parametrize("B,Ci,Co,H,W,act", [(24, 512, 256, 6, 20, 3), (16, 2048, 256, 10, 32, 3), (24, 256, 12, 6, 20, 0),
                                             (2, 512, 256, 2, 3, 3), (2, 256, 12, 2, 3, 0), (3, 64, 32, 8, 8, 1)])
def test_conv1x1_bias_act_vs_torch(B, Ci, Co, H, W, act):
    """The pose decoder's 1x1 convolutions (networks/pose_decoder.py:25,30): bias and ReLU in the epilogue; backward through
    dc_bias_act_bwd.  (2, *, 2, 3) is the reference fixture's feature size (general kernel), the others the tiled ones."""
    from depthcore import ops
    g = torch.Generator().manual_seed(B + Ci + act)
    x = torch.randn(B, Ci, H, W, generator=g).cuda().requires_grad_(True)
    w = (torch.randn(Co, Ci, 1, 1, generator=g) * (1.0 / Ci) ** 0.5).cuda().requires_grad_(True)
    b = torch.randn(Co, generator=g).cuda().requires_grad_(True)
    y = ops.conv1x1(x, w, 1, b, act)
    gy = torch.randn(y.shape, generator=g).cuda()
    y.backward(gy)
    xr, wr, br = (t.detach().double().requires_grad_(True) for t in (x, w, b))
    yr = F.conv2d(xr, wr, br)
    yr = F.relu(yr) if act == 3 else (F.elu(yr) if act == 1 else yr)
    yr.backward(gy.double())
    for name, got, ref in (("y", y, yr), ("dx", x.grad, xr.grad), ("dw", w.grad, wr.grad), ("db", b.grad, br.grad)):
        err = (got.double() - ref).abs().max().item()
        assert err <= 1e-5 * max(ref.abs().max().item(), 1e-6), "%s: %.3e" % (name, err)
