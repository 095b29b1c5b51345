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


# ---- split-operand GEMMs: fp32 through three bf16 pieces per operand on the bf16 matrix cores (csrc/gemm1x1_x3.hip) ----
X3_CASES = [
    # B, Ci, Co, H, W, stride
    (2, 256, 64, 80, 256, 1),      # layer1.x.conv1: half-empty 128-row tile
    (2, 64, 256, 80, 256, 1),      # layer1.x.conv3: two reduction chunks
    (2, 256, 512, 80, 256, 2),     # layer2.0.downsample: stride-2 gather (forward / weight gradient), cell-block scatter (data gradient)
    (3, 1024, 2048, 20, 64, 2),    # layer4.0.downsample
    (3, 2048, 512, 10, 32, 1),     # layer4.x.conv1: tiles span images (P = 320), N = 960 not a multiple of the 128-pixel tile
    (3, 512, 2048, 10, 32, 1),
    (2, 128, 96, 12, 40, 1),       # 96 output rows (row padding), P = 480
    (1, 96, 160, 8, 16, 1),        # 160 rows: one full + one partial row tile; 96 input channels (dgrad: partial tile), P = 128
]


def _x3_run(B, Ci, Co, H, W, s, split, bias=False, act=0, seed=0):
    from depthcore import ops, _lib
    L = _lib.lib()
    g = torch.Generator().manual_seed(1000 + seed + B * 100 + Ci + s)
    x = torch.relu(torch.randn(B, Ci, H, W, generator=g)).cuda().requires_grad_(True)       # post-ReLU activations: what these layers read
    w = (torch.randn(Co, Ci, 1, 1, generator=g) * (2.0 / Ci) ** 0.5).cuda().requires_grad_(True)
    b = (0.1 * torch.randn(Co, generator=g)).cuda().requires_grad_(True) if bias else None
    prev = L.dc_set_gemm_split(int(split))
    try:
        y = ops.conv1x1(x, w, s, b, act)
        gy = torch.randn(y.shape, generator=g).cuda()
        y.backward(gy)
    finally:
        L.dc_set_gemm_split(prev)
    torch.cuda.synchronize()
    return x, w, b, gy, y.detach(), x.grad.clone(), w.grad.clone(), (b.grad.clone() if bias else None)


@pytest.mark.parametrize("B,Ci,Co,H,W,s", X3_CASES)
def test_split_operand_gemms_are_fp32_accurate(B, Ci, Co, H, W, s):
    """The gate of VERDICT round 5 (item 1b): error against an fp64 GEMM no worse than the fp32-MFMA kernels' own.  Norm-wise error of
    y, dx, dw of both paths against torch's fp64 convolution: the split path's must not exceed the fp32-MFMA path's (measured on MI355X
    with the chunk-blocked, sign-alternating accumulation of x3_chunk: 0.2 - 0.35x for y and dx, 0.4 - 0.8x for dw; the margins are
    printed under -s), and stays within the absolute 5e-6-of-the-maximum bound the fp32 kernels are held to above."""
    from depthcore import _lib
    L = _lib.lib()
    assert L.dc_gemm1x1x3_fwd_ok(B, Ci, Co, H, W, s) and L.dc_gemm1x1x3_wgrad_ok(B, Ci, Co, H, W, s)
    r32 = _x3_run(B, Ci, Co, H, W, s, 0)
    r3 = _x3_run(B, Ci, Co, H, W, s, 1)
    x, w, _, gy = r32[:4]
    xr, wr = x.detach().double().requires_grad_(True), w.detach().double().requires_grad_(True)
    yr = F.conv2d(xr, wr, None, s)
    yr.backward(gy.double())
    report = []
    for name, i, ref in (("y", 4, yr.detach()), ("dx", 5, xr.grad), ("dw", 6, wr.grad)):
        e32 = float((r32[i].double() - ref).norm() / ref.norm())
        e3 = float((r3[i].double() - ref).norm() / ref.norm())
        report.append((name, e3, e32))
        assert e3 <= e32 + 1e-9, report
        err = (r3[i].double() - ref).abs().max().item()
        assert err <= 5e-6 * max(ref.abs().max().item(), 1e-6), "%s: %.3e" % (name, err)
    print("split-operand 1x1 %s: (tensor, |x3 - f64| / |f64|, |f32-MFMA - f64| / |f64|) %s" % ((B, Ci, Co, H, W, s), report))
    if s == 2:
        assert torch.count_nonzero(r3[5][:, :, 1::2, :]) == 0 and torch.count_nonzero(r3[5][:, :, :, 1::2]) == 0


def test_split_operand_gemm_epilogues_and_determinism():
    """bias + activation in the forward's epilogue, the residual fork's addend in the data gradient's, bitwise run-to-run."""
    from depthcore import ops, _lib
    L = _lib.lib()
    B, Ci, Co, H, W = 2, 128, 256, 16, 32
    a = _x3_run(B, Ci, Co, H, W, 1, 1, bias=True, act=3)
    b = _x3_run(B, Ci, Co, H, W, 1, 1, bias=True, act=3)
    for i in (4, 5, 6, 7):
        assert torch.equal(a[i], b[i])
    x, w, bs, gy = a[:4]
    xr, wr, br = (t.detach().double().requires_grad_(True) for t in (x, w, bs))
    yr = torch.relu(F.conv2d(xr, wr, br, 1))
    yr.backward(gy.double())
    for got, ref in ((a[4], yr.detach()), (a[5], xr.grad), (a[6], wr.grad), (a[7], br.grad)):
        assert float((got.double() - ref).norm() / ref.norm()) <= 3e-7
    # addend epilogue: dc_gemm1x1x3_dgrad(addend) == dc_gemm1x1x3_dgrad() + addend, one fp32 addition either way
    gyc = torch.randn(B, Co, H, W, device="cuda")
    add = torch.randn(B, Ci, H, W, device="cuda")
    ww = w.detach().reshape(Co, Ci).contiguous()
    ws = torch.empty(L.dc_gemm1x1x3_workspace(Ci, Co), dtype=torch.uint8, device="cuda")
    d0, d1 = torch.empty_like(add), torch.empty_like(add)
    assert L.dc_gemm1x1x3_dgrad(gyc.data_ptr(), ww.data_ptr(), d0.data_ptr(), ws.data_ptr(), None, None, B, Ci, Co, H, W, 1, None) == 0
    assert L.dc_gemm1x1x3_dgrad(gyc.data_ptr(), ww.data_ptr(), d1.data_ptr(), ws.data_ptr(), add.data_ptr(), None, B, Ci, Co, H, W, 1, None) == 0
    torch.cuda.synchronize()
    assert torch.equal(d1, d0 + add)
    # stride 2: every cell of the 2 x 2 blocks is written once -- value + addends at (2 py, 2 px), the addends alone elsewhere
    Bs, Cis, Cos, Hs, Wsz = 2, 64, 128, 16, 32
    gys = torch.randn(Bs, Cos, Hs // 2, Wsz // 2, device="cuda")
    w2 = torch.randn(Cos, Cis, device="cuda") / 8
    a1, a2 = torch.randn(Bs, Cis, Hs, Wsz, device="cuda"), torch.randn(Bs, Cis, Hs, Wsz, device="cuda")
    ws2 = torch.empty(L.dc_gemm1x1x3_workspace(Cis, Cos), dtype=torch.uint8, device="cuda")
    e0, e1 = torch.full_like(a1, float("nan")), torch.full_like(a1, float("nan"))
    assert L.dc_gemm1x1x3_dgrad(gys.data_ptr(), w2.data_ptr(), e0.data_ptr(), ws2.data_ptr(), None, None, Bs, Cis, Cos, Hs, Wsz, 2, None) == 0
    assert L.dc_gemm1x1x3_dgrad(gys.data_ptr(), w2.data_ptr(), e1.data_ptr(), ws2.data_ptr(), a1.data_ptr(), a2.data_ptr(), Bs, Cis, Cos, Hs, Wsz, 2, None) == 0
    torch.cuda.synchronize()
    assert torch.count_nonzero(e0[:, :, 1::2, :]) == 0 and torch.count_nonzero(e0[:, :, :, 1::2]) == 0 and torch.isfinite(e0).all()
    refs = torch.einsum("mk,bmhw->bkhw", w2.double(), gys.double())
    assert float((e0[:, :, ::2, ::2].double() - refs).norm() / refs.norm()) <= 3e-7
    assert torch.allclose(e1, e0 + a1 + a2, rtol=0, atol=2e-6)
    # shapes outside the split kernels are refused by the _ok queries (the callers then keep the fp32-MFMA kernels)
    assert not L.dc_gemm1x1x3_fwd_ok(24, 512, 256, 6, 20, 1)        # P = 120: pixel runs of 16 would straddle images
    assert L.dc_gemm1x1x3_dgrad_ok(2, 256, 512, 80, 256, 2)         # stride-2 data gradient: the scatter epilogue
    assert not L.dc_gemm1x1x3_fwd_ok(5, 260, 12, 6, 20, 1)
