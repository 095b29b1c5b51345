"""The strided trunk convolutions (dc_convs2_*: 7x7 / 2 stem, 3x3 / 2) against torch's fp64 convolution: forward, data
gradient (3x3) and weight gradient; determinism; the encoder no longer calls a library convolution at the bench shapes."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

CASES = [
    # B, Ci, Co, H, W, k
    (2, 3, 64, 64, 96, 7),         # depth stem (small)
    (2, 6, 64, 64, 96, 7),         # pose stem: 6 input channels (K = 294 -> padded rows of 320)
    (12, 3, 64, 192, 640, 7),      # depth stem at BASELINE configs[1]
    (3, 6, 64, 66, 200, 7),        # patch-staged stem: odd number of output rows (33), ragged last column tile (100 = 64 + 36)
    (2, 3, 64, 34, 40, 7),         # a single, mostly empty tile column (Wo = 20), odd Ho
    (2, 6, 64, 320, 1024, 7),      # pose stem at BASELINE configs[2]
    (2, 6, 32, 64, 96, 7),         # 32 output channels: the gather kernels
    (4, 64, 128, 48, 160, 3),      # resnet18 layer2.0.conv1 at 192x640
    (2, 128, 256, 24, 80, 3),      # layer3.0.conv1
    (3, 256, 512, 12, 40, 3),      # layer4.0.conv1: 120 output pixels per image (partial last reduction chunk at B = 3)
    (1, 256, 512, 12, 40, 3),      # BASELINE configs[0] batch
    (2, 128, 128, 80, 256, 3),     # resnet50 layer2.0.conv2 at 320x1024
    (2, 512, 512, 20, 64, 3),      # resnet50 layer4.0.conv2
    (2, 64, 96, 16, 24, 3),        # small map, 96 output channels (three 32-channel reduction chunks in the data gradient)
]


@pytest.mark.parametrize("B,Ci,Co,H,W,k", CASES)
def test_convs2_vs_torch(B, Ci, Co, H, W, k):
    from depthcore import ops
    g = torch.Generator().manual_seed(B * 100 + Ci + k)
    need_dx = k == 3
    x = torch.randn(B, Ci, H, W, generator=g).cuda().requires_grad_(need_dx)
    w = (torch.randn(Co, Ci, k, k, generator=g) * (1.0 / (Ci * k * k)) ** 0.5).cuda().requires_grad_(True)
    assert ops.conv_s2_supported(x, w)
    y = ops.conv_s2(x, w)
    gy = torch.randn(y.shape, generator=g).cuda()
    y.backward(gy)
    xr, wr = x.detach().double().requires_grad_(need_dx), w.detach().double().requires_grad_(True)
    yr = F.conv2d(xr, wr, None, 2, k // 2)
    yr.backward(gy.double())
    checks = [("y", y, yr), ("dw", w.grad, wr.grad)] + ([("dx", x.grad, xr.grad)] if need_dx else [])
    for name, got, ref in checks:
        err = (got.double() - ref).abs().max().item()
        assert err <= 1e-5 * max(ref.abs().max().item(), 1e-6), "%s: %.3e (scale %.3e)" % (name, err, ref.abs().max().item())


def test_convs2_deterministic_and_stem_has_no_data_gradient():
    from depthcore import ops, _lib
    x = torch.randn(4, 64, 48, 160).cuda().requires_grad_(True)
    w = torch.randn(128, 64, 3, 3).cuda().requires_grad_(True)
    gy = torch.randn(4, 128, 24, 80).cuda()
    outs = []
    for _ in range(2):
        x.grad = w.grad = None
        y = ops.conv_s2(x, w)
        y.backward(gy)
        outs.append((y.detach().clone(), x.grad.clone(), w.grad.clone()))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    xs = torch.randn(1, 3, 32, 64).cuda().requires_grad_(True)
    ws = torch.randn(64, 3, 7, 7).cuda().requires_grad_(True)
    with pytest.raises(_lib.DepthcoreError):
        ops.conv_s2(xs, ws).sum().backward()


def test_encoder_runs_without_library_convolutions(monkeypatch):
    """At the bench shapes every convolution of the trunk is a depthcore launch: torch's convolution is never entered."""
    import networks
    calls = []
    orig = F.conv2d
    monkeypatch.setattr(torch.nn.functional, "conv2d", lambda *a, **k: (calls.append(1), orig(*a, **k))[1])
    real = torch.nn.Conv2d.forward
    monkeypatch.setattr(torch.nn.Conv2d, "forward", lambda self, x: (calls.append(self), real(self, x))[1])
    for layers, shape in ((18, (2, 3, 192, 640)), (50, (1, 3, 320, 1024))):
        torch.manual_seed(0)
        enc = networks.ResnetEncoder(layers, False).cuda()
        enc.train()
        x = torch.rand(shape, device="cuda")
        feats = enc(x)
        sum(f.mean() for f in feats).backward()
        assert calls == [], "library convolution entered for %r" % calls[:3]


@pytest.mark.parametrize("nf,Bf,H,W", [(1, 3, 64, 128), (3, 2, 64, 128), (3, 2, 96, 192), (1, 1, 192, 640)])
def test_stem_on_raw_frames_equals_stem_on_materialised_input(nf, Bf, H, W):
    """dc_stem_fwd / dc_stem_wgrad: `(x - 0.45) / 0.225` (networks/resnet_encoder.py:89) and the pose pairs' concat
    (trainer.py:398-412) inside the patch loader -- bit for bit the result of dc_convs2_* on the tensors it replaces."""
    from depthcore import ops
    g = torch.Generator().manual_seed(nf * 10 + Bf)
    frames = [torch.rand(Bf, 3, H, W, generator=g).cuda() for _ in range(nf)]
    w = (torch.randn(64, 3 * (2 if nf == 3 else 1), 7, 7, generator=g) * 0.05).cuda().requires_grad_()
    assert ops.stem_supported(frames, w)
    y = ops.stem_conv(frames, w)
    if nf == 3:
        x = torch.cat([torch.cat([frames[0], frames[1]], 1), torch.cat([frames[1], frames[2]], 1)], 0)
    else:
        x = frames[0]
    # the reference's arithmetic is the CPU's: a subtraction and a TRUE division (on the GPU, ATen turns `/ scalar` into a
    # multiplication by the reciprocal -- one bit less faithful); the kernels' loader does what the CPU does
    xn = ((x.cpu() - 0.45) / 0.225).cuda()
    y_ref = ops.conv_s2(xn, w)
    assert y.shape == y_ref.shape and torch.equal(y, y_ref)
    gy = torch.randn(y.shape, generator=g).cuda()
    (gw,) = torch.autograd.grad(y, w, gy)
    (gw_ref,) = torch.autograd.grad(y_ref, w, gy)
    assert torch.equal(gw, gw_ref)


def test_pose_encoder_forward_pairs_equals_stacked_forward():
    import networks
    torch.manual_seed(0)
    enc = networks.ResnetEncoder(18, False, num_input_images=2).cuda()
    enc.train()
    g = torch.Generator().manual_seed(3)
    f = [torch.rand(2, 3, 64, 128, generator=g).cuda() for _ in range(3)]
    sd = {k: v.clone() for k, v in enc.state_dict().items()}
    a = enc.forward_pairs(*f)
    ga = torch.autograd.grad(sum(t.square().sum() for t in a), enc.encoder.conv1.weight)[0]
    enc.load_state_dict(sd)                      # (training-mode BatchNorm advanced the running statistics)
    x = torch.cat([torch.cat([f[0], f[1]], 1), torch.cat([f[1], f[2]], 1)], 0)
    b = enc(x, bn_groups=2)
    gb = torch.autograd.grad(sum(t.square().sum() for t in b), enc.encoder.conv1.weight)[0]
    # (the stacked path normalises with ATen's GPU `/ scalar` = multiply by the reciprocal, the fused loader with the CPU
    # reference's true division: equal to the last bit or two of the input, not bitwise)
    def rel(u, v):
        return float((u - v).norm() / v.norm())
    for u, v in zip(a, b):
        assert rel(u, v) < 1e-4, rel(u, v)
    assert rel(ga, gb) < 5e-3, rel(ga, gb)          # (a ReLU decision may flip on a last-bit input difference)


@pytest.mark.parametrize("B,Ci,Co,H,W,k,s,p,bias", [
    (2, 3, 8, 17, 23, 7, 2, 3, False),       # a stem on an odd map, input gradient requested
    (2, 16, 24, 9, 13, 3, 1, 1, False),      # stride-1 3x3 on an odd width
    (3, 16, 32, 9, 13, 3, 2, 1, False),      # 3x3 / 2 -> 5 x 7
    (2, 32, 64, 9, 13, 1, 2, 0, False),      # `downsample` 1x1 / 2 on an odd map
    (1, 5, 7, 6, 10, 3, 1, 1, True),         # channel counts outside every tile, with a bias
    (2, 4, 4, 3, 3, 3, 1, 0, True),          # one output pixel (no padding)
    (1, 2, 3, 12, 8, 5, 3, 2, False),        # 5x5 / 3
])
def test_conv2d_direct_vs_torch(B, Ci, Co, H, W, k, s, p, bias):
    """dc_conv2d_direct_*: forward, data gradient, weight and bias gradients against torch's fp64 convolution; bitwise
    reproducible."""
    from depthcore import ops
    g = torch.Generator().manual_seed(B * 100 + Ci + k)
    x = torch.randn(B, Ci, H, W, generator=g).cuda().requires_grad_(True)
    w = (torch.randn(Co, Ci, k, k, generator=g) * (1.0 / (Ci * k * k)) ** 0.5).cuda().requires_grad_(True)
    b = torch.randn(Co, generator=g).cuda().requires_grad_(True) if bias else None
    outs = []
    for _ in range(2):
        x.grad = w.grad = None
        if b is not None:
            b.grad = None
        y = ops.conv2d_direct(x, w, b, s, p)
        gy = torch.randn(y.shape, generator=torch.Generator().manual_seed(7)).cuda()
        y.backward(gy)
        outs.append([y.detach().clone(), x.grad.clone(), w.grad.clone()] + ([b.grad.clone()] if bias else []))
    for u, v in zip(*outs):
        assert torch.equal(u, v)
    xr, wr = x.detach().double().requires_grad_(True), w.detach().double().requires_grad_(True)
    br = b.detach().double().requires_grad_(True) if bias else None
    yr = F.conv2d(xr, wr, br, s, p)
    assert yr.shape == y.shape
    yr.backward(gy.double())
    checks = [("y", y, yr), ("dx", x.grad, xr.grad), ("dw", w.grad, wr.grad)] + ([("db", b.grad, br.grad)] if bias else [])
    for name, got, ref in checks:
        err = (got.double() - ref).abs().max().item()
        assert err <= 1e-5 * max(ref.abs().max().item(), 1e-6), "%s: %.3e (scale %.3e)" % (name, err, ref.abs().max().item())


def test_conv2d_direct_refuses_what_it_cannot_do():
    from depthcore import ops, _lib
    x = torch.randn(1, 2, 4, 4).cuda()
    with pytest.raises(_lib.DepthcoreError):
        ops.conv2d_direct(x, torch.randn(3, 2, 13, 13).cuda(), None, 1, 6)      # kernel beyond 11
    with pytest.raises(_lib.DepthcoreError):
        ops.conv2d_direct(x, torch.randn(3, 2, 3, 3).cuda(), None, 1, 3)        # padding >= kernel
    with pytest.raises(_lib.DepthcoreError):
        ops.conv2d_direct(x, torch.randn(3, 4, 3, 3).cuda(), None, 1, 1)        # channel mismatch


@pytest.mark.parametrize("B,Ci,H,W", [(2, 3, 64, 96), (3, 6, 66, 200), (12, 3, 192, 640), (4, 6, 192, 640)])
def test_stem_forward_on_split_operands_is_fp32_accurate(B, Ci, H, W):
    """The stem forward through three bf16 pieces per operand on the bf16 matrix cores (stem_fwd_x3_kernel, dc_set_gemm_split) against
    the fp32-MFMA kernel: norm-wise error of both against torch's fp64 convolution -- the split path at most 1.25x the fp32 path's
    (+ 2e-8) -- and the pointwise bound of test_convs2_vs_torch."""
    from depthcore import ops, _lib
    L = _lib.lib()
    g = torch.Generator().manual_seed(B * 10 + Ci)
    x = ((torch.rand(B, Ci, H, W, generator=g) - 0.45) / 0.225).cuda()             # what the network feeds the stem
    w = (torch.randn(64, Ci, 7, 7, generator=g) * (2.0 / (Ci * 49)) ** 0.5).cuda()
    ref = F.conv2d(x.double(), w.double(), None, 2, 3)
    out = {}
    prev = L.dc_get_gemm_split()
    try:
        for mode in (0, 1):
            L.dc_set_gemm_split(mode)
            out[mode] = ops.conv_s2(x, w).detach()
    finally:
        L.dc_set_gemm_split(prev)
    e32 = float((out[0].double() - ref).norm() / ref.norm())
    e3 = float((out[1].double() - ref).norm() / ref.norm())
    print("stem forward %s: |x3 - f64| / |f64| = %.2e, |f32-MFMA - f64| / |f64| = %.2e" % ((B, Ci, H, W), e3, e32))
    assert e3 <= 1.25 * e32 + 2e-8, (e3, e32)
    err = (out[1].double() - ref).abs().max().item()
    assert err <= 1e-5 * ref.abs().max().item()
    assert not torch.equal(out[0], out[1])               # (the two kernels really are different arithmetic orders)


@pytest.mark.parametrize("B,Ci,Co,H,W", [(4, 64, 128, 48, 160), (2, 128, 256, 24, 80), (2, 128, 128, 80, 256), (2, 512, 512, 20, 64),
                                         (2, 64, 96, 16, 24), (12, 256, 512, 12, 40)])
def test_conv3x3s2_on_split_operands_is_fp32_accurate(B, Ci, Co, H, W):
    """The 3x3 / 2 forward and weight gradient through the split-operand kernels (g1x3_kernel / g1x3_wgrad_kernel with the S = 3 gather
    loaders, dc_set_gemm_split(3): an opt-in, slower than the default on most shapes) against the fp32-MFMA kernels (cg_fwd3 / cg_wgrad3):
    norm-wise error of both against torch's fp64 convolution, the split path at most the fp32 path's (measured: 0.3 - 0.8x); the data
    gradient is the same kernel in both modes."""
    from depthcore import ops, _lib
    L = _lib.lib()
    g = torch.Generator().manual_seed(B * 7 + Ci)
    x0 = torch.relu(torch.randn(B, Ci, H, W, generator=g)).cuda()                      # post-ReLU activations, as in the trunks
    w0 = (torch.randn(Co, Ci, 3, 3, generator=g) * (2.0 / (Ci * 9)) ** 0.5).cuda()
    gy = torch.randn(B, Co, H // 2, W // 2, generator=g).cuda()
    xr, wr = x0.double().requires_grad_(True), w0.double().requires_grad_(True)
    yr = F.conv2d(xr, wr, None, 2, 1)
    yr.backward(gy.double())
    ref = {"y": yr.detach(), "dw": wr.grad, "dx": xr.grad}
    out = {}
    prev = L.dc_get_gemm_split()
    try:
        for mode in (0, 3):
            L.dc_set_gemm_split(mode)
            x, w = x0.clone().requires_grad_(True), w0.clone().requires_grad_(True)
            y = ops.conv_s2(x, w)
            y.backward(gy)
            out[mode] = {"y": y.detach(), "dw": w.grad, "dx": x.grad}
    finally:
        L.dc_set_gemm_split(prev)
    eligible = (H // 2) * (W // 2) % 16 == 0
    for name in ("y", "dw"):
        e32 = float((out[0][name].double() - ref[name]).norm() / ref[name].norm())
        e3 = float((out[3][name].double() - ref[name]).norm() / ref[name].norm())
        print("3x3/2 %s %s: |x3 - f64| / |f64| = %.2e, |f32-MFMA - f64| / |f64| = %.2e" % (name, (B, Ci, Co, H, W), e3, e32))
        assert e3 <= e32 + 1e-9, (name, e3, e32)
        err = (out[3][name].double() - ref[name]).abs().max().item()
        assert err <= 1e-5 * ref[name].abs().max().item(), (name, err)
        assert torch.equal(out[0][name], out[3][name]) == (not eligible)    # (120 output pixels per image at 12 x 40: not a multiple of 16, the shape stays on the fp32 kernels)
    assert torch.equal(out[0]["dx"], out[3]["dx"])
