"""Winograd F(2x2,3x3) trunk convolution (dc_wino3x3_*) against the plain fp32 torch convolution.

Floating-point kernel -> the checker is torch's direct conv2d in fp64 on the same inputs; tolerance 2e-5 of the
output scale (Winograd rounding; the north-star tolerance for the path is 1e-3)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

CASES = [
    # B, Ci, Co, H, W
    (2, 64, 64, 48, 160),      # layer1 (2x8 regions, MR 2 or 1)
    (12, 64, 64, 48, 160),     # layer1 at the bench batch (MR 2)
    (3, 128, 128, 24, 80),     # layer2
    (2, 256, 256, 12, 40),     # layer3 (3x5 regions)
    (2, 512, 512, 6, 20),      # layer4 (3x5 regions)
    (1, 3, 5, 7, 10),          # ragged: odd H, channel counts below one block
    (2, 20, 40, 9, 34),        # K not a multiple of the chunk, M straddles blocks
    (1, 8, 16, 2, 2),          # single tile
    (2, 64, 256, 80, 256),     # bottleneck-style, C3 geometry
    (1, 512, 512, 6, 20),      # layer4 at batch 1: one sub-region per image, the reduction split 8 ways (small outputs)
    (1, 256, 256, 12, 40),     # layer3 at batch 1 (split 4 or 8)
    (3, 512, 512, 6, 20),      # the ConvGRU sequence batch
    (1, 72, 40, 6, 20),        # 9 chunks: splits whose last slab gets fewer chunks than the others
]


@pytest.mark.parametrize("B,Ci,Co,H,W", CASES)
def test_wino_forward_and_grads(B, Ci, Co, H, W):
    from depthcore import ops
    g = torch.Generator().manual_seed(B * 1000 + Ci + H)
    x = torch.randn(B, Ci, H, W, generator=g).cuda().requires_grad_(True)
    w = (torch.randn(Co, Ci, 3, 3, generator=g) * (2.0 / (9 * Ci)) ** 0.5).cuda().requires_grad_(True)
    gy = torch.randn(B, Co, H, W, generator=g).cuda()
    y = ops.wino_conv3x3(x, w)
    y.backward(gy)
    xr = x.detach().double().requires_grad_(True)
    wr = w.detach().double().requires_grad_(True)
    yr = F.conv2d(xr, wr, None, 1, 1)
    yr.backward(gy.double())
    for name, got, ref in (("y", y, yr), ("dx", x.grad, xr.grad), ("dw", w.grad, wr.grad)):
        err = (got.double() - ref).abs().max().item()
        scale = ref.abs().max().item()
        assert err <= 2e-5 * scale, "%s: max err %.3e vs scale %.3e" % (name, err, scale)


def test_wino_rejects_odd_width():
    from depthcore import ops
    x = torch.randn(1, 4, 6, 7).cuda()
    w = torch.randn(4, 4, 3, 3).cuda()
    with pytest.raises(RuntimeError):
        ops.wino_conv3x3(x, w)


def test_wino_deterministic():
    """forward, data gradient and the split-reduced weight gradient are bit-reproducible (no atomics)"""
    from depthcore import ops
    x = torch.randn(2, 64, 24, 80).cuda().requires_grad_(True)
    w = torch.randn(64, 64, 3, 3).cuda().requires_grad_(True)
    gy = torch.randn(2, 64, 24, 80).cuda()
    outs = []
    for _ in range(2):
        x.grad = w.grad = None
        y = ops.wino_conv3x3(x, w)
        y.backward(gy)
        outs.append((y.detach().clone(), x.grad.clone(), w.grad.clone()))
    for a, b in zip(*outs):
        assert torch.equal(a, b)


@pytest.mark.parametrize("B,C,H,W", [(12, 64, 48, 160), (24, 128, 24, 80), (24, 512, 6, 20)])
def test_wino_full_size_adjoint_and_linearity(B, C, H, W):
    """BASELINE-size trunk shapes, checked through size-independent properties instead of a CPU convolution:
    <conv(x), g> = <x, dgrad(g)> = <w, wgrad(x, g)> (the three kernels are mutually adjoint) and linearity in x."""
    from depthcore import ops
    g0 = torch.Generator().manual_seed(7)
    x = torch.randn(B, C, H, W, generator=g0).cuda().requires_grad_(True)
    x2 = torch.randn(B, C, H, W, generator=g0).cuda()
    w = (torch.randn(C, C, 3, 3, generator=g0) * (2.0 / (9 * C)) ** 0.5).cuda().requires_grad_(True)
    g = torch.randn(B, C, H, W, generator=g0).cuda()
    y = ops.wino_conv3x3(x, w)
    y.backward(g)
    lhs = (y.detach().double() * g.double()).sum().item()
    via_dx = (x.detach().double() * x.grad.double()).sum().item()
    via_dw = (w.detach().double() * w.grad.double()).sum().item()
    scale = (y.detach().double().abs() * g.double().abs()).sum().item()
    assert abs(lhs - via_dx) <= 1e-6 * scale and abs(lhs - via_dw) <= 1e-6 * scale, (lhs, via_dx, via_dw, scale)
    with torch.no_grad():
        y12 = ops.wino_conv3x3(x.detach() + 0.5 * x2, w.detach())
        y2 = ops.wino_conv3x3(x2, w.detach())
        err = (y12 - (y.detach() + 0.5 * y2)).abs().max().item()
    assert err <= 2e-5 * y.detach().abs().max().item()


def test_two_gib_tensor_takes_the_direct_kernels():
    """A 2 GiB activation is beyond the Winograd kernels' 32-bit buffer offsets: wino_conv3x3 must route it to the direct
    implicit-GEMM kernels (size_t indexing) instead of failing, forward and backward.  Checked on crops against the CPU
    convolution (the operator is local) and, for the weight gradient, by the adjoint identity <gy, conv(x, w)> = <gw, w>."""
    from depthcore import ops
    dev = "cuda:0"
    B, Ci, Co, H, W = 8, 64, 32, 1024, 1024
    assert B * Ci * H * W * 4 >= 0x7fffffff
    g = torch.Generator(device=dev).manual_seed(0)
    x = (torch.rand(B, Ci, H, W, device=dev, generator=g) - 0.5).requires_grad_()
    w = (torch.randn(Co, Ci, 3, 3, device=dev, generator=g) * 0.05).requires_grad_()
    y = ops.wino_conv3x3(x, w)
    gy = torch.rand(y.shape, device=dev, generator=g) - 0.5
    gx, gw = torch.autograd.grad(y, [x, w], gy)
    for b, r0 in ((0, 0), (B - 1, H - 40), (3, 500)):
        xc = x.detach()[b:b + 1, :, r0:r0 + 40].cpu().requires_grad_()
        wc = w.detach().cpu()
        yc = F.conv2d(xc, wc, None, 1, 1)
        lo, hi = (0 if r0 == 0 else 1), (40 if r0 + 40 == H else 39)          # rows whose 3x3 support lies inside the crop
        assert torch.allclose(y[b, :, r0 + lo:r0 + hi].cpu(), yc[0, :, lo:hi].detach(), rtol=1e-4, atol=1e-5)
        gyc = gy[b:b + 1, :, r0:r0 + 40].cpu()
        (gxc,) = torch.autograd.grad(yc, [xc], gyc)
        assert torch.allclose(gx[b, :, r0 + lo:r0 + hi].cpu(), gxc[0, :, lo:hi], rtol=1e-4, atol=1e-5)
    lhs = float((gy.double() * y.detach().double()).sum())
    rhs = float((gw.double() * w.detach().double()).sum())
    assert abs(lhs - rhs) <= 1e-4 * max(abs(lhs), 1.0), (lhs, rhs)


F4_CASES = [
    # B, Ci, Co, H, W          (W % 4 == 0; the dispatch also asks for >= 85 % tile-group coverage)
    (2, 64, 64, 48, 160),      # layer1: 12 x 40 tiles, 3 x 5 groups
    (12, 64, 64, 48, 160),     # layer1 at the bench batch
    (3, 128, 128, 24, 80),     # layer2: 6 x 20 tiles
    (2, 256, 256, 12, 40),     # layer3: 3 x 10 tiles
    (2, 64, 128, 80, 256),     # C3 geometry, Ci != Co
    (1, 20, 40, 12, 20),       # K = 20 (not a multiple of 16), M = 40 straddles 16-channel blocks
    (2, 6, 18, 8, 16),         # K % 4 != 0: the last step's missing channels read as zero
    (1, 32, 32, 10, 20),       # H % 4 != 0: the last tile row is half outside the map
]


@pytest.mark.parametrize("B,Ci,Co,H,W", F4_CASES)
def test_wino_f4_forward_and_data_gradient(B, Ci, Co, H, W):
    """dc_set_wino_f4(1): Winograd F(4x4,3x3) (csrc/wino4.hip) for the plain trunk convolutions -- forward, data gradient and
    the data gradient with a residual-fork addend -- against an fp64 direct convolution.  MEASURED, not assumed:
    the bound is the same 2e-5 of the output scale the F(2x2,3x3) kernel is held to, and the test prints both kernels' errors
    (worst element and relative L2) side by side."""
    from depthcore import _lib
    from depthcore.ops import ptr
    L = _lib.lib()
    g = torch.Generator().manual_seed(B * 1000 + Ci + H)
    x = torch.randn(B, Ci, H, W, generator=g).cuda()
    w = (torch.randn(Co, Ci, 3, 3, generator=g) * (2.0 / (9 * Ci)) ** 0.5).cuda()
    gy = torch.randn(B, Co, H, W, generator=g).cuda()
    add = torch.randn(B, Ci, H, W, generator=g).cuda()
    yr = F.conv2d(x.double(), w.double(), None, 1, 1)
    dxr = torch.nn.grad.conv2d_input(x.shape, w.double(), gy.double(), 1, 1)
    ws = torch.empty(L.dc_wino3x3_workspace(B, Ci, Co, H, W), dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    res = {}
    for mode in (0, 1):
        prev = L.dc_set_wino_f4(mode)
        try:
            y, dx, dxa = torch.empty(B, Co, H, W, device="cuda"), torch.empty_like(x), torch.empty_like(x)
            assert L.dc_wino3x3_fwd(ptr(x), ptr(w), ptr(y), ws.data_ptr(), B, Ci, Co, H, W, st) == 0
            assert L.dc_wino3x3_dgrad(ptr(gy), ptr(w), ptr(dx), ws.data_ptr(), B, Ci, Co, H, W, st) == 0
            assert L.dc_wino3x3_dgrad_add(ptr(gy), ptr(w), ptr(dxa), ptr(add), ws.data_ptr(), B, Ci, Co, H, W, st) == 0
            torch.cuda.synchronize()
        finally:
            L.dc_set_wino_f4(prev)
        res[mode] = [(float((got.double() - ref).abs().max() / ref.abs().max()), float((got.double() - ref).norm() / ref.norm()))
                     for got, ref in ((y, yr), (dx, dxr), (dxa, dxr + add.double()))]
    print("B=%d %d->%d %dx%d  F(2x2) [max, L2] y / dx / dx+add: %s | F(4x4): %s" % (
        B, Ci, Co, H, W, ["%.1e %.1e" % e for e in res[0]], ["%.1e %.1e" % e for e in res[1]]))
    for e_max, e_l2 in res[1]:
        assert e_max <= 2e-5 and e_l2 <= 5e-6, res[1]


def test_wino_f4_is_deterministic_and_off_by_default():
    from depthcore import _lib
    from depthcore.ops import ptr
    L = _lib.lib()
    assert L.dc_set_wino_f4(0) == 0                     # default: F(2x2,3x3) everywhere
    B, C, H, W = 4, 64, 48, 160
    x, w = torch.randn(B, C, H, W).cuda(), (torch.randn(C, C, 3, 3) * 0.05).cuda()
    ws = torch.empty(L.dc_wino3x3_workspace(B, C, C, H, W), dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    outs = []
    L.dc_set_wino_f4(1)
    try:
        for _ in range(2):
            y = torch.empty_like(x)
            assert L.dc_wino3x3_fwd(ptr(x), ptr(w), ptr(y), ws.data_ptr(), B, C, C, H, W, st) == 0
            outs.append(y)
        torch.cuda.synchronize()
    finally:
        L.dc_set_wino_f4(0)
    assert torch.equal(outs[0], outs[1])
    assert L.dc_set_wino_f4(2) < 0 and L.dc_set_wino_f4(0) == 0       # bad mode refused, state unchanged


def test_wino_f4_full_step_deltas():
    """"the loss / gradient deltas of a full step".  The same training step (resnet18, 192 x 640, B = 2) with the
    trunk's forward and data-gradient convolutions on F(4x4,3x3) where it applies (layer1-3 maps) and on F(2x2,3x3): losses and
    every parameter gradient, relative to the F(2x2) step.  Measured (MI355X): loss delta 1.3e-7; parameter-gradient deltas median
    5.6e-3, worst 1.2e-2 (pose-encoder BatchNorm biases) -- NOT the kernels' 1e-6: un-forced gradients of a piecewise-smooth
    network move at the 0.5 % level between ANY two fp32 evaluations (a ReLU / max-pool / min() decision taken the other way on a
    rounding-level tie re-routes a gradient: DESIGN 2 "ReLU kinks"; the fp32 CPU oracle is as far from the HIP step), and five
    times the rounding error flips a few more of them.  The loss is the contract's quantity (1e-3); the bounds below are loose
    recorders, not a claim of gradient parity at that level."""
    import sys
    import trainer as T
    from depthcore import _lib
    from depthcore.synthetic import synthetic_batch
    L = _lib.lib()
    batch = synthetic_batch(2, 192, 640, torch.device("cuda:0"), seed=3)
    out = {}
    for mode in (0, 1):
        prev = L.dc_set_wino_f4(mode)
        try:
            tr = T.Trainer(T.default_options(height=192, width=640, batch_size=2, overlap_streams=False), device="cuda:0", seed=6)
            tr.set_train()
            _, losses = tr.process_batch(dict(batch))
            tr.buckets.zero()
            losses["loss"].backward()
            torch.cuda.synchronize()
            out[mode] = (float(losses["loss"].detach()),
                         {k + "." + n: p.grad.detach().clone() for k, m in tr.models.items() for n, p in m.named_parameters() if p.grad is not None})
            tr.close()
        finally:
            L.dc_set_wino_f4(prev)
    dl = abs(out[1][0] - out[0][0]) / abs(out[0][0])
    rel = {n: float((out[1][1][n] - g).norm() / (g.norm() + 1e-30)) for n, g in out[0][1].items()}
    worst = sorted(rel.items(), key=lambda kv: -kv[1])[:3]
    print("F(4x4) vs F(2x2) full step: loss delta %.2e; gradient deltas median %.2e, worst %s" % (
        dl, sorted(rel.values())[len(rel) // 2], ["%s %.1e" % kv for kv in worst]), file=sys.stderr)
    assert dl <= 1e-5
    assert sorted(rel.values())[len(rel) // 2] <= 3e-2 and max(rel.values()) <= 0.15


@pytest.mark.parametrize("B,Ci,Co,H,W", [
    (12, 64, 64, 48, 160),        # layer1 at C2, depth batch: 1.4 rounds of the 32 x 64 tile
    (24, 64, 64, 48, 160),        # pose batch: 2.8 rounds
    (24, 128, 128, 24, 80),       # layer2
    (8, 64, 64, 80, 256),         # resnet50 layer1.conv2 at C3
    (24, 64, 128, 24, 80),        # more output than reduction channels (16 chunks... 8 here), two channel blocks per tile
    (5, 32, 64, 50, 74),          # ragged: partial sub-regions, an odd number of tile blocks, 4 chunks
])
def test_wino_persistent_launch_is_bitwise_the_classic_one(B, Ci, Co, H, W):
    """dc_set_wino_persist(1): launches that run in more than one round of resident blocks become ONE round of persistent
    blocks that pipeline their work items' chunks (the next item's first loads fly under the current item's epilogue).  The
    arithmetic per output is untouched, so forward, data gradient and data gradient + fork addend must be BITWISE the
    classic launch's -- on shapes with 1.4 to 3 rounds, ragged maps, and a shape (the last) where some blocks get one item
    and others two."""
    from depthcore import _lib
    from depthcore._lib import ptr, stream, check
    L = _lib.lib()
    g = torch.Generator().manual_seed(B + Ci + H)
    x = torch.randn(B, Ci, H, W, generator=g).cuda()
    w = (torch.randn(Co, Ci, 3, 3, generator=g) / (3.0 * Ci ** 0.5)).cuda()
    gy = torch.randn(B, Co, H, W, generator=g).cuda()
    add = torch.randn(B, Ci, H, W, generator=g).cuda()
    ws = torch.empty(L.dc_wino3x3_workspace(B, Ci, Co, H, W), dtype=torch.uint8, device="cuda")
    outs = []
    prev = L.dc_set_wino_persist(0)
    try:
        for mode in (0, 1):
            assert L.dc_set_wino_persist(mode) in (0, 1)
            y = torch.empty(B, Co, H, W, device="cuda")
            dx = torch.empty(B, Ci, H, W, device="cuda")
            dxa = torch.empty(B, Ci, H, W, device="cuda")
            check(L.dc_wino3x3_fwd(ptr(x), ptr(w), ptr(y), ws.data_ptr(), B, Ci, Co, H, W, stream(x)), "fwd")
            check(L.dc_wino3x3_dgrad(ptr(gy), ptr(w), ptr(dx), ws.data_ptr(), B, Ci, Co, H, W, stream(x)), "dgrad")
            check(L.dc_wino3x3_dgrad_add(ptr(gy), ptr(w), ptr(dxa), ptr(add), ws.data_ptr(), B, Ci, Co, H, W, stream(x)), "dgrad_add")
            torch.cuda.synchronize()
            outs.append((y, dx, dxa))
    finally:
        L.dc_set_wino_persist(prev)
    for name, u, v in zip(("y", "dx", "dx+addend"), outs[0], outs[1]):
        assert torch.equal(u, v), (name, float((u - v).abs().max()))
    ref = torch.nn.functional.conv2d(x.double(), w.double(), None, 1, 1)
    assert float((outs[1][0].double() - ref).abs().max()) <= 2e-5 * float(ref.abs().max())
