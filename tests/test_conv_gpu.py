"""The fused decoder conv blocks (dc_conv3x3_fwd/bwd: fp32 MFMA implicit GEMM with fused nearest-x2,
concat, reflection pad, bias, ELU/sigmoid) vs the CPU oracle and the reference's golden vectors."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import ref_cpu as R
import make_golden as MG
from helpers import T, close, close_frac, rel_l2
from self_check_shapes import dec_state, pose_state

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def oracle_block(x0, x1, w, b, up, act, pad):
    x = R.upsample_nearest2(x0) if up else x0
    if x1 is not None:
        x = torch.cat([x, x1], 1)
    xp = R._reflect_pad1(x) if pad == 0 else F.pad(x, (1, 1, 1, 1))
    y = F.conv2d(xp, w, b)
    return F.elu(y) if act == 1 else (torch.sigmoid(y) if act == 2 else (F.relu(y) if act == 3 else
                                                                          (torch.tanh(y) if act == 4 else y)))


CASES = [
    # B, C0, C1, Co, H, W, up, act, pad
    (2, 5, 0, 7, 10, 12, False, 1, 0),
    (1, 16, 0, 16, 32, 48, True, 1, 0),
    (2, 32, 64, 32, 24, 40, True, 1, 0),
    (2, 24, 0, 1, 20, 36, False, 2, 0),
    (1, 40, 24, 72, 18, 22, True, 1, 0),
    (2, 12, 0, 20, 17, 19, False, 0, 1),
    (1, 64, 0, 64, 6, 20, False, 1, 0),
    # W % 16 == 0: the vectorised / double-buffered v2 kernels
    (2, 32, 64, 32, 32, 48, True, 1, 0),
    (1, 24, 0, 8, 16, 32, False, 1, 0),
    (2, 16, 16, 20, 32, 64, True, 1, 0),
    (1, 40, 0, 36, 48, 80, False, 0, 1),
    (2, 8, 0, 1, 32, 32, False, 2, 0),
    (1, 16, 32, 16, 20, 16, False, 1, 0),
    # the thin disparity heads (16 / 32 -> 1, reflection pad + sigmoid): plain-FMA forward, data gradient AND weight gradient
    # (dispconv.hip) -- more than one 2,048-pixel block per image, a ragged last block, a map smaller than one block, zero padding
    (2, 16, 0, 1, 48, 96, False, 2, 0),
    (3, 32, 0, 1, 36, 60, False, 2, 0),
    (2, 16, 0, 1, 9, 13, False, 2, 0),
    (1, 32, 0, 1, 64, 64, False, 0, 1),
    # pose decoder: zero padding + ReLU (networks/pose_decoder.py), Winograd in all three passes and the direct path
    (4, 256, 0, 256, 6, 20, False, 3, 1),
    (2, 40, 0, 24, 7, 9, False, 3, 1),
    # Fusion_v3's three tiny convolutions (networks/fusion_v2.py:290,301-302): conv_1 1->2 zero pad, conv3x3 4->1 reflect,
    # UpscalePS 4->4 zero pad + tanh
    (2, 1, 0, 2, 12, 20, False, 0, 1),
    (2, 4, 0, 1, 12, 20, False, 0, 0),
    (2, 4, 0, 4, 12, 20, False, 4, 1),
    (3, 4, 0, 4, 24, 80, False, 4, 1),
    (3, 4, 0, 1, 24, 80, False, 0, 0),
]


@pytest.mark.parametrize("case", CASES)
def test_block_vs_oracle(case):
    from depthcore import ops
    B, C0, C1, Co, H, W, up, act, pad = case
    g = torch.Generator().manual_seed(sum(int(v) for v in case))
    h0, w0 = (H // 2, W // 2) if up else (H, W)
    x0 = torch.randn(B, C0, h0, w0, generator=g)
    x1 = torch.randn(B, C1, H, W, generator=g) if C1 else None
    w = torch.randn(Co, C0 + C1, 3, 3, generator=g) * (1.0 / np.sqrt(9 * (C0 + C1)))
    b = torch.randn(Co, generator=g) * 0.1
    cot = torch.randn(B, Co, H, W, generator=g)
    leaves = [t.clone().requires_grad_() for t in (x0, w, b)] + ([x1.clone().requires_grad_()] if C1 else [])
    yo = oracle_block(leaves[0], leaves[3] if C1 else None, leaves[1], leaves[2], up, act, pad)
    go = torch.autograd.grad((yo * cot).sum(), leaves)
    dl = [t.to(DEV).requires_grad_() for t in (x0, w, b)] + ([x1.to(DEV).requires_grad_()] if C1 else [])
    yh = ops.conv3x3_block(dl[0], dl[3] if C1 else None, dl[1], dl[2], up, act, pad)
    gh = torch.autograd.grad((yh * cot.to(DEV)).sum(), dl)
    close(yh, yo, rtol=1e-4, atol=1e-5)
    for a, c, name in zip(gh, go, ["dx0", "dw", "db", "dx1"]):
        assert rel_l2(a, c) < 1e-4, (name, rel_l2(a, c))
        close(a, c, rtol=1e-3, atol=1e-3 * float(c.abs().max()), msg=name)


SPLIT_CASES = [
    # B, C0, C1, Co, H, W, up, act, pad -- shapes whose data gradient runs on the Winograd kernels with an unsplit reduction
    (2, 32, 64, 32, 24, 40, True, 1, 0), (2, 64, 64, 64, 48, 160, True, 1, 0), (1, 64, 0, 32, 48, 160, False, 1, 0),
    (2, 32, 0, 16, 96, 320, False, 1, 0), (1, 16, 16, 24, 32, 64, True, 1, 0), (2, 64, 64, 128, 24, 80, False, 2, 1),
    (1, 128, 128, 128, 24, 80, True, 1, 0), (2, 40, 24, 72, 18, 22, True, 1, 0), (2, 64, 0, 64, 4, 8, False, 1, 0),
    (1, 32, 32, 32, 4, 4, True, 4, 0), (1, 24, 8, 16, 30, 34, False, 0, 0),
]


@pytest.mark.parametrize("case", SPLIT_CASES)
def test_data_gradient_without_the_padded_scratch(case):
    """dc_set_dgrad_split: the fused block's data gradient written straight to dx0 / dx1 from the interior of the correlation
    (+ conv_ring_kernel for what ReflectionPad folds back) against the full correlation + fold pass -- the same sums in
    another order -- and against the oracle; with addends (in place for dx1, as the ConvGRU level node uses it)."""
    import ctypes
    from depthcore import _lib, ops
    from depthcore._lib import check, ptr, stream
    L = _lib.lib()
    B, C0, C1, Co, H, W, up, act, pad = case
    g = torch.Generator().manual_seed(sum(case[:6]))
    x0 = torch.randn(B, C0, H >> up, W >> up, generator=g)
    x1 = torch.randn(B, C1, H, W, generator=g) if C1 else None
    w = torch.randn(Co, C0 + C1, 3, 3, generator=g) / (3.0 * (C0 + C1) ** 0.5)
    b = 0.1 * torch.randn(Co, generator=g)
    gy = torch.randn(B, Co, H, W, generator=g)
    a0 = torch.randn(x0.shape, generator=g)
    a1 = torch.randn(x1.shape, generator=g) if C1 else None
    res = {}
    for mode in (2, 0):          # 2: the new path wherever it exists (mode 1 keeps it for the shapes where it pays)
        prev = L.dc_set_dgrad_split(mode)
        try:
            hx0, hx1 = x0.to(DEV).requires_grad_(), (x1.to(DEV).requires_grad_() if C1 else None)
            hw, hb = w.to(DEV).requires_grad_(), b.to(DEV).requires_grad_()
            y = ops.conv3x3_block(hx0, hx1, hw, hb, up, act, pad)
            res[mode] = torch.autograd.grad(y, [t for t in (hx0, hx1, hw, hb) if t is not None], gy.to(DEV))
            # the C entry with addends: addend0 a separate tensor, addend1 the output itself
            yy = y.detach()
            gyd, a0d = gy.to(DEV), a0.to(DEV)           # (named: alive until the launches are enqueued)
            d0 = torch.empty_like(hx0)
            d1 = a1.to(DEV).clone() if C1 else None
            ws = torch.empty(L.dc_conv3x3_bwd_workspace(C0, C1, B, Co, H, W), dtype=torch.uint8, device=DEV)
            check(L.dc_conv3x3_bwd_add(ptr(hx0.detach()), C0, int(up), ptr(hx1.detach()) if C1 else None, C1, ptr(hw.detach()), ptr(yy),
                                       ptr(gyd), ptr(d0), ptr(d1) if C1 else None, ptr(a0d), ptr(d1) if C1 else None, None, None,
                                       ws.data_ptr(), B, Co, H, W, act, pad, stream(yy)), "dc_conv3x3_bwd_add")
            res[(mode, "add")] = (d0, d1)
        finally:
            L.dc_set_dgrad_split(prev)
    for a, c in zip(res[2], res[0]):
        assert rel_l2(a, c) < 2e-6, rel_l2(a, c)
    assert rel_l2(res[(2, "add")][0], res[2][0] + a0.to(DEV)) < 1e-6
    if C1:
        assert rel_l2(res[(2, "add")][1], res[2][1] + a1.to(DEV)) < 1e-6
        assert rel_l2(res[(0, "add")][1], res[(2, "add")][1]) < 2e-6
    # and against the oracle
    ox0, ox1 = x0.clone().requires_grad_(), (x1.clone().requires_grad_() if C1 else None)
    oy = oracle_block(ox0, ox1, w, b, up, act, pad)
    og = torch.autograd.grad(oy, [t for t in (ox0, ox1) if t is not None], gy)
    for a, c in zip(res[2], og):
        assert rel_l2(a, c) < 1e-5, rel_l2(a, c)


def test_convblock_golden(golden):
    import layers
    g = golden["layers_ops"]
    cb = layers.ConvBlock(5, 7).to(DEV)
    with torch.no_grad():
        cb.conv.conv.weight.copy_(T(g["cb_w"]))
        cb.conv.conv.bias.copy_(T(g["cb_b"]))
    x = T(g["cb_x"]).to(DEV).requires_grad_()
    y = layers.upsample(cb(x))
    close(y, g["cb_out"], rtol=1e-4, atol=1e-5)
    gx, gw, gb = torch.autograd.grad((y * T(g["cb_cot"]).to(DEV)).sum(), [x, cb.conv.conv.weight, cb.conv.conv.bias])
    close(gx, g["cb_gx"], rtol=1e-3, atol=1e-4)
    close(gw, g["cb_gw"], rtol=1e-3, atol=1e-3)
    close(gb, g["cb_gb"], rtol=1e-3, atol=1e-3)


def test_depth_decoder_golden(golden):
    """networks.DepthDecoder (all levels through dc_conv3x3) vs the reference decoder's golden outputs/grads."""
    import networks
    g = golden["decoders"]
    nce = np.array([64, 64, 128, 256, 512])
    dec = networks.DepthDecoder(nce).to(DEV)
    sd = dec_state(nce, 3)
    assert list(dec.state_dict().keys()) == list(g["dec_keys"])
    dec.load_state_dict(sd)
    feats, gen = MG.decoder_features(nce)
    feats = [f.to(DEV).requires_grad_() for f in feats]
    o = dec(feats)
    tot = 0
    for s in range(4):
        close(o[("disp", s)], g["dec_disp%d" % s], rtol=1e-3, atol=1e-5)
        tot = tot + (o[("disp", s)] * T(g["dec_cot%d" % s]).to(DEV)).sum()
    names = [n for n, _ in dec.named_parameters()]
    grads = torch.autograd.grad(tot, feats + [p for _, p in dec.named_parameters()])
    for i in range(5):
        got = grads[i].cpu() if i >= 3 else torch.from_numpy(MG.summ(grads[i]))
        close(got, g["dec_gfeat%d" % i], rtol=2e-3, atol=2e-4)
    for j, k in enumerate(names):
        gk = grads[5 + j].cpu()
        got = gk if gk.numel() <= 4096 else torch.from_numpy(MG.summ(gk))
        close(got, g["dec_g_" + k], rtol=3e-3, atol=3e-3)
    o2 = dec([f.detach() for f in feats], pre_disp=True)
    close(MG.summ(o2[("disp", 0)]), g["dec_predisp0"], rtol=1e-3, atol=1e-3)


def test_decoder_full_size_determinism():
    import networks
    nce = np.array([64, 64, 128, 256, 512])
    torch.manual_seed(0)
    dec = networks.DepthDecoder(nce).to(DEV)
    g = torch.Generator(device=DEV).manual_seed(1)
    feats = [torch.randn(4, int(c), 96 >> i, 320 >> i, device=DEV, generator=g, requires_grad=True) for i, c in enumerate(nce)]
    res = []
    for _ in range(2):
        o = dec(feats)
        tot = sum(o[("disp", s)].square().sum() for s in range(4))
        gr = torch.autograd.grad(tot, feats + list(dec.parameters()))
        res.append([o[("disp", s)].clone() for s in range(4)] + [x.clone() for x in gr])
    for a, b in zip(*res):
        assert torch.equal(a, b)              # split-K slabs reduced in fixed order: bitwise reproducible
    assert o[("disp", 0)].shape == (4, 1, 192, 640)


@pytest.mark.parametrize("fixture", ["2x3", "4x8"])
def test_pose_decoder_golden(golden, fixture):
    """a4: networks.PoseDecoder on the GPU (squeeze / pose_2 on dc_conv1x1_bias_act_fwd, the 3x3 pairs on dc_conv3x3_fwd)
    against the reference decoder's outputs and gradients.  "2x3" is the round-1 fixture (odd width: general kernels),
    "4x8" the round-2 one (tiled 1x1 GEMMs, Winograd 3x3)."""
    import networks
    import make_golden_r2 as MG2
    nce = np.array([64, 64, 128, 256, 512])
    pose = networks.PoseDecoder(nce, num_input_features=1, num_frames_to_predict_for=2).to(DEV)
    assert list(pose.state_dict().keys()) == list(golden["decoders"]["pose_keys"])
    pose.load_state_dict(pose_state(nce, 5))
    if fixture == "2x3":
        g = golden["decoders"]
        key = lambda k: "pose_" + k                  # noqa: E731
        f4 = T(g["pose_feat"])
    else:
        g = golden["pose_even"]
        key = lambda k: k                            # noqa: E731
        f4, _ = MG2.pose_even_feature()
    f4 = f4.to(DEV).requires_grad_()
    a, t = pose([[f4]])
    close(a, g[key("aa")], rtol=1e-3, atol=1e-6)
    close(t, g[key("tr")], rtol=1e-3, atol=1e-6)
    names = [n for n, _ in pose.named_parameters()]
    gr = torch.autograd.grad((a * T(g[key("cot_aa")]).to(DEV)).sum() + (t * T(g[key("cot_tr")]).to(DEV)).sum(),
                             [f4] + [p for _, p in pose.named_parameters()])
    want = g[key("gfeat")]
    got = gr[0].cpu() if want.ndim == 4 else torch.from_numpy(MG.summ(gr[0]))
    close(got, want, rtol=2e-3, atol=1e-6 if want.ndim == 4 else 1e-5)
    for j, k in enumerate(names):
        gk = gr[1 + j].cpu()
        got = gk if gk.numel() <= 4096 else torch.from_numpy(MG.summ(gk))
        close(got, g[key("g_" + k)], rtol=3e-3, atol=2e-4)
