"""Fused photometric loss (dc_photo_fwd / dc_photo_bwd) vs the reference's golden vectors and the
CPU oracle.  All calls go through the C ABI of libdepthcore.so."""
import numpy as np
import pytest
import torch

from oracle import ref_cpu as R
import make_golden as MG
from helpers import T, close, close_frac, rel_l2, oracle_photo, random_poses, pass_rate_1e3

pytestmark = pytest.mark.gpu
B, H, W = MG.B, MG.H, MG.W


def hip_photo(inputs, disps, Ts, noise, materialize=False, **kw):
    from depthcore import ops
    dev = torch.device("cuda:0")
    ns = len(disps)
    cfg = ops.PhotoConfig(
        inputs[("color", 0, 0)].to(dev), inputs[("color", -1, 0)].to(dev), inputs[("color", 1, 0)].to(dev),
        [inputs[("color", 0, s)].to(dev) for s in range(ns)], inputs[("K", 0)].to(dev), inputs[("inv_K", 0)].to(dev),
        noise=None if noise is None else [n.to(dev) for n in noise], materialize=materialize, **kw)
    d = [x.detach().to(dev).requires_grad_() for x in disps]
    t = [x.detach().to(dev).requires_grad_() for x in Ts]
    losses = ops.photometric_loss(cfg, t[0], t[1], d)
    grads = torch.autograd.grad(losses[ns], d + t)
    torch.cuda.synchronize()
    return losses, cfg.extras, grads[:ns], grads[ns:]


VARIANTS = [("auto", {}), ("noauto", dict(disable_automasking=True)),
            ("avg", dict(avg_reprojection=True)), ("nossim", dict(no_ssim=True))]


@pytest.mark.parametrize("tag,kw", VARIANTS)
def test_golden_trainer_level(golden, tag, kw):
    """Same inputs as tests/golden/make_golden.py fed to the reference's Trainer methods."""
    g = golden["trainer_losses"]
    p = tag + "_"
    inputs = R.synthetic_inputs(B, H, W, seed=0)
    disps = [T(g[p + "disp%d" % s]) for s in range(4)]
    Ts = [T(g[p + "T_-1"]), T(g[p + "T_1"])]
    if kw.get("disable_automasking"):
        noise = None
    elif kw.get("avg_reprojection"):
        g2 = torch.Generator().manual_seed(1234)
        noise = [torch.randn(B, 1, H, W, generator=g2) for _ in range(4)]
    else:
        noise = R.tiebreak_noise(B, H, W)
    losses, ex, gd, gT = hip_photo(inputs, disps, Ts, noise, materialize=(tag == "auto"), **kw)
    # tolerance: north_star "within 1e-3 rel fp32"
    close(losses[4], g[p + "loss"], rtol=1e-3, atol=0)
    for s in range(4):
        close(losses[s], g[p + "loss%d" % s], rtol=1e-3, atol=0)
        # measured margin (profiles/round6_parity_passrates.txt): with NO absolute floor 97.3-99.9 % of the elements meet the plain
        # 1e-3 (scale 2 of "auto" 97.3 %, "nossim" 99.9-100 %); the misses are elements near zero of gradient maps that are sums of
        # mixed-sign SSIM derivatives -- with the 1e-7 floor fewer than 1 % remain
        close_frac(gd[s], g[p + "gdisp%d" % s], rtol=1e-3, atol=1e-7, bad=1e-2, msg="gdisp%d" % s)
        assert rel_l2(gd[s], g[p + "gdisp%d" % s]) < 3e-2
        if not kw.get("disable_automasking"):
            first_reproj = 1 if kw.get("avg_reprojection") else 2
            sel = (ex["argmin"][s] >= first_reproj).cpu().numpy().astype(np.uint8)
            want = np.unpackbits(g[p + "idsel%d" % s])[:sel.size].reshape(sel.shape)
            assert (sel != want).mean() < 2e-3
    # the distance to north_star's PLAIN tolerance, as a number (the gates above are the calibrated ones)
    print("%s: fraction of d(disp) elements within 1e-3 pointwise of the reference's fixture, scales 0-3:" % tag,
          [round(pass_rate_1e3(gd[s], g[p + "gdisp%d" % s]), 4) for s in range(4)])
    # dT against the oracle's autograd (the golden file pins d axisangle / d translation instead)
    opt = R.Opt(height=H, width=W, **kw)
    _, _, _, ogT = oracle_photo(inputs, disps, Ts, opt, noise)
    for f in range(2):
        # pose grads are sums with heavy cancellation over few pixels here: normwise
        assert rel_l2(gT[f][:, :3, :], ogT[f][:, :3, :]) < 3e-2
    if tag == "auto":
        for s in (0, 3):
            close(ex["depth"][s], g[p + "depth%d" % s], rtol=1e-4)
            for j, f in enumerate((-1, 1)):
                close(ex["sample"][s][j], g[p + "sample_%d_%d" % (f, s)], atol=1e-5)
                close_frac(ex["color"][s][j], g[p + "color_%d_%d" % (f, s)], rtol=1e-3, atol=1e-4, bad=1e-3)
                idsel = ex["identity_selection"][s].cpu().numpy()
                assert np.array_equal(idsel, (ex["argmin"][s] >= 2).float().cpu().numpy())


@pytest.mark.parametrize("shape", [(1, 32, 64), (3, 40, 72), (2, 96, 136)])
def test_vs_oracle_ragged_shapes(shape):
    """Widths / heights that are not multiples of the 62/60-column strips or the 16-row blocks."""
    b, h, w = shape
    inputs = R.synthetic_inputs(b, h, w, num_scales=3, seed=3)
    g = torch.Generator().manual_seed(9)
    disps = [torch.rand(b, 1, h >> s, w >> s, generator=g) for s in range(3)]
    Ts = random_poses(b, 5)
    noise = R.tiebreak_noise(b, h, w, num_scales=3)
    opt = R.Opt(height=h, width=w, scales=(0, 1, 2))
    ol, oo, ogd, ogT = oracle_photo(inputs, disps, Ts, opt, noise)
    losses, ex, gd, gT = hip_photo(inputs, disps, Ts, noise)
    close(losses[3], ol["loss"], rtol=1e-3, atol=0)
    for s in range(3):
        close(losses[s], ol["loss/%d" % s], rtol=1e-3, atol=0)
        close_frac(gd[s], ogd[s], rtol=2e-3, atol=0, atol_rel=2e-3, bad=2e-2, msg="gdisp%d" % s)
    for f in range(2):
        assert rel_l2(gT[f][:, :3, :], ogT[f][:, :3, :]) < 3e-2


def test_full_size_c2_vs_oracle():
    """BASELINE config 2: B=12, 192x640, 4 scales -- loss and gradients vs the CPU oracle, and
    the error put next to the fp32-vs-fp64 conditioning of the oracle itself."""
    b, h, w = 12, 192, 640
    inputs = R.synthetic_inputs(b, h, w, seed=0)
    g = torch.Generator().manual_seed(77)
    disps = [torch.rand(b, 1, h >> s, w >> s, generator=g) for s in range(4)]
    Ts = random_poses(b, 6)
    noise = R.tiebreak_noise(b, h, w)
    opt = R.Opt()
    ol, oo, ogd, ogT = oracle_photo(inputs, disps, Ts, opt, noise)
    losses, ex, gd, gT = hip_photo(inputs, disps, Ts, noise)
    close(losses[4], ol["loss"], rtol=1e-3, atol=0)
    for s in range(4):
        close(losses[s], ol["loss/%d" % s], rtol=1e-3, atol=0)
        sel = (ex["argmin"][s] >= 2).cpu()
        assert (sel != oo["identity_selection/%d" % s].bool()).float().mean() < 1e-3
        # each scale-s pixel sums 4^s full-res gradients of mixed sign: allow more outliers there
        # measured (profiles/round6_parity_passrates.txt, "C2 full size"): plain 1e-3 pass-rates 98.2 / 96.9 / 95.3 / 91.7 % at scales 0-3
        # against the fp32 oracle, whose OWN distance from fp64 on these tensors is 0.96-1.3e-2 in the norm ("conditioning": the HIP path
        # 0.75-1.15e-2, i.e. closer to fp64 than the oracle at every scale); the pose gradients -- 12 numbers per frame, each a sum over
        # 1.5 M pixels -- meet the plain 1e-3 in 15-17 % of their entries and 2.4e-3 in the norm, the fp32 oracle's own 2.3e-3
        close_frac(gd[s], ogd[s], rtol=2e-3, atol=0, atol_rel=2e-3, bad=1e-2 * (s + 1), msg="gdisp%d" % s)
        assert rel_l2(gd[s], ogd[s]) < 2e-2
    for f in range(2):
        assert rel_l2(gT[f][:, :3, :], ogT[f][:, :3, :]) < 2e-2
    print("C2 full size: fraction of gradient elements within 1e-3 pointwise of the fp32 oracle -- d(disp) scales 0-3:",
          [round(pass_rate_1e3(gd[s], ogd[s]), 4) for s in range(4)], "d(T) frames -1, +1:",
          [round(pass_rate_1e3(gT[f][:, :3, :], ogT[f][:, :3, :]), 4) for f in range(2)])


@pytest.mark.parametrize("shape", [(1, 192, 640), (2, 160, 512), (1, 320, 1024)])
def test_vs_oracle_c1_and_c3_shapes(shape):
    """BASELINE configs[0] (a single 192x640 triplet) and the 320x1024 frame of configs[2] (one full-size frame and a
    half-size batch of the same aspect), 4 scales, against the CPU oracle."""
    b, h, w = shape
    inputs = R.synthetic_inputs(b, h, w, seed=4)
    g = torch.Generator().manual_seed(10)
    disps = [torch.rand(b, 1, h >> s, w >> s, generator=g) for s in range(4)]
    Ts = random_poses(b, 8)
    noise = R.tiebreak_noise(b, h, w)
    opt = R.Opt(height=h, width=w)
    ol, oo, ogd, ogT = oracle_photo(inputs, disps, Ts, opt, noise)
    losses, ex, gd, gT = hip_photo(inputs, disps, Ts, noise)
    close(losses[4], ol["loss"], rtol=1e-3, atol=0)
    for s in range(4):
        close(losses[s], ol["loss/%d" % s], rtol=1e-3, atol=0)
        sel = (ex["argmin"][s] >= 2).cpu()
        assert (sel != oo["identity_selection/%d" % s].bool()).float().mean() < 1e-3
        assert rel_l2(gd[s], ogd[s]) < 2e-2
    for f in range(2):
        assert rel_l2(gT[f][:, :3, :], ogT[f][:, :3, :]) < 2e-2


def test_gradient_error_is_the_conditioning_of_the_fp32_problem():
    """Why the gradient budgets above are 2e-2 and not 1e-3: the photometric gradient is ill-conditioned in fp32
    (E[x^2]-mu^2 cancellation over C2 = 9e-4, floor/argmin/clamp switching).  At the full configs[1] size the oracle is
    evaluated in fp64 and in fp32; the HIP path (fp32) must be no further from the fp64 gradients than twice the
    oracle's own fp32 evaluation is -- i.e. its error is the problem's, not the kernels'."""
    b, h, w = 12, 192, 640
    inputs = R.synthetic_inputs(b, h, w, seed=0)
    g = torch.Generator().manual_seed(77)
    disps = [torch.rand(b, 1, h >> s, w >> s, generator=g) for s in range(4)]
    Ts = random_poses(b, 6)
    noise = R.tiebreak_noise(b, h, w)
    opt = R.Opt()
    l64, _, gd64, gT64 = oracle_photo(inputs, disps, Ts, opt, noise, dtype=torch.float64)
    l32, _, gd32, gT32 = oracle_photo(inputs, disps, Ts, opt, noise)
    losses, ex, gd, gT = hip_photo(inputs, disps, Ts, noise)
    assert abs(float(losses[4]) - float(l64["loss"])) <= 1e-4 * abs(float(l64["loss"]))
    report = []
    for s in range(4):
        e_hip, e_32 = rel_l2(gd[s], gd64[s]), rel_l2(gd32[s], gd64[s])
        report.append(("gdisp%d" % s, e_hip, e_32))
        assert e_hip <= 2.0 * e_32 + 1e-4, report
    for f in range(2):
        e_hip, e_32 = rel_l2(gT[f][:, :3, :], gT64[f][:, :3, :]), rel_l2(gT32[f][:, :3, :], gT64[f][:, :3, :])
        report.append(("dT%d" % f, e_hip, e_32))
        assert e_hip <= 2.0 * e_32 + 1e-4, report
    print("conditioning (name, |hip-f64|/|f64|, |f32-f64|/|f64|):", report)


@pytest.mark.parametrize("shape", [(12, 192, 640), (8, 320, 1024)])
def test_properties_full_size(shape):
    """Size-independent properties at the bench sizes (configs[1] and configs[2] per rank): determinism (bitwise),
    finite gradients, loss = mean of the scale losses, on-device tie-break noise only breaks ties."""
    from depthcore import ops
    b, h, w = shape
    inputs = R.synthetic_inputs(b, h, w, seed=1)
    g = torch.Generator().manual_seed(2)
    disps = [torch.rand(b, 1, h >> s, w >> s, generator=g) for s in range(4)]
    Ts = random_poses(b, 7)
    noise = R.tiebreak_noise(b, h, w)
    l1, e1, gd1, gT1 = hip_photo(inputs, disps, Ts, noise)
    l2, e2, gd2, gT2 = hip_photo(inputs, disps, Ts, noise)
    assert torch.equal(l1, l2)
    for a, c in zip(list(gd1) + list(gT1), list(gd2) + list(gT2)):
        assert torch.equal(a, c)                      # no atomics: bitwise reproducible
    assert all(torch.isfinite(x).all() for x in list(gd1) + list(gT1))
    close(l1[4], l1[:4].mean(), rtol=1e-6)
    # on-device tie-break RNG: same loss to ~1e-5 (noise only breaks ties)
    l3, _, _, _ = hip_photo(inputs, disps, Ts, None, rng_seed=123)
    close(l3[4], l1[4], rtol=1e-3)


@pytest.mark.parametrize("tag,kw", VARIANTS)
def test_evaluation_forward_equals_training_forward(tag, kw):
    """Round 4: with gradients required the forward also emits d(loss)/d(source coordinates) for the pointwise backward
    (photo_fwdg_kernel, 60-lane strips); without (torch.no_grad / nothing requires a gradient) the evaluation kernel runs
    (photo_fwd_kernel, DC_OPT_NO_GRAD, the smaller workspace).  Same formulas, two compilations: the compiler contracts
    multiply-adds differently where the training kernel has further uses of a product, so the log tensors agree to rounding
    (not bitwise), the argmin maps everywhere but on rounding-level ties, and the five losses to summation order (60- vs
    62-column strips)."""
    import ctypes
    from depthcore import ops, _lib
    dev = torch.device("cuda:0")
    inputs = R.synthetic_inputs(3, 64, 160, seed=4)          # ragged: 160 is not a multiple of 60, 62 or 64
    g = torch.Generator().manual_seed(5)
    disps = [torch.rand(3, 1, 64 >> s, 160 >> s, generator=g) for s in range(4)]
    Ts = random_poses(3, 11)
    noise = R.tiebreak_noise(3, 64, 160)
    if kw.get("avg_reprojection"):
        noise = [n[:, :1].contiguous() for n in noise]

    def run(grad):
        cfg = ops.PhotoConfig(
            inputs[("color", 0, 0)].to(dev), inputs[("color", -1, 0)].to(dev), inputs[("color", 1, 0)].to(dev),
            [inputs[("color", 0, s)].to(dev) for s in range(4)], inputs[("K", 0)].to(dev), inputs[("inv_K", 0)].to(dev),
            noise=[n.to(dev) for n in noise], materialize=True, **kw)
        d = [x.to(dev).requires_grad_(grad) for x in disps]
        t = [x.to(dev).requires_grad_(grad) for x in Ts]
        return ops.photometric_loss(cfg, t[0], t[1], d), cfg.extras
    (lt, et), (le, ee) = run(True), run(False)
    assert lt.requires_grad and not le.requires_grad
    close(lt.detach(), le, rtol=2e-6)
    for s in range(4):
        assert float((et["argmin"][s] != ee["argmin"][s]).float().mean()) <= 1e-4
        close(et["depth"][s], ee["depth"][s], rtol=1e-5)
        for f in range(2):
            close(et["color"][s][f], ee["color"][s][f], rtol=1e-5, atol=1e-6)
            close(et["sample"][s][f], ee["sample"][s][f], rtol=1e-5, atol=1e-5)
    # C ABI: the evaluation workspace is the smaller one, and dc_photo_bwd refuses an evaluation descriptor
    L = _lib.lib()
    d = _lib.PhotoDesc()
    d.B, d.H, d.W, d.num_scales = 3, 64, 160, 4
    prev = L.dc_set_photo_full(0)            # the round-4 split keeps 16 B per pixel and scale of d(loss)/d(source coords)
    try:
        w_split = L.dc_photo_workspace(ctypes.byref(d))
        L.dc_set_photo_full(1)               # the all-the-way forward does not: its workspace is the evaluation one to within the
        w_train = L.dc_photo_workspace(ctypes.byref(d))        # partial-sum tables (different strip widths)
    finally:
        L.dc_set_photo_full(prev)
    d.flags = _lib.OPT_NO_GRAD
    w_eval = L.dc_photo_workspace(ctypes.byref(d))
    assert w_eval < w_split and abs(w_train - w_eval) < 0.02 * w_eval and w_train < 0.6 * w_split


@pytest.mark.parametrize("tag,kw", VARIANTS[:2])
def test_packed_inputs_change_nothing(tag, kw):
    """dc_photo_desc.packed: the pixel-interleaved RGBx copies of the three frames supplied by the caller (the data step writes
    them, dc_data_to_rgbx / dc_pack_rgbx) instead of rebuilt inside every forward.  Same values in the same layout: losses,
    argmin maps and every gradient are bit-identical, and the workspace no longer holds the three copies."""
    import ctypes
    from depthcore import ops, _lib
    dev = torch.device("cuda:0")
    b, h, w = 3, 64, 160
    inputs = R.synthetic_inputs(b, h, w, seed=4)
    g = torch.Generator().manual_seed(5)
    disps = [torch.rand(b, 1, h >> s, w >> s, generator=g) for s in range(4)]
    Ts = random_poses(b, 11)
    noise = R.tiebreak_noise(b, h, w)
    frames = [inputs[("color", f, 0)].to(dev) for f in (0, -1, 1)]
    packed = [ops.pack_rgbx(x) for x in frames]
    for x, p in zip(frames, packed):
        assert p.shape == (b, h, w, 4) and torch.equal(p[..., :3].permute(0, 3, 1, 2), x) and not p[..., 3].any()

    def run(pk):
        cfg = ops.PhotoConfig(frames[0], frames[1], frames[2], [inputs[("color", 0, s)].to(dev) for s in range(4)],
                              inputs[("K", 0)].to(dev), inputs[("inv_K", 0)].to(dev), noise=[n.to(dev) for n in noise], packed=pk, **kw)
        d = [x.to(dev).requires_grad_() for x in disps]
        t = [x.to(dev).requires_grad_() for x in Ts]
        losses = ops.photometric_loss(cfg, t[0], t[1], d)
        grads = torch.autograd.grad(losses[4], d + t)
        return losses.detach(), cfg.extras["argmin"], grads
    (l0, a0, g0), (l1, a1, g1) = run(None), run(packed)
    assert torch.equal(l0, l1)
    assert all(torch.equal(x, y) for x, y in zip(a0, a1)) and all(torch.equal(x, y) for x, y in zip(g0, g1))
    L = _lib.lib()
    d = _lib.PhotoDesc()
    d.B, d.H, d.W, d.num_scales = b, h, w, 4
    w_plain = L.dc_photo_workspace(ctypes.byref(d))
    for k in range(3):
        d.packed[k] = packed[k].data_ptr()
    assert w_plain - L.dc_photo_workspace(ctypes.byref(d)) >= 3 * b * h * w * 16


@pytest.mark.parametrize("shape,ext_noise,materialize", [((2, 64, 96), True, False), ((2, 64, 96), False, True), ((3, 40, 72), True, True),
                                                          ((12, 192, 640), False, False)])
def test_all_the_way_forward_equals_the_split_chain(shape, ext_noise, materialize):
    """dc_set_photo_full: the training forward that contracts to d(loss)/d(upsampled disp) + pose sums (default) against the
    round-4 split (forward emits d(loss)/d(source coords), pointwise backward chains them): same losses and outputs bit for bit
    (the forward's loss arithmetic is untouched), gradients to rounding (same chain, the upstream weight applied after the
    linear transposed upsample instead of before; pose sums over different block shapes)."""
    from depthcore import _lib
    b, h, w = shape
    ns = 4 if h % 32 == 0 else 3
    inputs = R.synthetic_inputs(b, h, w, num_scales=ns, seed=5)
    g = torch.Generator().manual_seed(17)
    disps = [torch.rand(b, 1, h >> s, w >> s, generator=g) for s in range(ns)]
    Ts = random_poses(b, 8)
    noise = R.tiebreak_noise(b, h, w, num_scales=ns) if ext_noise else None
    L = _lib.lib()
    res = {}
    prev = L.dc_set_photo_full(1)
    try:
        for mode in (1, 0):
            L.dc_set_photo_full(mode)
            res[mode] = hip_photo(inputs, disps, Ts, noise, materialize=materialize)
    finally:
        L.dc_set_photo_full(prev)
    for a, c in zip(res[1][0], res[0][0]):
        assert torch.equal(a, c)
    for s in range(ns):
        assert torch.equal(res[1][1]["argmin"][s], res[0][1]["argmin"][s])
        # (v_rcp-based projection re-derived from the parked depth instead of the re-read disparity taps, cancellation in
        # d(depth): 1e-5 relative in the norm, no outliers)
        assert rel_l2(res[1][2][s], res[0][2][s]) < 1e-4, (s, rel_l2(res[1][2][s], res[0][2][s]))
        close_frac(res[1][2][s], res[0][2][s], rtol=1e-3, atol=0, atol_rel=1e-3, bad=1e-3, msg="gdisp%d" % s)
    for f in range(2):
        assert rel_l2(res[1][3][f][:, :3, :], res[0][3][f][:, :3, :]) < 1e-4, f


@pytest.mark.parametrize("fwd_mode", [1, 0])
def test_photo_mode_is_pinned_at_the_forward(fwd_mode):
    """dc_set_photo_full toggled BETWEEN a forward and its backward (another thread, a test fixture): the op read the mode once at
    the forward and pinned it in both descs (DC_OPT_PHOTO_FULL / DC_OPT_PHOTO_SPLIT), so the backward carves the workspace the
    forward wrote -- gradients bit-identical to an undisturbed run; a desc carrying both bits is refused."""
    import ctypes
    from depthcore import _lib, ops
    b, h, w = 2, 64, 96
    inputs = R.synthetic_inputs(b, h, w, seed=5)
    g = torch.Generator().manual_seed(17)
    disps = [torch.rand(b, 1, h >> s, w >> s, generator=g) for s in range(4)]
    Ts = random_poses(b, 8)
    noise = R.tiebreak_noise(b, h, w)
    L = _lib.lib()
    prev = L.dc_set_photo_full(fwd_mode)
    try:
        want = hip_photo(inputs, disps, Ts, noise)
        dev = torch.device("cuda:0")
        cfg = ops.PhotoConfig(
            inputs[("color", 0, 0)].to(dev), inputs[("color", -1, 0)].to(dev), inputs[("color", 1, 0)].to(dev),
            [inputs[("color", 0, s)].to(dev) for s in range(4)], inputs[("K", 0)].to(dev), inputs[("inv_K", 0)].to(dev),
            noise=[n.to(dev) for n in noise])
        d = [x.to(dev).requires_grad_() for x in disps]
        t = [x.to(dev).requires_grad_() for x in Ts]
        losses = ops.photometric_loss(cfg, t[0], t[1], d)
        L.dc_set_photo_full(1 - fwd_mode)            # ... the setter moves under the op's feet
        grads = torch.autograd.grad(losses[4], d + t)
        torch.cuda.synchronize()
    finally:
        L.dc_set_photo_full(prev)
    for a, c in zip(grads[:4], want[2]):
        assert torch.equal(a, c)
    for a, c in zip(grads[4:], want[3]):
        assert torch.equal(a, c)
    ws = torch.empty(1 << 26, dtype=torch.uint8, device=dev)
    lo = torch.empty(5, device=dev)
    am = [torch.empty(b, h, w, dtype=torch.uint8, device=dev) for _ in range(4)]
    for mode, rc in ((_lib.OPT_PHOTO_FULL, 0), (_lib.OPT_PHOTO_FULL | _lib.OPT_PHOTO_SPLIT, -1)):      # -1 = DC_EINVAL
        desc = ops._fill_desc(cfg, t[0].detach(), t[1].detach(), [x.detach() for x in d], mode=mode)
        desc.workspace, desc.workspace_bytes, desc.losses = ws.data_ptr(), ws.numel(), lo.data_ptr()
        for s in range(4):
            desc.argmin[s] = am[s].data_ptr()
        assert L.dc_photo_fwd(ctypes.byref(desc), None) == rc
    torch.cuda.synchronize()
