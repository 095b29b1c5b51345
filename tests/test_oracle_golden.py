"""The CPU oracle (oracle/) against the golden vectors captured from the reference.

Golden vectors: tests/golden/*.npz, written by tests/golden/make_golden.py by
running /root/reference's layers.py, depth_decoder.py, pose_decoder.py and the
Trainer hot methods (trainer.py:465-622) in the build container.
"""
import numpy as np
import pytest
import torch

from oracle import ref_cpu as R
import make_golden as MG

B, H, W = MG.B, MG.H, MG.W


def T(a):
    return torch.from_numpy(np.asarray(a))


def close(a, b, rtol=1e-5, atol=1e-6):
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    np.testing.assert_allclose(a, np.asarray(b), rtol=rtol, atol=atol)


def close_frac(a, b, rtol, atol, bad=5e-3):
    """Pointwise tolerance with a small budget of outliers: floor / argmin / clamp make the
    path piecewise-smooth, so 1-ulp input differences legitimately flip isolated pixels."""
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = np.asarray(b)
    viol = np.abs(a - b) > atol + rtol * np.abs(b)
    assert viol.mean() <= bad, "%.4f%% of elements out of tolerance" % (100 * viol.mean())


def test_transformation_from_parameters(golden):
    g = golden["layers_ops"]
    for inv in (0, 1):
        aa = T(g["tfp_aa"]).requires_grad_()
        tr = T(g["tfp_tr"]).requires_grad_()
        M = R.transformation_from_parameters(aa, tr, invert=bool(inv))
        close(M, g["tfp_M_inv%d" % inv])
        ga, gt = torch.autograd.grad((M * T(g["tfp_cot"])).sum(), [aa, tr])
        close(ga, g["tfp_gaa_inv%d" % inv], atol=2e-6)
        close(gt, g["tfp_gtr_inv%d" % inv], atol=2e-6)
    close(R.rot_from_axisangle(torch.zeros(1, 1, 3)), g["rot_zero"])


def test_known_answers():
    sd, d = R.disp_to_depth(torch.tensor([0.0, 1.0]), 0.1, 100.0)
    close(sd, [0.01, 10.0]); close(d, [100.0, 0.1], rtol=1e-6)
    Rz = R.rot_from_axisangle(torch.tensor([[[0.0, 0.0, np.pi / 2]]]))[0, :3, :3]
    close(Rz, [[0, -1, 0], [1, 0, 0], [0, 0, 1]], atol=1e-6)
    t = torch.tensor([[[1.0, 2.0, 3.0]]])
    M = R.transformation_from_parameters(torch.zeros(1, 1, 3), t)
    close(M[0, :3, 3], [1, 2, 3]); close(M[0, :3, :3], np.eye(3))
    Mi = R.transformation_from_parameters(torch.zeros(1, 1, 3), t, invert=True)
    close(Mi[0, :3, 3], [-1, -2, -3])
    x = torch.rand(1, 3, 8, 9)
    assert float(R.ssim(x, x).abs().max()) < 1e-6
    assert float(R.smooth_loss(torch.full((1, 1, 8, 9), 0.3), x)) == 0.0
    inp = R.synthetic_inputs(1, 192, 640)
    assert abs(float(inp[("inv_K", 0)][0, 0, 0]) - 1 / (0.58 * 640)) < 1e-9


def test_disp_to_depth(golden):
    g = golden["layers_ops"]
    sd, d = R.disp_to_depth(T(g["d2d_disp"]), 0.1, 100.0)
    close(sd, g["d2d_scaled"], rtol=1e-6); close(d, g["d2d_depth"], rtol=1e-6)


def test_pix_coords_bit_exact(golden):
    g = golden["layers_ops"]
    assert np.array_equal(R.pix_coords(B, H, W).numpy(), g["pix_coords"])
    assert bool(g["pix_coords_192x640_exact"])
    big = R.pix_coords(1, 192, 640).numpy()[0]
    assert np.array_equal(big[:, 639:642], g["pix_coords_192x640_cols639_642"])
    assert np.array_equal(big[:, 639:642], np.array([[639, 0, 1], [0, 1, 1], [1, 1, 1]], np.float32))


def test_backproject_project_warp(golden):
    g = golden["layers_ops"]
    depth = T(g["geo_depth"]).requires_grad_()
    Tm = T(g["geo_T"]).requires_grad_()
    K, invK = T(g["geo_K"]), T(g["geo_invK"])
    cam = R.backproject(depth, invK)
    close(cam, g["geo_cam"], rtol=1e-5, atol=1e-5)
    grid = R.project3d(cam, K, Tm, H, W)
    close(grid, g["geo_grid"], rtol=1e-5, atol=2e-6)
    warped = R.grid_sample_border(T(g["geo_img"]), grid)
    close(warped, g["geo_warped"], rtol=1e-4, atol=2e-5)
    gd, gT = torch.autograd.grad((warped * T(g["geo_cot"])).sum(), [depth, Tm])
    close(gd, g["geo_gdepth"], rtol=2e-3, atol=2e-4)
    close(gT, g["geo_gT"], rtol=2e-3, atol=2e-2)
    # identity pose => linspace grid (SURVEY 8c KAT)
    gid = R.project3d(R.backproject(depth.detach(), invK), K, torch.eye(4).expand(B, 4, 4), H, W)
    close(gid, g["geo_grid_identity"], atol=2e-6)
    lx = torch.linspace(-1, 1, W)
    ly = torch.linspace(-1, 1, H)
    assert float((gid[..., 0] - lx.view(1, 1, W)).abs().max()) < 2e-6
    assert float((gid[..., 1] - ly.view(1, H, 1)).abs().max()) < 2e-6


def test_grid_sample_matches_aten_incl_clamped_grad():
    """The explicit-gather restatement vs ATen's grid_sample on far-out-of-range grids."""
    g = torch.Generator().manual_seed(3)
    img = torch.rand(2, 3, 7, 9, generator=g)
    grid = (torch.rand(2, 5, 6, 2, generator=g) * 3 - 1.5).requires_grad_()
    grid2 = grid.detach().clone().requires_grad_()
    a = R.grid_sample_border(img, grid)
    b = torch.nn.functional.grid_sample(img, grid2, padding_mode="border", align_corners=False)
    close(a, b.detach(), atol=1e-6)
    cot = torch.rand(a.shape, generator=g)
    (ga,) = torch.autograd.grad((a * cot).sum(), [grid])
    (gb,) = torch.autograd.grad((b * cot).sum(), [grid2])
    close(ga, gb, atol=1e-5)
    assert (gb == 0).any()


def test_upsample_bilinear(golden):
    g = golden["layers_ops"]
    x = T(g["up_in"]).requires_grad_()
    up = R.upsample_bilinear(x, H, W)
    close(up, g["up_out"], atol=1e-6)
    (gx,) = torch.autograd.grad((up * T(g["up_cot"])).sum(), [x])
    close(gx, g["up_gin"], rtol=1e-4, atol=1e-5)
    same = torch.rand(1, 1, 5, 6)
    assert torch.equal(R.upsample_bilinear(same, 5, 6), same)


def test_ssim_and_smooth(golden):
    g = golden["layers_ops"]
    x = T(g["ssim_x"]).requires_grad_()
    s = R.ssim(x, T(g["ssim_y"]))
    close(s, g["ssim_out"], rtol=1e-4, atol=2e-6)
    (gx,) = torch.autograd.grad((s * T(g["ssim_cot"])).sum(), [x])
    close(gx, g["ssim_gx"], rtol=1e-3, atol=1e-4)
    d = T(g["smooth_disp"]).requires_grad_()
    sm = R.smooth_loss(d, T(g["smooth_img"]))
    close(sm, g["smooth_out"], rtol=1e-5)
    (gd,) = torch.autograd.grad(sm, [d])
    close(gd, g["smooth_gdisp"], rtol=1e-4, atol=1e-9)


def test_conv_block(golden):
    g = golden["layers_ops"]
    x = T(g["cb_x"]).requires_grad_()
    w = T(g["cb_w"]).requires_grad_()
    b = T(g["cb_b"]).requires_grad_()
    y = R.upsample_nearest2(R.conv_block(x, w, b))
    close(y, g["cb_out"], rtol=1e-5, atol=1e-6)
    gx, gw, gb = torch.autograd.grad((y * T(g["cb_cot"])).sum(), [x, w, b])
    close(gx, g["cb_gx"], rtol=1e-4, atol=1e-5)
    close(gw, g["cb_gw"], rtol=1e-4, atol=1e-4)
    close(gb, g["cb_gb"], rtol=1e-4, atol=1e-4)


def run_trainer_oracle(g, tag, **opt_kw):
    opt = R.Opt(height=H, width=W, **opt_kw)
    inputs = R.synthetic_inputs(B, H, W, seed=0)
    p = tag + "_"
    disp = [T(g[p + "disp%d" % s]).requires_grad_() for s in range(4)]
    aa = {f: T(g[p + "aa_%d" % f]).requires_grad_() for f in (-1, 1)}
    tr = {f: T(g[p + "tr_%d" % f]).requires_grad_() for f in (-1, 1)}
    outputs = {("disp", s): disp[s] for s in range(4)}
    for f in (-1, 1):
        outputs[("cam_T_cam", 0, f)] = R.transformation_from_parameters(aa[f], tr[f], invert=(f < 0))
    R.generate_images_pred(inputs, outputs, opt)
    noise = R.tiebreak_noise(B, H, W)
    if opt.avg_reprojection:
        g2 = torch.Generator().manual_seed(1234)
        noise = [torch.randn(B, 1, H, W, generator=g2) for _ in range(4)]
    losses = R.compute_losses(inputs, outputs, opt, noise)
    leaves = disp + [aa[-1], aa[1], tr[-1], tr[1]]
    grads = torch.autograd.grad(losses["loss"], leaves)
    return outputs, losses, grads


@pytest.mark.parametrize("tag,kw", [("auto", {}), ("noauto", dict(disable_automasking=True)),
                                    ("avg", dict(avg_reprojection=True)), ("nossim", dict(no_ssim=True))])
def test_trainer_level(golden, tag, kw):
    g = golden["trainer_losses"]
    outputs, losses, grads = run_trainer_oracle(g, tag, **kw)
    p = tag + "_"
    close(losses["loss"], g[p + "loss"], rtol=2e-6)
    for s in range(4):
        close(losses["loss/%d" % s], g[p + "loss%d" % s], rtol=2e-6)
        close_frac(grads[s], g[p + "gdisp%d" % s], rtol=1e-3, atol=2e-8)
        if not kw.get("disable_automasking"):
            sel = outputs["identity_selection/%d" % s].numpy().astype(np.uint8)
            want = np.unpackbits(g[p + "idsel%d" % s])[:sel.size].reshape(sel.shape)
            assert (sel != want).mean() < 1e-4
    for j, f in enumerate((-1, 1)):
        # pose grads sum ill-conditioned SSIM derivatives (E[x^2]-mu^2 cancellation over a
        # C2=9e-4 denominator) over every pixel: fp32 evaluation-order noise is ~0.5% here.
        close(grads[4 + j], g[p + "gaa_%d" % f], rtol=1e-2, atol=3e-4)
        close(grads[6 + j], g[p + "gtr_%d" % f], rtol=1e-2, atol=3e-4)
        close(outputs[("cam_T_cam", 0, f)], g[p + "T_%d" % f], atol=1e-7)
    if tag == "auto":
        for s in (0, 3):
            close(outputs[("depth", 0, s)], g[p + "depth%d" % s], rtol=1e-5)
            for f in (-1, 1):
                close(outputs[("sample", f, s)], g[p + "sample_%d_%d" % (f, s)], atol=3e-6)
                close(outputs[("color", f, s)], g[p + "color_%d_%d" % (f, s)], atol=2e-4)


def test_decoders(golden):
    g = golden["decoders"]
    from self_check_shapes import dec_state, pose_state
    nce = np.array([64, 64, 128, 256, 512])
    sd = dec_state(nce, 3)
    assert list(sd.keys()) == list(g["dec_keys"])
    assert [str(tuple(v.shape)) for v in sd.values()] == list(g["dec_shapes"])
    feats, gen = MG.decoder_features(nce)
    feats = [f.requires_grad_() for f in feats]
    sd = {k: v.requires_grad_() for k, v in sd.items()}
    o = R.depth_decoder_forward(sd, feats, nce)
    tot = 0
    for s in range(4):
        close(o[("disp", s)], g["dec_disp%d" % s], rtol=1e-4, atol=1e-5)
        tot = tot + (o[("disp", s)] * T(g["dec_cot%d" % s])).sum()
    names = list(sd)
    grads = torch.autograd.grad(tot, feats + [sd[k] for k in names])
    for i in range(5):
        got = grads[i] if i >= 3 else torch.from_numpy(MG.summ(grads[i]))
        close(got, g["dec_gfeat%d" % i], rtol=1e-3, atol=1e-4)
    for j, k in enumerate(names):
        gk = grads[5 + j]
        got = gk if gk.numel() <= 4096 else torch.from_numpy(MG.summ(gk))
        close(got, g["dec_g_" + k], rtol=2e-3, atol=2e-3)
    o2 = R.depth_decoder_forward({k: v.detach() for k, v in sd.items()}, [f.detach() for f in feats], nce,
                                 pre_disp=True)
    close(MG.summ(o2[("disp", 0)]), g["dec_predisp0"], rtol=1e-4, atol=1e-4)
    # pose decoder
    ps = pose_state(nce, 5)
    assert list(ps.keys()) == list(g["pose_keys"])
    ps = {k: v.requires_grad_() for k, v in ps.items()}
    f4 = T(g["pose_feat"]).requires_grad_()
    a, t = R.pose_decoder_forward(ps, [[f4]], 2)
    close(a, g["pose_aa"], rtol=1e-4, atol=1e-6); close(t, g["pose_tr"], rtol=1e-4, atol=1e-6)
    pn = list(ps)
    gr = torch.autograd.grad((a * T(g["pose_cot_aa"])).sum() + (t * T(g["pose_cot_tr"])).sum(),
                             [f4] + [ps[k] for k in pn])
    close(gr[0], g["pose_gfeat"], rtol=1e-3, atol=1e-6)
    for j, k in enumerate(pn):
        gk = gr[1 + j]
        got = gk if gk.numel() <= 4096 else torch.from_numpy(MG.summ(gk))
        close(got, g["pose_g_" + k], rtol=2e-3, atol=1e-4)


def test_pose_decoder_even_fixture(golden):
    """Round-2 fixture (tests/golden/make_golden_r2.py): the reference PoseDecoder on an even-sized (4x8) feature map."""
    import make_golden_r2 as MG2
    from self_check_shapes import pose_state
    g = golden["pose_even"]
    nce = np.array([64, 64, 128, 256, 512])
    ps = {k: v.requires_grad_() for k, v in pose_state(nce, 5).items()}
    f4, _ = MG2.pose_even_feature()
    f4.requires_grad_()
    a, t = R.pose_decoder_forward(ps, [[f4]], 2)
    close(a, g["aa"], rtol=1e-4, atol=1e-6); close(t, g["tr"], rtol=1e-4, atol=1e-6)
    pn = list(ps)
    gr = torch.autograd.grad((a * T(g["cot_aa"])).sum() + (t * T(g["cot_tr"])).sum(), [f4] + [ps[k] for k in pn])
    close(MG.summ(gr[0]), g["gfeat"], rtol=1e-3, atol=1e-6)
    for j, k in enumerate(pn):
        gk = gr[1 + j]
        got = gk if gk.numel() <= 4096 else torch.from_numpy(MG.summ(gk))
        close(got, g["g_" + k], rtol=2e-3, atol=1e-4)


def _ablation_setup(tag):
    import make_golden_r2 as MG2
    B, H, W = MG.B, MG.H, MG.W
    inputs = R.synthetic_inputs(B, H, W, seed=0)
    disp, aa, tr, mask = MG2.ablation_inputs(tag)
    kw = dict(v1_multiscale=True) if tag == "v1ms" else dict(disable_automasking=True, predictive_mask=True)
    opt = R.Opt(height=H, width=W, **kw)
    torch.manual_seed(1234)          # trainer.py:594-595 draws with the default CPU generator, one tensor per scale
    noise = None if tag == "pmask" else [torch.randn(B, 2, H >> s, W >> s) for s in range(4)]
    return inputs, disp, aa, tr, mask, opt, noise


@pytest.mark.parametrize("tag", ["v1ms", "pmask"])
def test_trainer_ablations(golden, tag):
    """a15 ablations `v1_multiscale` (trainer.py:471,541) and `predictive_mask` (:571-590) vs the reference's outputs."""
    g = golden["trainer_ablations"]
    inputs, disp, aa, tr, mask, opt, noise = _ablation_setup(tag)
    leaves = [disp[s].requires_grad_() for s in range(4)] + [aa[-1].requires_grad_(), aa[1].requires_grad_(),
                                                             tr[-1].requires_grad_(), tr[1].requires_grad_()]
    outputs = {("disp", s): disp[s] for s in range(4)}
    if mask is not None:
        outputs["predictive_mask"] = {("disp", s): mask[s].requires_grad_() for s in range(4)}
        leaves += [mask[s] for s in range(4)]
    for f in (-1, 1):
        outputs[("cam_T_cam", 0, f)] = R.transformation_from_parameters(aa[f], tr[f], invert=(f < 0))
        close(outputs[("cam_T_cam", 0, f)], g[tag + "_T_%d" % f], rtol=1e-5, atol=1e-7)
    R.generate_images_pred(inputs, outputs, opt)
    losses = R.compute_losses(inputs, outputs, opt, noise)
    close(losses["loss"], g[tag + "_loss"], rtol=1e-5)
    grads = torch.autograd.grad(losses["loss"], leaves)
    for s in range(4):
        close(losses["loss/%d" % s], g[tag + "_loss%d" % s], rtol=1e-5)
        close_frac(grads[s], g[tag + "_gdisp%d" % s], rtol=1e-3, atol=1e-7, bad=1e-2)
        if mask is not None:
            close(grads[8 + s], g[tag + "_gmask%d" % s], rtol=1e-3, atol=1e-8)
    for j, f in enumerate((-1, 1)):
        for k, gi in (("gaa", 4 + j), ("gtr", 6 + j)):
            want = T(g[tag + "_%s_%d" % (k, f)])
            assert float((grads[gi] - want).norm() / want.norm()) < 2e-2
