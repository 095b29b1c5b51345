#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE.

Run in the build container only (the reference does not travel):

    python tests/golden/make_golden.py [--reference /root/reference]

It imports the reference's own `layers.py`, loads `networks/depth_decoder.py`
and `networks/pose_decoder.py` by file path (the `networks` package itself needs
torchvision, which is not installed), and drives the unbound
`Trainer.generate_images_pred` / `compute_reprojection_loss` / `compute_losses`
methods on a hand-built instance.  `trainer.py` imports logging / dataset
packages that are absent here and are not on the hot path (tensorboardX, GPUtil,
IPython, datasets, networks); empty placeholder modules satisfy those imports.

Fixtures hold only plain arrays (inputs, expected outputs, expected gradients);
no reference source, bytecode or pickled reference classes.
"""
import argparse
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
from oracle import ref_cpu  # noqa: E402  (input recipe only: synthetic batch + intrinsics)

B, H, W = 2, 64, 96


def t2n(t):
    return t.detach().cpu().numpy()


def summ(t, n=512):
    """Compact pin for a large tensor: [sum, sum|.|, sum sq] in fp64 + n strided samples."""
    a = t2n(t).astype(np.float64).ravel()
    step = max(1, a.size // n)
    return np.concatenate([[a.sum(), np.abs(a).sum(), (a * a).sum()], a[::step][:n]])


def seeded_state(module, seed, scale=0.05):
    """Deterministic weights by OUR recipe (so tests can rebuild them without the reference)."""
    g = torch.Generator().manual_seed(seed)
    sd = module.state_dict()
    for k in sd:
        sd[k] = scale * torch.randn(sd[k].shape, generator=g)
    module.load_state_dict(sd)


def load_reference(ref_root):
    sys.path.insert(0, ref_root)
    import layers as ref_layers  # the reference's layers.py
    for name in ("tensorboardX", "GPUtil", "IPython", "datasets", "networks"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["tensorboardX"].SummaryWriter = object
    sys.modules["IPython"].embed = lambda *a, **k: None
    import trainer as ref_trainer

    def by_path(modname, rel):
        spec = importlib.util.spec_from_file_location(modname, os.path.join(ref_root, rel))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod

    dd = by_path("ref_depth_decoder", "networks/depth_decoder.py")
    pd = by_path("ref_pose_decoder", "networks/pose_decoder.py")
    return ref_layers, ref_trainer, dd, pd


def make_trainer(ref_layers, ref_trainer, opt_kw=None):
    T = ref_trainer.Trainer.__new__(ref_trainer.Trainer)
    opt = types.SimpleNamespace(
        height=H, width=W, scales=[0, 1, 2, 3], min_depth=0.1, max_depth=100.0,
        disparity_smoothness=1e-3, frame_ids=[0, -1, 1], v1_multiscale=False,
        avg_reprojection=False, disable_automasking=False, predictive_mask=False,
        no_ssim=False, pose_model_type="separate_resnet", batch_size=B)
    for k, v in (opt_kw or {}).items():
        setattr(opt, k, v)
    T.opt = opt
    T.device = torch.device("cpu")
    T.num_scales = 4
    T.ssim = ref_layers.SSIM()
    T.backproject_depth = {}
    T.project_3d = {}
    for s in opt.scales:
        h, w = H // 2 ** s, W // 2 ** s
        T.backproject_depth[s] = ref_layers.BackprojectDepth(B, h, w)
        T.project_3d[s] = ref_layers.Project3D(B, h, w)
    return T


def gen_layers(L, out):
    g = torch.Generator().manual_seed(11)
    # a5
    aa = (0.3 * torch.randn(B, 1, 3, generator=g)).requires_grad_()
    tr = (0.5 * torch.randn(B, 1, 3, generator=g)).requires_grad_()
    cot = torch.randn(B, 4, 4, generator=g)
    for inv in (False, True):
        M = L.transformation_from_parameters(aa, tr, invert=inv)
        ga, gt = torch.autograd.grad((M * cot).sum(), [aa, tr])
        out["tfp_M_inv%d" % inv] = t2n(M)
        out["tfp_gaa_inv%d" % inv] = t2n(ga)
        out["tfp_gtr_inv%d" % inv] = t2n(gt)
    out["tfp_aa"], out["tfp_tr"], out["tfp_cot"] = t2n(aa), t2n(tr), t2n(cot)
    out["rot_zero"] = t2n(L.rot_from_axisangle(torch.zeros(1, 1, 3)))
    # a6
    disp = torch.rand(B, 1, H, W, generator=g)
    sd, dep = L.disp_to_depth(disp, 0.1, 100.0)
    out["d2d_disp"], out["d2d_scaled"], out["d2d_depth"] = t2n(disp), t2n(sd), t2n(dep)
    # a7 / a8 / a9
    bp = L.BackprojectDepth(B, H, W)
    pj = L.Project3D(B, H, W)
    out["pix_coords"] = t2n(bp.pix_coords)
    big = L.BackprojectDepth(1, 192, 640)
    pcb = t2n(big.pix_coords[0])
    i = np.arange(192 * 640)
    out["pix_coords_192x640_exact"] = np.array(
        bool(np.array_equal(pcb[0], (i % 640).astype(np.float32))
             and np.array_equal(pcb[1], (i // 640).astype(np.float32))
             and np.all(pcb[2] == 1.0)))
    out["pix_coords_192x640_cols639_642"] = pcb[:, 639:642]
    inp = ref_cpu.synthetic_inputs(B, H, W, seed=5)
    K, invK = inp[("K", 0)], inp[("inv_K", 0)]
    depth = (dep.clone() * 0.2 + 1.0).requires_grad_()
    T = L.transformation_from_parameters(0.02 * torch.randn(B, 1, 3, generator=g),
                                         0.05 * torch.randn(B, 1, 3, generator=g)).requires_grad_()
    cam = bp(depth, invK)
    grid = pj(cam, K, T)
    img = inp[("color", -1, 0)]
    warped = torch.nn.functional.grid_sample(img, grid, padding_mode="border")
    cw = torch.randn(B, 3, H, W, generator=g)
    gdepth, gT = torch.autograd.grad((warped * cw).sum(), [depth, T])
    out.update(geo_K=t2n(K), geo_invK=t2n(invK), geo_depth=t2n(depth), geo_T=t2n(T), geo_cam=t2n(cam),
               geo_grid=t2n(grid), geo_img=t2n(img), geo_warped=t2n(warped), geo_cot=t2n(cw),
               geo_gdepth=t2n(gdepth), geo_gT=t2n(gT))
    # identity pose KAT
    gid = pj(bp(depth.detach(), invK), K, torch.eye(4).expand(B, 4, 4))
    out["geo_grid_identity"] = t2n(gid)
    # a10
    small = torch.rand(B, 1, H // 4, W // 4, generator=g).requires_grad_()
    up = torch.nn.functional.interpolate(small, [H, W], mode="bilinear", align_corners=False)
    cu = torch.randn(B, 1, H, W, generator=g)
    (gs,) = torch.autograd.grad((up * cu).sum(), [small])
    out.update(up_in=t2n(small), up_out=t2n(up), up_cot=t2n(cu), up_gin=t2n(gs))
    # a11 / a12
    x = inp[("color", 1, 0)].clone().requires_grad_()
    y = inp[("color", 0, 0)]
    s = L.SSIM()(x, y)
    cs = torch.randn(B, 3, H, W, generator=g)
    (gx,) = torch.autograd.grad((s * cs).sum(), [x])
    out.update(ssim_x=t2n(x), ssim_y=t2n(y), ssim_out=t2n(s), ssim_cot=t2n(cs), ssim_gx=t2n(gx))
    # a13
    d = torch.rand(B, 1, H, W, generator=g).requires_grad_()
    sm = L.get_smooth_loss(d, y)
    (gd,) = torch.autograd.grad(sm, [d])
    out.update(smooth_disp=t2n(d), smooth_img=t2n(y), smooth_out=t2n(sm), smooth_gdisp=t2n(gd))
    # a3: ConvBlock + upsample
    torch.manual_seed(21)
    cb = L.ConvBlock(5, 7)
    xin = torch.randn(B, 5, 10, 12, generator=g).requires_grad_()
    yo = L.upsample(cb(xin))
    cc = torch.randn(yo.shape, generator=g)
    gi, gw, gb = torch.autograd.grad((yo * cc).sum(), [xin, cb.conv.conv.weight, cb.conv.conv.bias])
    out.update(cb_x=t2n(xin), cb_w=t2n(cb.conv.conv.weight), cb_b=t2n(cb.conv.conv.bias), cb_out=t2n(yo),
               cb_cot=t2n(cc), cb_gx=t2n(gi), cb_gw=t2n(gw), cb_gb=t2n(gb))


def gen_trainer(L, TR, out, tag, opt_kw=None):
    """Trainer-level: fixed disp + pose params -> losses, masks, grads."""
    T = make_trainer(L, TR, opt_kw)
    inp = ref_cpu.synthetic_inputs(B, H, W, seed=0)
    g = torch.Generator().manual_seed(77)
    disp = {s: torch.rand(B, 1, H >> s, W >> s, generator=g).requires_grad_() for s in range(4)}
    aa = {f: (0.01 * torch.randn(B, 1, 3, generator=g)).requires_grad_() for f in (-1, 1)}
    tr = {f: (0.01 * torch.randn(B, 1, 3, generator=g)).requires_grad_() for f in (-1, 1)}
    outputs = {("disp", s): disp[s] for s in range(4)}
    for f in (-1, 1):
        outputs[("cam_T_cam", 0, f)] = L.transformation_from_parameters(aa[f], tr[f], invert=(f < 0))
    inputs = dict(inp)
    T.generate_images_pred(inputs, outputs)
    torch.manual_seed(1234)                      # the tie-break randn of trainer.py:594
    losses = T.compute_losses(inputs, outputs)
    # the very same draws, to store them
    torch.manual_seed(1234)
    auto = not T.opt.disable_automasking
    nch = 1 if T.opt.avg_reprojection else 2
    noise = [torch.randn(B, nch, H, W) for _ in range(4)] if auto else []
    leaves = [disp[s] for s in range(4)] + [aa[-1], aa[1], tr[-1], tr[1]]
    grads = torch.autograd.grad(losses["loss"], leaves)
    p = tag + "_"
    for s in range(4):
        out[p + "disp%d" % s] = t2n(disp[s])
        out[p + "gdisp%d" % s] = t2n(grads[s])
        out[p + "loss%d" % s] = t2n(losses["loss/%d" % s])
        if auto:
            out[p + "idsel%d" % s] = np.packbits(t2n(outputs["identity_selection/%d" % s]).astype(np.uint8))
            assert torch.equal(noise[s], ref_cpu.tiebreak_noise(B, H, W)[s][:, :nch]) or nch == 1
        if tag == "auto" and s in (0, 3):
            out[p + "depth%d" % s] = t2n(outputs[("depth", 0, s)])
            for f in (-1, 1):
                out[p + "color_%d_%d" % (f, s)] = t2n(outputs[("color", f, s)])
                out[p + "sample_%d_%d" % (f, s)] = t2n(outputs[("sample", f, s)])
    out[p + "loss"] = t2n(losses["loss"])
    for j, f in enumerate((-1, 1)):
        out[p + "aa_%d" % f] = t2n(aa[f])
        out[p + "tr_%d" % f] = t2n(tr[f])
        out[p + "gaa_%d" % f] = t2n(grads[4 + j])
        out[p + "gtr_%d" % f] = t2n(grads[6 + j])
        out[p + "T_%d" % f] = t2n(outputs[("cam_T_cam", 0, f)])
    out[p + "seed_inputs"] = np.array(0)


def decoder_features(num_ch_enc, seed=4, h5=2, w5=3):
    """Input recipe shared with the tests (plain seeded randn; no reference involved)."""
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(B, int(c), h5 * 2 ** (4 - i), w5 * 2 ** (4 - i), generator=g)
            for i, c in enumerate(num_ch_enc)], g


def gen_decoders(DD, PD, out):
    num_ch_enc = np.array([64, 64, 128, 256, 512])
    dec = DD.DepthDecoder(num_ch_enc)
    seeded_state(dec, 3)
    out["dec_keys"] = np.array(list(dec.state_dict().keys()))
    out["dec_shapes"] = np.array([str(tuple(v.shape)) for v in dec.state_dict().values()])
    feats, g = decoder_features(num_ch_enc)
    feats = [f.requires_grad_() for f in feats]
    o = dec(feats)
    cots = {s: torch.randn(o[("disp", s)].shape, generator=g) for s in range(4)}
    tot = sum((o[("disp", s)] * cots[s]).sum() for s in range(4))
    params = dict(dec.named_parameters())
    pnames = list(params)
    grads = torch.autograd.grad(tot, feats + [params[k] for k in pnames])
    for i in range(5):
        out["dec_gfeat%d" % i] = t2n(grads[i]) if i >= 3 else summ(grads[i])
    for s in range(4):
        out["dec_disp%d" % s] = t2n(o[("disp", s)])
        out["dec_cot%d" % s] = t2n(cots[s])
    for j, k in enumerate(pnames):
        gk = grads[5 + j]
        out["dec_g_" + k] = t2n(gk) if gk.numel() <= 4096 else summ(gk)
    o2 = dec([f.detach() for f in feats], pre_disp=True)
    out["dec_predisp0"] = summ(o2[("disp", 0)])
    # pose decoder
    pose = PD.PoseDecoder(num_ch_enc, num_input_features=1, num_frames_to_predict_for=2)
    seeded_state(pose, 5)
    out["pose_keys"] = np.array(list(pose.state_dict().keys()))
    f4 = torch.randn(B, 512, 2, 3, generator=g).requires_grad_()
    a, t = pose([[f4]])
    ca, ct = torch.randn(a.shape, generator=g), torch.randn(t.shape, generator=g)
    pp = dict(pose.named_parameters())
    pn = list(pp)
    gr = torch.autograd.grad((a * ca).sum() + (t * ct).sum(), [f4] + [pp[k] for k in pn])
    out.update(pose_feat=t2n(f4), pose_aa=t2n(a), pose_tr=t2n(t), pose_cot_aa=t2n(ca), pose_cot_tr=t2n(ct),
               pose_gfeat=t2n(gr[0]))
    for j, k in enumerate(pn):
        gk = gr[1 + j]
        out["pose_g_" + k] = t2n(gk) if gk.numel() <= 4096 else summ(gk)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    args = ap.parse_args()
    torch.set_num_threads(4)
    L, TR, DD, PD = load_reference(args.reference)

    lay = {}
    gen_layers(L, lay)
    np.savez_compressed(os.path.join(HERE, "layers_ops.npz"), **lay)

    tr = {}
    gen_trainer(L, TR, tr, "auto")
    gen_trainer(L, TR, tr, "noauto", dict(disable_automasking=True))
    gen_trainer(L, TR, tr, "avg", dict(avg_reprojection=True))
    gen_trainer(L, TR, tr, "nossim", dict(no_ssim=True))
    np.savez_compressed(os.path.join(HERE, "trainer_losses.npz"), **tr)

    dec = {}
    gen_decoders(DD, PD, dec)
    np.savez_compressed(os.path.join(HERE, "decoders.npz"), **dec)
    for f in ("layers_ops.npz", "trainer_losses.npz", "decoders.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KiB")
    print("auto loss", tr["auto_loss"], "noauto", tr["noauto_loss"])


if __name__ == "__main__":
    main()
