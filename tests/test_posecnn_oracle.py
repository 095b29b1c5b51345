"""oracle/ref_cpu.py against the reference-generated trainer-level fixture of `pose_model_type == "posecnn"`
(tests/golden/trainer_posecnn.npz, tests/golden/make_golden_r6.py): generate_images_pred rescales the translation of every
(scale, frame) by that scale's mean inverse depth and rebuilds the pose matrix (reference trainer.py:490-499)."""
import os

import numpy as np
import pytest
import torch

from oracle import ref_cpu as R
import make_golden as MG
from helpers import T, close, close_frac, rel_l2

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "trainer_posecnn.npz"))
B, H, W = MG.B, MG.H, MG.W
TAGS = [("cnn", {}), ("cnn_v1", dict(v1_multiscale=True))]


def noise_for(kw):
    torch.manual_seed(1234)          # trainer.py:594-595: the default CPU generator, one draw per scale in scale order
    v1 = kw.get("v1_multiscale", False)
    return [torch.randn(B, 2, H >> (s if v1 else 0), W >> (s if v1 else 0)) for s in range(4)]


def run_oracle(tag, kw, dtype=torch.float32):
    g, p = GOLD, tag + "_"
    opt = R.Opt(height=H, width=W, pose_model_type="posecnn", **kw)
    inputs = {k: v.to(dtype) for k, v in R.synthetic_inputs(B, H, W, seed=0).items()}
    disp = [T(g[p + "disp%d" % s]).to(dtype).requires_grad_() for s in range(4)]
    aa = {f: T(g[p + "aa_%d" % f]).to(dtype).requires_grad_() for f in (-1, 1)}
    tr = {f: T(g[p + "tr_%d" % f]).to(dtype).requires_grad_() for f in (-1, 1)}
    outputs = {("disp", s): disp[s] for s in range(4)}
    for f in (-1, 1):
        outputs[("axisangle", 0, f)], outputs[("translation", 0, f)] = aa[f], tr[f]
        outputs[("cam_T_cam", 0, f)] = R.transformation_from_parameters(aa[f][:, 0], tr[f][:, 0], invert=(f < 0))
    R.generate_images_pred(inputs, outputs, opt)
    losses = R.compute_losses(inputs, outputs, opt, [n.to(dtype) for n in noise_for(kw)])
    grads = torch.autograd.grad(losses["loss"], disp + [aa[-1], aa[1], tr[-1], tr[1]])
    return outputs, losses, grads


@pytest.mark.parametrize("tag,kw", TAGS)
def test_posecnn_trainer_level(tag, kw):
    g, p = GOLD, tag + "_"
    outputs, losses, grads = run_oracle(tag, kw)
    close(losses["loss"], g[p + "loss"], rtol=2e-6)
    for s in range(4):
        close(losses["loss/%d" % s], g[p + "loss%d" % s], rtol=2e-6)
        close_frac(grads[s], g[p + "gdisp%d" % s], rtol=1e-3, atol=2e-8, bad=1e-2)
        sel = outputs["identity_selection/%d" % s].numpy().astype(np.uint8)
        want = np.unpackbits(g[p + "idsel%d" % s])[:sel.size].reshape(sel.shape)
        assert (sel != want).mean() < 1e-4
    for j, f in enumerate((-1, 1)):
        # pose gradients: sums of ill-conditioned SSIM derivatives over every pixel (see test_oracle_golden.test_trainer_level)
        assert rel_l2(grads[4 + j], g[p + "gaa_%d" % f]) < 2e-2
        assert rel_l2(grads[6 + j], g[p + "gtr_%d" % f]) < 2e-2
        # slot 1 of the PoseCNN outputs is not consumed (trainer.py:497-499 read [:, 0]): exactly zero gradient
        assert float(grads[4 + j][:, 1].abs().max()) == 0.0 and float(grads[6 + j][:, 1].abs().max()) == 0.0
    if tag == "cnn":
        for s in (0, 3):
            close(outputs[("depth", 0, s)], g[p + "depth%d" % s], rtol=1e-5)
            for f in (-1, 1):
                close(outputs[("sample", f, s)], g[p + "sample_%d_%d" % (f, s)], rtol=1e-4, atol=3e-6)   # (points near z = 0 project far out)
                close(outputs[("color", f, s)], g[p + "color_%d_%d" % (f, s)], atol=2e-4)


def test_posecnn_rescaling_is_not_a_no_op():
    """The fixture separates the two functions: without the mean-inverse-depth rescaling the loss is off by far more than 1e-3."""
    g = GOLD
    opt = R.Opt(height=H, width=W)       # separate_resnet: no rescaling
    inputs = R.synthetic_inputs(B, H, W, seed=0)
    outputs = {("disp", s): T(g["cnn_disp%d" % s]) for s in range(4)}
    for f in (-1, 1):
        outputs[("cam_T_cam", 0, f)] = R.transformation_from_parameters(T(g["cnn_aa_%d" % f])[:, 0], T(g["cnn_tr_%d" % f])[:, 0], invert=(f < 0))
    R.generate_images_pred(inputs, outputs, opt)
    loss = R.compute_losses(inputs, outputs, opt, noise_for({}))["loss"]
    assert abs(float(loss) - float(g["cnn_loss"])) / float(g["cnn_loss"]) > 3e-3
