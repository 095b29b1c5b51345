"""`pose_model_type == "posecnn"` through the Trainer's loss paths on the GPU (reference trainer.py:490-499: the translation of
every (scale, frame) is rescaled by that scale's mean inverse depth) against the reference-generated trainer-level fixture
tests/golden/trainer_posecnn.npz: the layer-by-layer path (generate_images_pred + compute_losses), the fused kernels with
per-scale poses (dc_photo_desc.T_scale / d_T_scale), and `v1_multiscale` on both."""
import numpy as np
import pytest
import torch

from oracle import ref_cpu as R
import make_golden as MG
from helpers import T, close, close_frac, rel_l2
from test_posecnn_oracle import GOLD, TAGS, run_oracle

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")
B, H, W = MG.B, MG.H, MG.W


def run_hip(tag, kw, fused):
    import trainer as TR
    g, p = GOLD, tag + "_"
    tr = TR.Trainer(TR.default_options(batch_size=B, height=H, width=W, cpu_tiebreak_noise=True, pose_model_type="posecnn",
                                       materialize_logs=True, **kw), device=DEV, seed=0)
    inputs = {k: v.to(DEV) for k, v in R.synthetic_inputs(B, H, W, seed=0).items()}
    disp = [T(g[p + "disp%d" % s]).to(DEV).requires_grad_() for s in range(4)]
    pose = [T(g[p + k]).to(DEV).requires_grad_() for k in ("aa_-1", "aa_1", "tr_-1", "tr_1")]
    outputs = {("disp", s): disp[s] for s in range(4)}
    from layers import transformation_from_parameters
    for j, f in enumerate((-1, 1)):
        outputs[("axisangle", 0, f)], outputs[("translation", 0, f)] = pose[j], pose[2 + j]
        outputs[("cam_T_cam", 0, f)] = transformation_from_parameters(pose[j][:, 0], pose[2 + j][:, 0], invert=(f < 0))
    torch.manual_seed(1234)
    if fused:
        losses = (tr.fused_losses_v1 if kw.get("v1_multiscale") else tr.fused_losses)(inputs, outputs)
    else:
        tr.generate_images_pred(inputs, outputs)
        losses = tr.compute_losses(inputs, outputs)
    grads = torch.autograd.grad(losses["loss"], disp + pose)
    torch.cuda.synchronize()
    return outputs, losses, grads


@pytest.mark.parametrize("fused", [False, True], ids=["layers", "fused"])
@pytest.mark.parametrize("tag,kw", TAGS)
def test_posecnn_trainer_level_golden(tag, kw, fused):
    g, p = GOLD, tag + "_"
    outputs, losses, grads = run_hip(tag, kw, fused)
    # tolerance: north_star "within 1e-3 rel fp32"
    close(losses["loss"], g[p + "loss"], rtol=1e-3, atol=0)
    _, _, g64 = run_oracle(tag, kw, torch.float64)
    _, _, g32 = run_oracle(tag, kw, torch.float32)
    report = []
    for s in range(4):
        close(losses["loss/%d" % s], g[p + "loss%d" % s], rtol=1e-3, atol=0)
        sel = outputs["identity_selection/%d" % s].cpu().numpy().astype(np.uint8) if not fused else \
            (outputs[("argmin", s)] >= 2).cpu().numpy().astype(np.uint8)
        want = np.unpackbits(g[p + "idsel%d" % s])[:sel.size].reshape(sel.shape)
        assert (sel != want).mean() < 2e-3
    # gradients (disparities incl. the path through the mean inverse depth, axis-angles, translations): calibrated like the other
    # ablations -- at most 3x as far from the fp64 oracle as the oracle's own fp32 evaluation (floor 2e-4) -- and against the
    # reference's fp32 numbers within twice the fixture's own distance from fp64 (+ 1e-3)
    names = ["gdisp%d" % s for s in range(4)] + ["gaa_-1", "gaa_1", "gtr_-1", "gtr_1"]
    for i, name in enumerate(names):
        e_hip, e_32 = rel_l2(grads[i], g64[i]), rel_l2(g32[i], g64[i])
        report.append((name, e_hip, e_32))
        assert e_hip <= 3.0 * e_32 + 2e-4, report
        allowed = 2.0 * rel_l2(g[p + name], g64[i]) + 1e-3
        assert rel_l2(grads[i], g[p + name]) <= allowed, (name, rel_l2(grads[i], g[p + name]), allowed)
    for i in range(4, 8):                 # slot 1 of PoseCNN's outputs is never read (trainer.py:497-499)
        assert float(grads[i][:, 1].abs().max()) == 0.0
    print("posecnn %s %s gradients (leaf, |hip-f64|/|f64|, |f32-f64|/|f64|):" % (tag, "fused" if fused else "layers"), report)
    if tag == "cnn":
        for s in (0, 3):
            close(outputs[("depth", 0, s)], g[p + "depth%d" % s], rtol=1e-4)
            for f in (-1, 1):
                close_frac(outputs[("sample", f, s)], g[p + "sample_%d_%d" % (f, s)], rtol=1e-3, atol=1e-5, bad=1e-3)
                close_frac(outputs[("color", f, s)], g[p + "color_%d_%d" % (f, s)], rtol=1e-3, atol=1e-4, bad=1e-3)


def test_posecnn_whole_step_uses_the_rescaled_pose():
    """A whole training step in posecnn mode: the fused loss equals the layer-by-layer loss of the same step (both rescale), and
    differs from the loss the un-rescaled cam_T_cam would give."""
    import trainer as TR
    from depthcore.synthetic import synthetic_batch
    o = TR.default_options(batch_size=B, height=H, width=W, cpu_tiebreak_noise=True, pose_model_type="posecnn")
    tr = TR.Trainer(o, device=DEV, seed=0)
    tr.set_train()
    batch = synthetic_batch(B, H, W, DEV)
    torch.manual_seed(7)
    outputs, losses = tr.process_batch(dict(batch))
    ref_out = dict(outputs)
    torch.manual_seed(7)
    tr.generate_images_pred(batch, ref_out)
    ref = tr.compute_losses(batch, ref_out)
    close(losses["loss"], ref["loss"], rtol=1e-4, atol=0)
    tr.buckets.zero()
    losses["loss"].backward()
    tr.buckets.finish()                    # (gradients are written into the flat buckets' slices; finish() binds p.grad)
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in tr.models["pose"].parameters())
