"""The pose modes outside the default (pose_model_type shared / posecnn, pose_model_input all; reference trainer.py:84-109,
378-442, networks/pose_cnn.py) on the GPU against the reference-generated fixture tests/golden/pose_modes.npz."""
import os

import numpy as np
import pytest
import torch

import make_golden as MG
import make_golden_r5 as G5
from helpers import close, rel_l2
from test_pose_modes_oracle import GOLD, pose_cnn_shapes, pose_decoder_shapes, seeded

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("nf", [2, 3])
def test_pose_cnn_vs_reference(nf):
    import networks
    net = networks.PoseCNN(nf).to(DEV)
    st = seeded(pose_cnn_shapes(nf), 40 + nf)
    assert list(net.state_dict()) == list(st)                      # the reference's key order
    net.load_state_dict(st)
    g = torch.Generator().manual_seed(50 + nf)
    x = torch.rand(G5.B, 3 * nf, G5.H, G5.W, generator=g).to(DEV).requires_grad_()
    aa, tr = net(x)
    p = "cnn%d_" % nf
    close(aa, GOLD[p + "aa"], rtol=1e-4, atol=1e-7)
    close(tr, GOLD[p + "tr"], rtol=1e-4, atol=1e-7)
    pp = dict(net.named_parameters())
    names = list(pp)
    ca, ct = torch.from_numpy(GOLD[p + "cot_aa"]).to(DEV), torch.from_numpy(GOLD[p + "cot_tr"]).to(DEV)
    gr = torch.autograd.grad((aa * ca).sum() + (tr * ct).sum(), [x] + [pp[k] for k in names])
    assert rel_l2(MG.summ(gr[0]), GOLD[p + "gx"]) < 1e-4
    for j, k in enumerate(names):
        got = MG.t2n(gr[1 + j]) if gr[1 + j].numel() <= 4096 else MG.summ(gr[1 + j])
        assert rel_l2(got, GOLD[p + "g_" + k]) < 2e-4, k


@pytest.mark.parametrize("tag,ptype,pinput", [("shared_pairs", "shared", "pairs"), ("shared_all", "shared", "all"), ("cnn_pairs", "posecnn", "pairs"),
                                              ("cnn_all", "posecnn", "all"), ("resnet_all", "separate_resnet", "all")])
def test_trainer_predict_poses_modes_vs_reference(tag, ptype, pinput):
    """Trainer.predict_poses with the reference's seeded pose network: which frames / features go in, in which order, which
    poses are inverted (the pose ENCODER of separate_resnet is the fixture's linear stand-in: only the wiring is pinned)."""
    import trainer as T
    tr = T.Trainer(T.default_options(batch_size=G5.B, height=G5.H, width=G5.W, pose_model_type=ptype, pose_model_input=pinput), device=DEV)
    npf = 2 if pinput == "pairs" else 3
    assert tr.num_pose_frames == npf and ("pose_encoder" in tr.models) == (ptype == "separate_resnet")
    if ptype == "posecnn":
        st = seeded(pose_cnn_shapes(3 if pinput == "all" else 2), 60)
    else:
        nin, npred = (1, 2) if ptype == "separate_resnet" else (npf, npf - 1)
        st = seeded(pose_decoder_shapes(nin, npred), 60)
    tr.models["pose"].load_state_dict(st)
    if ptype == "separate_resnet":
        tr.models["pose_encoder"] = G5.StandInEncoder(3 * npf).to(DEV)
    inputs = {k: v.to(DEV) for k, v in G5.frames([0, -1, 1]).items()}
    feats = {f: [t.to(DEV) for t in fl] for f, fl in G5.shared_features([0, -1, 1]).items()} if ptype == "shared" else None
    with torch.no_grad():
        out = tr.predict_poses(inputs, feats)
    assert {k[2] for k in out} == {-1, 1}
    for k, v in out.items():
        close(v, GOLD["%s_%s_%d" % (tag, k[0], k[2])], rtol=2e-4, atol=1e-6, msg=str(k))
    tr.close()


@pytest.mark.parametrize("ptype,pinput", [("shared", "pairs"), ("shared", "all"), ("posecnn", "pairs"), ("posecnn", "all"), ("separate_resnet", "all")])
def test_training_steps_in_every_pose_mode(ptype, pinput):
    """Whole training steps (forward, fused loss, backward, Adam) in each mode: finite, every trainable tensor but the unused
    classifier receives a gradient, the loss moves, and no convolution falls back to the framework."""
    import trainer as T
    from depthcore.synthetic import synthetic_batch
    B, H, W = 2, 64, 128
    tr = T.Trainer(T.default_options(batch_size=B, height=H, width=W, pose_model_type=ptype, pose_model_input=pinput), device=DEV, seed=2)
    tr.set_train()
    batch = synthetic_batch(B, H, W, torch.device(DEV), seed=4)
    entered = []
    real = torch.nn.Conv2d.forward
    torch.nn.Conv2d.forward = lambda self, x: (entered.append(self), real(self, x))[1]
    try:
        losses = []
        for _ in range(3):
            _, l = tr.train_step(dict(batch))
            losses.append(float(l["loss"].detach()))
    finally:
        torch.nn.Conv2d.forward = real
    assert entered == [], entered[:3]
    assert all(np.isfinite(losses)) and losses[0] != losses[-1]
    missing = [k + "." + n for k, m in tr.models.items() for n, p in m.named_parameters() if p.grad is None and ".fc." not in n]
    assert not missing, missing[:5]
    tr.close()
