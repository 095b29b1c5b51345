"""oracle/ref_cpu.py against the reference-generated fixture tests/golden/pose_modes.npz (tests/golden/make_golden_r5.py):
PoseCNN (reference networks/pose_cnn.py) and `predict_poses` in every pose_model_type / pose_model_input (trainer.py:378-442)."""
import os

import numpy as np
import pytest
import torch

from oracle import ref_cpu as R
import make_golden as MG
import make_golden_r5 as G5
from helpers import close, rel_l2

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pose_modes.npz"))


def seeded(shapes, seed, scale=0.05):
    """make_golden.seeded_state restated on a {name: shape} dict, in state_dict order."""
    g = torch.Generator().manual_seed(seed)
    return {k: scale * torch.randn(shp, generator=g) for k, shp in shapes.items()}


def pose_cnn_shapes(nf):
    ch = [3 * nf, 16, 32, 64, 128, 256, 256, 256]
    ks = [7, 5, 3, 3, 3, 3, 3]
    sh = {}
    # state_dict order of the reference module: pose_conv first (registered before `net`), then net.0 .. net.6
    sh["pose_conv.weight"] = (6 * (nf - 1), 256, 1, 1)
    sh["pose_conv.bias"] = (6 * (nf - 1),)
    for i in range(7):
        sh["net.%d.weight" % i] = (ch[i + 1], ch[i], ks[i], ks[i])
        sh["net.%d.bias" % i] = (ch[i + 1],)
    return sh


def pose_decoder_shapes(nin, npred):
    return {"net.0.weight": (256, 512, 1, 1), "net.0.bias": (256,), "net.1.weight": (256, nin * 256, 3, 3), "net.1.bias": (256,),
            "net.2.weight": (256, 256, 3, 3), "net.2.bias": (256,), "net.3.weight": (6 * npred, 256, 1, 1), "net.3.bias": (6 * npred,)}


@pytest.mark.parametrize("nf", [2, 3])
def test_pose_cnn_oracle_vs_reference(nf):
    st = {k: v.requires_grad_() for k, v in seeded(pose_cnn_shapes(nf), 40 + nf).items()}
    g = torch.Generator().manual_seed(50 + nf)
    x = torch.rand(G5.B, 3 * nf, G5.H, G5.W, generator=g).requires_grad_()
    aa, tr = R.pose_cnn_forward(st, x, nf)
    p = "cnn%d_" % nf
    close(aa, GOLD[p + "aa"], rtol=1e-5, atol=1e-8)
    close(tr, GOLD[p + "tr"], rtol=1e-5, atol=1e-8)
    names = list(st)
    gr = torch.autograd.grad((aa * torch.from_numpy(GOLD[p + "cot_aa"])).sum() + (tr * torch.from_numpy(GOLD[p + "cot_tr"])).sum(), [x] + [st[k] for k in names])
    close(MG.summ(gr[0]), GOLD[p + "gx"], rtol=1e-4, atol=1e-9)
    for j, k in enumerate(names):
        want = GOLD[p + "g_" + k]
        got = MG.t2n(gr[1 + j]) if gr[1 + j].numel() <= 4096 else MG.summ(gr[1 + j])
        assert rel_l2(got, want) < 1e-5, k


@pytest.mark.parametrize("tag,ptype,pinput", [("shared_pairs", "shared", "pairs"), ("shared_all", "shared", "all"), ("cnn_pairs", "posecnn", "pairs"),
                                              ("cnn_all", "posecnn", "all"), ("resnet_all", "separate_resnet", "all")])
def test_predict_poses_modes_vs_reference(tag, ptype, pinput):
    frame_ids = [0, -1, 1]
    npf = 2 if pinput == "pairs" else 3
    inputs = G5.frames(frame_ids)
    feats = G5.shared_features(frame_ids) if ptype == "shared" else None
    enc = G5.StandInEncoder(3 * npf) if ptype == "separate_resnet" else None
    if ptype == "posecnn":
        nf = 3 if pinput == "all" else 2
        st = seeded(pose_cnn_shapes(nf), 60)
        pose = lambda t: R.pose_cnn_forward(st, t, nf)              # noqa: E731
    else:
        nin, npred = (1, 2) if ptype == "separate_resnet" else (npf, npf - 1)
        st = seeded(pose_decoder_shapes(nin, npred), 60)
        pose = lambda f: R.pose_decoder_forward(st, f, npred)       # noqa: E731
    with torch.no_grad():
        out = R.predict_poses_modes(inputs, feats, ptype, pinput, frame_ids, enc, pose)
    assert {k[2] for k in out} == {-1, 1}
    for k, v in out.items():
        close(v, GOLD["%s_%s_%d" % (tag, k[0], k[2])], rtol=1e-5, atol=1e-7, msg=str(k))
