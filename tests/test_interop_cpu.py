"""f3: option / checkpoint compatibility with the reference (options.py:15-287, trainer.py:700-763) -- no GPU needed.

The expected names, defaults and file layout below are data read off the reference's options.py / trainer.py."""
import json
import os

import numpy as np
import pytest
import torch

REF_DEFAULTS = {          # options.py: name -> default (the fields a hot-path run reads, plus a sample of the others)
    "model": "dpt_gru", "model_name": "mdp", "split": "eigen_zhou", "num_layers": 18, "len_sequence": 10,
    "gru_version": "v5", "dataset": "kitti", "height": 192, "width": 640, "disparity_smoothness": 1e-3,
    "scales": [0, 1, 2, 3], "min_depth": 0.1, "max_depth": 100.0, "frame_ids": [0, -1, 1], "batch_size": 12,
    "learning_rate": 1e-4, "num_epochs": 20, "scheduler_step_size": 15, "weights_init": "pretrained",
    "pose_model_input": "pairs", "pose_model_type": "separate_resnet", "num_workers": 12, "load_weights_folder": None,
    "models_to_load": ["pose_encoder", "pose", "encoder", "depth", "gru", "head"], "log_frequency": 250,
    "save_frequency": 1, "eval_split": "eigen", "pred_depth_scale_factor": 1, "disable_attention": False,
    "v1_multiscale": False, "avg_reprojection": False, "disable_automasking": False, "predictive_mask": False,
    "no_ssim": False, "use_stereo": False, "png": False, "no_cuda": False, "post_process": False,
}
REF_NAMES = 61            # `--name` add_argument calls in options.py (besides "-f")


def test_options_match_reference_defaults():
    from options import MonodepthOptions, reference_option_names
    opt = MonodepthOptions().parse([])
    assert len(reference_option_names()) == REF_NAMES
    for k, v in REF_DEFAULTS.items():
        assert getattr(opt, k) == v, k
    o2 = MonodepthOptions().parse("--num_layers 50 --height 320 --width 1024 --batch_size 8 --frame_ids 0 -2 -1 1 "
                                  "--disable_automasking --predictive_mask --scales 0 1 --weights_init scratch "
                                  "--models_to_load encoder depth".split())
    assert (o2.num_layers, o2.height, o2.width, o2.batch_size) == (50, 320, 1024, 8)
    assert o2.frame_ids == [0, -2, -1, 1] and o2.scales == [0, 1] and o2.predictive_mask and o2.disable_automasking
    assert o2.models_to_load == ["encoder", "depth"]
    with pytest.raises(SystemExit):
        MonodepthOptions().parse(["--num_layers", "19"])


def test_opt_json_written_by_the_reference_loads(tmp_path):
    """trainer.py:700-709 dumps `self.opt.__dict__`; such a file (here: reference option names only, no build knobs)
    must load, and this build's own opt.json must contain every reference option."""
    from options import MonodepthOptions, options_from_json, reference_option_names
    ref_like = {k: v for k, v in vars(MonodepthOptions().parse([])).items() if k in reference_option_names()}
    ref_like.update(num_layers=50, height=320, width=1024, batch_size=8, frame_ids=[0, -1, 1])
    p = tmp_path / "opt.json"
    p.write_text(json.dumps(ref_like, indent=2))
    opt = options_from_json(str(p))
    assert (opt.num_layers, opt.height, opt.width, opt.batch_size) == (50, 320, 1024, 8)
    assert opt.fused_loss is True and opt.fusion is None           # build knobs keep their defaults


def _reference_shaped_checkpoint(folder, trainer, epoch):
    """Files as the reference's save_model writes them (trainer.py:711-729)."""
    d = os.path.join(folder, "weights_{}".format(epoch))
    os.makedirs(d)
    g = torch.Generator().manual_seed(5)
    want = {}
    for name, m in trainer.models.items():
        sd = {k: (0.1 * torch.randn(v.shape, generator=g) if v.is_floating_point() else v.clone()) for k, v in m.state_dict().items()}
        want[name] = {k: v.clone() for k, v in sd.items()}
        if name == "encoder":
            sd["height"], sd["width"], sd["use_stereo"] = 192, 640, False
        torch.save(sd, os.path.join(d, name + ".pth"))
    return d, want


def test_checkpoint_layout_interchanges_with_the_reference(tmp_path):
    import trainer as T
    from oracle import ref_cpu as R
    opt = T.default_options(batch_size=1, height=64, width=96, log_dir=str(tmp_path), model_name="run")
    a = T.Trainer(opt, device="cpu", seed=1)
    src, want = _reference_shaped_checkpoint(str(tmp_path / "theirs"), a, 7)
    # a reference-shaped folder loads through the reference's own option names (default models_to_load names gru / head too)
    b = T.Trainer(T.default_options(batch_size=1, height=64, width=96, load_weights_folder=src, log_dir=str(tmp_path),
                                    model_name="run"), device="cpu", seed=2)
    for name, sd in want.items():
        got = b.models[name].state_dict()
        assert set(got) == set(sd)                       # height / width / use_stereo were dropped
        for k in sd:
            assert torch.equal(got[k], sd[k]), (name, k)
    # and what this build writes has the reference's layout
    b.epoch = 3
    out = b.save_model()
    assert out == os.path.join(str(tmp_path), "run", "models", "weights_3")      # <log_dir>/<model_name>/models/weights_{epoch}
    files = sorted(os.listdir(out))
    assert files == ["adam.pth", "depth.pth", "encoder.pth", "pose.pth", "pose_encoder.pth"]
    assert os.path.isfile(os.path.join(os.path.dirname(out), "opt.json"))          # <log_path>/models/opt.json
    enc = torch.load(os.path.join(out, "encoder.pth"))
    assert enc["height"] == 64 and enc["width"] == 96 and enc["use_stereo"] is False
    dec = torch.load(os.path.join(out, "depth.pth"))
    lay = R.depth_decoder_layout(np.array([64, 64, 128, 256, 512]))
    assert len(dec) == 2 * len(lay) and "decoder.0.conv.conv.weight" in dec and "decoder.13.conv.bias" in dec
    pose = torch.load(os.path.join(out, "pose.pth"))
    assert list(pose) == ["net.%d.%s" % (i, w) for i in range(4) for w in ("weight", "bias")]
    adam = torch.load(os.path.join(out, "adam.pth"))
    assert set(adam) == {"state", "param_groups"} and len(adam["param_groups"]) == 1        # one Adam group, trainer.py:127


def test_imagenet_weights_are_tiled_and_divided_for_stacked_frames():
    """networks/resnet_encoder.py:52-57 with a stand-in for the torchvision file (same keys and shapes)."""
    import networks
    from networks.resnet_encoder import ResNetTrunk
    torch.manual_seed(0)
    tv = ResNetTrunk(18, 1).state_dict()                      # torchvision resnet18 layout: conv1.weight (64,3,7,7), ..., fc.*
    tv = {k: torch.randn_like(v) if v.is_floating_point() else v for k, v in tv.items()}
    pose_enc = networks.ResnetEncoder(18, tv, num_input_images=2)
    w = pose_enc.encoder.conv1.weight
    assert w.shape == (64, 6, 7, 7)
    assert torch.equal(w[:, :3], tv["conv1.weight"] / 2) and torch.equal(w[:, 3:], tv["conv1.weight"] / 2)
    assert torch.equal(pose_enc.encoder.layer3[1].conv2.weight, tv["layer3.1.conv2.weight"])
    depth_enc = networks.ResnetEncoder(18, tv)
    assert torch.equal(depth_enc.encoder.conv1.weight, tv["conv1.weight"])
    with pytest.raises(RuntimeError):
        networks.ResnetEncoder(18, True)


def test_pretrained_init_is_decided_per_encoder_when_fine_tuning(tmp_path):
    """weights_init="pretrained" (the reference's CLI default) without `imagenet_weights`: an encoder may start from a scratch
    init only if load_model() is going to overwrite it.  A fine-tune run that loads encoder + depth must NOT silently leave the
    pose encoder at a random init (the reference gives it ImageNet weights, networks/resnet_encoder.py:52-57): it is refused."""
    import trainer as T
    a = T.Trainer(T.default_options(batch_size=1, height=64, width=96), device="cpu", seed=1)
    src, want = _reference_shaped_checkpoint(str(tmp_path / "theirs"), a, 7)
    kw = dict(batch_size=1, height=64, width=96, load_weights_folder=src, weights_init="pretrained")
    # every model is in the folder and in models_to_load: scratch stand-ins are fine, load_model() overwrites them
    b = T.Trainer(T.default_options(**kw), device="cpu", seed=2)
    assert torch.equal(b.models["pose_encoder"].state_dict()["encoder.conv1.weight"], want["pose_encoder"]["encoder.conv1.weight"])
    # encoder + depth only: the pose encoder would keep a random init -> refused unless the ImageNet file is given
    with pytest.raises(RuntimeError):
        T.Trainer(T.default_options(models_to_load=["encoder", "depth"], **kw), device="cpu", seed=2)
    # a listed model whose file is missing is not a stand-in either
    os.remove(os.path.join(src, "pose_encoder.pth"))
    with pytest.raises(RuntimeError):
        T.Trainer(T.default_options(**kw), device="cpu", seed=2)
    from networks.resnet_encoder import ResNetTrunk
    torch.manual_seed(0)
    tv = str(tmp_path / "resnet18-stand-in.pth")
    torch.save(ResNetTrunk(18, 1).state_dict(), tv)
    c = T.Trainer(T.default_options(models_to_load=["encoder", "depth"], imagenet_weights=tv, **kw), device="cpu", seed=2)
    w = torch.load(tv)["conv1.weight"]
    assert torch.equal(c.models["pose_encoder"].encoder.conv1.weight[:, :3], w / 2)      # tiled and divided: ImageNet init
    assert torch.equal(c.models["encoder"].state_dict()["encoder.conv1.weight"], want["encoder"]["encoder.conv1.weight"])
