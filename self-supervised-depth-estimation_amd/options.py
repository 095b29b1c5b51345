"""Command-line / opt.json compatibility with the reference's options.py:15-287 (`MonodepthOptions`).

Same option names, types, defaults and choices, so that command lines and `opt.json` files written by either side load in
the other (`Trainer.save_opts` / `options_from_json`).  Options that steer parts of the reference outside the hot path
(dataset paths, logging, evaluation, per-model GPU ids) are parsed and carried along unchanged; the trainer reads only the
hot-path ones (trainer.py: height, width, scales, frame_ids, batch_size, ...).  A second group holds the build's own knobs.
"""
import argparse
import json
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_T = True      # store_true flag

# (name, type | _T, default, choices)            -- order and grouping of reference options.py
_REFERENCE = [
    # PATHS
    ("data_path", str, os.path.join(_HERE, "kitti_data"), None),
    ("log_dir", str, os.path.join(os.path.expanduser("~"), "tmp"), None),
    # TRAINING
    ("model", str, "dpt_gru", None),
    ("model_name", str, "mdp", None),
    ("split", str, "eigen_zhou", ["eigen_zhou", "eigen_full", "odom", "benchmark"]),
    ("disable_attention", _T, False, None),
    ("num_layers", int, 18, [18, 34, 50, 101, 152]),
    ("len_sequence", int, 10, None),
    ("train_n_tuples", int, 60, None),
    ("test_n_tuples", int, 10, None),
    ("pose_mask", int, 1, None),
    ("mono_pretrained", int, 1, None),
    ("gru_pre_disp", int, 1, None),
    ("h_s_epoch", int, 10, None),
    ("gru_version", str, "v5", None),
    ("fuse", int, 1, None),
    ("dataset", str, "kitti", ["kitti", "kitti_odom", "kitti_depth", "kitti_test"]),
    ("png", _T, False, None),
    ("height", int, 192, None),
    ("width", int, 640, None),
    ("disparity_smoothness", float, 1e-3, None),
    ("scales", [int], [0, 1, 2, 3], None),
    ("min_depth", float, 0.1, None),
    ("max_depth", float, 100.0, None),
    ("use_stereo", _T, False, None),
    ("frame_ids", [int], [0, -1, 1], None),
    ("depth_encoder_gpu_id", [int], 0, None),
    ("depth_decoder_gpu_id", [int], 0, None),
    ("pose_encoder_gpu_id", [int], 0, None),
    ("pose_decoder_gpu_id", [int], 0, None),
    ("gru_gpu_id", [int], 0, None),
    ("main_gpu_id", [int], 0, None),
    # OPTIMIZATION
    ("batch_size", int, 12, None),
    ("learning_rate", float, 1e-4, None),
    ("num_epochs", int, 20, None),
    ("scheduler_step_size", int, 15, None),
    # ABLATION
    ("v1_multiscale", _T, False, None),
    ("avg_reprojection", _T, False, None),
    ("disable_automasking", _T, False, None),
    ("predictive_mask", _T, False, None),
    ("no_ssim", _T, False, None),
    ("weights_init", str, "pretrained", ["pretrained", "scratch"]),
    ("pose_model_input", str, "pairs", ["pairs", "all"]),
    ("pose_model_type", str, "separate_resnet", ["posecnn", "separate_resnet", "shared"]),
    # SYSTEM
    ("no_cuda", _T, False, None),
    ("num_workers", int, 12, None),
    # LOADING
    ("load_weights_folder", str, None, None),
    ("models_to_load", [str], ["pose_encoder", "pose", "encoder", "depth", "gru", "head"], None),
    # LOGGING
    ("log_frequency", int, 250, None),
    ("save_frequency", int, 1, None),
    # EVALUATION
    ("eval_stereo", _T, False, None),
    ("eval_mono", _T, False, None),
    ("disable_median_scaling", _T, False, None),
    ("pred_depth_scale_factor", float, 1, None),
    ("ext_disp_to_eval", str, None, None),
    ("eval_split", str, "eigen", ["eigen", "eigen_benchmark", "benchmark", "odom_9", "odom_10"]),
    ("save_pred_disps", _T, False, None),
    ("no_eval", _T, False, None),
    ("eval_eigen_to_benchmark", _T, False, None),
    ("eval_out_dir", str, None, None),
    ("post_process", _T, False, None),
]

# knobs of this build (absent from the reference; ignored by it when found in an opt.json)
_BUILD = [
    ("fused_loss", int, 1, [0, 1]),              # 1: fused photometric kernels, 0: layer-by-layer kernels
    ("overlap_streams", int, 1, [0, 1]),         # pose and depth networks on two HIP streams
    ("step_priority", int, 2, [-1, 0, 2]),       # Trainer.on_step_stream(): the training loop (depth branch) on a HIGH-priority stream: -1 on, 0 off, 2 = where it was measured to pay (Trainer)
    ("wgrad_lanes", int, 2, [0, 1, 2]),          # weight-gradient kernels on companion streams of the backward's streams (ops.WgradLanes): 0 off, 1 on, 2 = on for GPU-bound step sizes (Trainer)
    ("bucket_mb", int, 32, None),                # gradient bucket size for the RCCL exchange
    ("cpu_tiebreak_noise", int, 0, [0, 1]),      # 1: the reference's CPU randn + H2D copy (trainer.py:594-595)
    ("materialize_logs", int, 0, [0, 1]),        # 1: fused path also writes depth / sample / color tensors
    ("fusion", str, None, [None, "v3"]),         # front-end of trainer_fusion_v3.py
    ("hip_graph", int, 0, [0, 1]),               # single-GPU: capture the whole step (fwd + bwd + Adam) in one hipGraph and replay it
    ("wino_weight_cache", int, 1, [0, 1]),       # all 3x3 weights -> Winograd domain once per step (one launch)
    ("torch_adam", int, 0, [0, 1]),              # 1: torch.optim.Adam(fused=True) instead of the depthcore Adam kernel (A/B)
    ("nets_dtype", str, "f32", ["f32", "bf16"]),  # matrix-core precision of the networks' convolutions (bf16: BASELINE configs[4] policy; the loss stays fp32)
    ("imagenet_weights", str, None, None),       # weights_init=pretrained: path of the torchvision resnet{N}-*.pth (no download here)
    ("gru", str, None, [None, "v5"]),            # front-end of trainer_gru.py (run_gru_v5; sequences of len_sequence frames)
]


class MonodepthOptions:
    def __init__(self):
        self.parser = argparse.ArgumentParser(description="Monodepthv2 options")
        self.parser.add_argument("-f")                                   # (notebook kernels pass -f; options.py:19)
        for name, typ, default, choices in _REFERENCE + _BUILD:
            kw = {}
            if typ is _T:
                kw = dict(action="store_true")
            elif isinstance(typ, list):
                kw = dict(nargs="+", type=typ[0], default=default)
            else:
                kw = dict(type=typ, default=default)
                if choices:
                    kw["choices"] = choices
            self.parser.add_argument("--" + name, **kw)

    def parse(self, argv=None):
        self.options = self.parser.parse_args(argv)
        for name, typ, _, _ in _BUILD:
            if typ is int and getattr(self.options, name) in (0, 1) and name != "bucket_mb":
                setattr(self.options, name, bool(getattr(self.options, name)))
        return self.options


def reference_option_names():
    return [n for n, _, _, _ in _REFERENCE]


def options_from_json(path, **overrides):
    """An opt.json written by the reference's `save_opts` (trainer.py:700-709) or by this build -> options namespace;
    options the file lacks keep their defaults."""
    opt = MonodepthOptions().parse([])
    with open(path) as f:
        for k, v in json.load(f).items():
            setattr(opt, k, v)
    for k, v in overrides.items():
        setattr(opt, k, v)
    return opt
