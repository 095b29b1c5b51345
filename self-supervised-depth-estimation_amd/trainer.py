"""The per-batch hot path of the reference's Trainer (reference trainer.py:256-622), MI355X-native.

Same method names and dict schemas as the reference so that code written against
`Trainer.process_batch / predict_poses / generate_images_pred / compute_reprojection_loss /
compute_losses` keeps working; the control plane around it (data loading, tensorboard, checkpoints,
validation) is out of scope (SURVEY 8, rows a14-a18 only).

Two equivalent loss paths, both on hand-written gfx950 kernels:
  * fused (default): generate_images_pred + compute_losses are ONE forward op and ONE backward op
    (`depthcore.ops.photometric_loss`); log tensors are produced only when asked for;
  * layer-by-layer (`opt.fused_loss = False`): the reference's own sequence of `layers.*` calls.
The a15 ablations run on the fused kernels too: `predictive_mask` (trainer.py:571-590: the masks multiply the reprojection
losses inside the kernels and get their gradient there; the BCE term stays a torch expression) and `v1_multiscale`
(trainer.py:471,541: one single-scale fused call per scale, `fused_losses_v1`).

`opt.fusion = "v3"` switches the front-end to the reference's trainer_fusion_v3.py:277-330: frames [-2, -1, 0] are stacked
through the depth encoder + decoder and fused by `networks.Fusion_v3` (BASELINE configs[4]).
`opt.gru = "v5"` switches it to trainer_gru.py:595-644 (`run_gru_v5`, BASELINE configs[3]): a batch is ONE sequence of
`opt.len_sequence` frames (dict keys carry the sequence index: ("color", f, s, j), ("K", s, j)), the encoder features of the
sequence pass through `networks.ConvGRUBlocks_v5`, and the loss runs on the sequence stacked along the batch.
"""
import contextlib
import json
import os
import types

import torch
import torch.nn.functional as F
import torch.optim as optim

import networks
from layers import (BackprojectDepth, Project3D, SSIM, disp_to_depth, get_smooth_loss, grid_sample,
                    interpolate_bilinear, transformation_from_parameters)
from depthcore import ops
from depthcore.ops import WinoWeightCache
from depthcore.ddp import GradBuckets, agree_all, broadcast_parameters


def default_options(**kw):
    """The reference's option set with its defaults (options.py, via options.MonodepthOptions) plus this build's knobs;
    `weights_init` defaults to "scratch" here (no network: the ImageNet download of "pretrained" is unavailable -- a
    torchvision checkpoint can be given to ResnetEncoder(pretrained=<path or state_dict>))."""
    from options import MonodepthOptions
    o = MonodepthOptions().parse([])
    o.weights_init = "scratch"
    for k, v in kw.items():
        setattr(o, k, v)
    return o


class _NoWinoCache:
    """wino_weight_cache = 0 (or a CPU trainer used for bookkeeping): every convolution transforms its own weights."""

    def refresh(self):
        pass

    invalidate = close = refresh

    def variants(self):
        return 0


class Trainer:
    def __init__(self, options, device="cuda:0", rank=0, world_size=1, process_group=None, seed=0):
        self.opt = options
        self.device = torch.device(device)
        if self.device.type == "cuda":
            # the dc_* kernels launch on the calling thread's current HIP device (depthcore._lib.stream refuses tensors
            # of another one): the reference pins each model to a fixed cuda:N as well (trainer.py:43-50)
            torch.cuda.set_device(self.device)
        self.rank, self.world_size = rank, world_size
        assert self.opt.height % 32 == 0 and self.opt.width % 32 == 0      # trainer.py:37-38
        assert self.opt.frame_ids[0] == 0
        if self.opt.pose_model_type not in ("separate_resnet", "shared", "posecnn") or self.opt.pose_model_input not in ("pairs", "all"):
            raise ValueError("pose_model_type: separate_resnet | shared | posecnn; pose_model_input: pairs | all (options.py:63-64)")
        if self.opt.pose_model_type == "shared" and (getattr(self.opt, "fusion", None) or getattr(self.opt, "gru", None)):
            raise ValueError("the sequence front-ends (trainer_fusion_v3.py, trainer_gru.py) run their own encoder pass; "
                             "pose_model_type='shared' belongs to the vanilla trainer (trainer.py:263-279)")
        if self.opt.predictive_mask and not self.opt.disable_automasking:       # trainer.py:116-117
            raise ValueError("When using predictive_mask, please disable automasking with disable_automasking")
        if getattr(self.opt, "fusion", None) not in (None, "v3"):
            raise NotImplementedError("fusion front-ends: None (trainer.py) or 'v3' (trainer_fusion_v3.py)")
        if self.opt.fusion and -2 not in self.opt.frame_ids:
            raise ValueError("the Fusion_v3 front-end stacks frames [-2, -1, 0]: frame_ids must be [0, -2, -1, 1]")
        if getattr(self.opt, "gru", None) not in (None, "v5"):
            raise NotImplementedError("ConvGRU front-ends: None or 'v5' (trainer_gru.py run_gru_v5)")
        if self.opt.gru and self.opt.fusion:
            raise ValueError("fusion and gru front-ends are separate trainers in the reference; pick one")
        if self.opt.gru and self.opt.batch_size != 1:
            raise ValueError("run_gru_v5 works for batch_size = 1 only (trainer_gru.py:596)")
        # images per loss evaluation: the reference's GRU trainer stacks the sequence along the batch (trainer_gru.py:256-263)
        self.loss_batch = self.opt.batch_size * (self.opt.len_sequence if self.opt.gru else 1)
        self.num_scales = len(self.opt.scales)
        self.num_input_frames = len(self.opt.frame_ids)                                   # trainer.py:50-51
        self.num_pose_frames = 2 if self.opt.pose_model_input == "pairs" else self.num_input_frames
        if self.opt.pose_model_input == "all" and self.opt.pose_model_type == "separate_resnet" and self.num_input_frames != 3:
            raise ValueError("pose_model_input='all' with separate_resnet predicts two poses (trainer.py:100-103): frame_ids must be "
                             "[0, a, b]")

        torch.manual_seed(seed)                                  # same initial weights on every rank
        self.models = {}
        # weights_init = "pretrained" (the reference's CLI default, also found in every opt.json it writes) means the
        # torchvision ImageNet download there (networks/resnet_encoder.py:53).  No network here: `opt.imagenet_weights`
        # (path of a torchvision resnet{N}-*.pth) supplies them.  Without it an encoder may start from a scratch init ONLY
        # if load_model() is going to overwrite it -- its name is in `models_to_load` AND its .pth exists in
        # `load_weights_folder`; decided per encoder (a fine-tune run that loads encoder + depth only must still give the
        # pose encoder its ImageNet weights, as the reference does); otherwise ResnetEncoder refuses (pretrained=True).
        def _init_for(name):
            if self.opt.weights_init != "pretrained":
                return False
            iw = getattr(self.opt, "imagenet_weights", None)
            if iw:
                return iw
            folder = getattr(self.opt, "load_weights_folder", None)
            if folder and name in (self.opt.models_to_load or []) and \
                    os.path.isfile(os.path.join(os.path.expanduser(folder), "{}.pth".format(name))):
                return False                                     # overwritten by load_model() below
            return True                                          # -> ResnetEncoder raises: no download, no silent scratch
        self.models["encoder"] = networks.ResnetEncoder(self.opt.num_layers, _init_for("encoder"))
        self.models["depth"] = networks.DepthDecoder(self.models["encoder"].num_ch_enc, self.opt.scales)
        if self.opt.fusion:                                      # trainer_fusion_v3.py:74
            self.models["fusion"] = networks.Fusion_v3(attention=not self.opt.disable_attention)
        if self.opt.gru:                                         # trainer_gru.py:128
            self.models["gru"] = networks.ConvGRUBlocks_v5(kernel_size=(3, 3), bias=True, device="cpu", height=self.opt.height,
                                                           width=self.opt.width,
                                                           num_ch_enc=tuple(int(c) for c in self.models["encoder"].num_ch_enc))
        if self.opt.pose_model_type == "separate_resnet":          # trainer.py:84-103
            self.models["pose_encoder"] = networks.ResnetEncoder(self.opt.num_layers, _init_for("pose_encoder"),
                                                                 num_input_images=self.num_pose_frames)
            self.models["pose"] = networks.PoseDecoder(self.models["pose_encoder"].num_ch_enc, num_input_features=1,
                                                       num_frames_to_predict_for=2)
        elif self.opt.pose_model_type == "shared":                 # trainer.py:104-106: the depth encoder's features of every frame
            self.models["pose"] = networks.PoseDecoder(self.models["encoder"].num_ch_enc, self.num_pose_frames)
        else:                                                      # trainer.py:107-109
            self.models["pose"] = networks.PoseCNN(self.num_input_frames if self.opt.pose_model_input == "all" else 2)
        if self.opt.predictive_mask:                             # trainer.py:115-125: one mask per source frame
            self.models["predictive_mask"] = networks.DepthDecoder(self.models["encoder"].num_ch_enc, self.opt.scales,
                                                                   num_output_channels=2)
        self.parameters_to_train = []
        for m in self.models.values():
            m.to(self.device)
            self.parameters_to_train += list(m.parameters())     # trainer.py:69-113: one Adam group
        if world_size > 1:
            broadcast_parameters(self.models.values(), 0, process_group)
        # in the order the forward runs the modules (GradBuckets exchanges in the reverse of it): with overlap_streams
        # the pose branch is issued first (process_batch), so its gradients are the last ones backward produces
        main = ["encoder"] + [k for k in ("gru",) if k in self.models] + ["depth"] + \
            [k for k in ("fusion", "predictive_mask") if k in self.models]
        order = (["pose_encoder", "pose"] + main) if getattr(self.opt, "overlap_streams", False) else (main + ["pose_encoder", "pose"])
        order = [k for k in order if k in self.models] + [k for k in self.models if k not in order]
        named = [(k + "." + n, p) for k in order for n, p in self.models[k].named_parameters()]
        self.buckets = GradBuckets(named, self.opt.bucket_mb, world_size, process_group)
        # hipGraph mode (opt.hip_graph, one GPU): the step is captured once and replayed, so everything that changes from
        # step to step must live in device memory -- Adam's step count (capturable) and the tie-break noise seed
        # With world_size > 1 the bucketed exchange is part of the captured step: the hooks' all-reduces are recorded on the
        # communication stream (which joins the capture through its stream waits), in the fixed bucket order -- RCCL only
        # (gloo moves data through host memory and cannot be captured)
        import torch.distributed as dist
        capturable_pg = world_size == 1 or (dist.is_available() and dist.is_initialized()
                                            and dist.get_backend(process_group) == "nccl")
        self.graph_enabled = bool(getattr(self.opt, "hip_graph", False)) and self.device.type == "cuda" and capturable_pg
        # weight-gradient kernels on companion streams (ops.WgradLanes): eager steps, with or without the bucketed exchange (its
        # communication stream waits for the lanes that wrote a bucket's slices, depthcore/ddp.py).  Not under hip_graph (as
        # forks of a captured step the lanes measured +10 % at C2 where they give -0.7 % eagerly).  2 = "auto": only where the
        # step is GPU-bound -- a lane costs ~15 us of host work per convolution (stream switches, an event), which a batch-1
        # step that waits for the host pays in full (C1 7.5 -> 8.4 ms) and a 12 x 192 x 640 step hides (C2 -0.7 %, C3 -1.9 %)
        lanes = int(getattr(self.opt, "wgrad_lanes", 0))
        if lanes == 2:
            lanes = int(self.opt.batch_size * self.opt.height * self.opt.width >= 4 * 192 * 640 and not self.opt.gru)
        self.wgrad_lanes = bool(lanes) and self.device.type == "cuda" and not self.graph_enabled
        # the loop's stream priority (on_step_stream): 2 = "auto": the GPU-bound step sizes of the lanes' rule, BasicBlock trunks, no
        # sequence front-end -- C2 -1.4 % (10.98 / 10.95 -> 10.83 / 10.79 ms, alternating runs); measured worse or flat elsewhere:
        # resnet50 at 320 x 1024 +0.8 % (43.19 / 43.28 -> 43.46 / 43.71), Fusion_v3 +-0.1 %, the host-bound steps (a high-priority
        # queue's launches cost the host more) ConvGRU 9.0 -> 9.7 ms, batch 1 7.1 -> 7.5 ms
        prio = int(getattr(self.opt, "step_priority", 0))
        if prio == 2:
            prio = -1 if (self.opt.batch_size * self.opt.height * self.opt.width >= 4 * 192 * 640 and not self.opt.gru
                          and not getattr(self.opt, "fusion", None) and int(self.opt.num_layers) <= 34) else 0
        # (single process only: with the bucketed exchange the RCCL stream's priority would have to follow the step's, and that
        # could not be measured on a one-GPU pool -- the world > 1 step stays exactly the one that was rehearsed)
        self.step_priority = prio if world_size == 1 else 0
        if self.graph_enabled and self.opt.cpu_tiebreak_noise:
            raise ValueError("hip_graph replays cannot include the reference's CPU randn + host-to-device copy (cpu_tiebreak_noise)")
        # reference trainer.py:110-113: one Adam over every trainable tensor.  On the GPU the depthcore kernel (one streaming
        # pass, step counts in device memory so the launch is capturable); `opt.torch_adam` keeps ATen's fused kernel (A/B)
        if self.device.type == "cuda" and not getattr(self.opt, "torch_adam", False):
            from depthcore.optim import Adam as DepthcoreAdam
            self.model_optimizer = DepthcoreAdam(self.parameters_to_train, self.opt.learning_rate)
        else:
            self.model_optimizer = optim.Adam(self.parameters_to_train, self.opt.learning_rate,
                                              fused=self.device.type == "cuda", capturable=self.graph_enabled)
        self._seed_dev = torch.full((1,), rank, dtype=torch.int64, device=self.device) if self.graph_enabled else None
        self._graphs, self._graph, self._graph_warm, self._graph_stream = {}, None, {}, None
        self._capture_pg = None           # world > 1: the process group whose collectives are CAPTURED (never used eagerly)
        self.model_lr_scheduler = optim.lr_scheduler.StepLR(self.model_optimizer, self.opt.scheduler_step_size, 0.1)

        self.ssim = SSIM()
        self.backproject_depth, self.project_3d = {}, {}
        for s in self.opt.scales:
            h, w = self.opt.height // (2 ** s), self.opt.width // (2 ** s)
            self.backproject_depth[s] = BackprojectDepth(self.loss_batch, h, w).to(self.device)
            self.project_3d[s] = Project3D(self.loss_batch, h, w).to(self.device)
        self.step = 0
        self.epoch = 0
        self._side_stream = None
        self._main_stream = None          # the high-priority stream of the eager step (_step_stream)
        use_cache = self.device.type == "cuda" and getattr(self.opt, "wino_weight_cache", True)
        self.wino_cache = WinoWeightCache(self.parameters_to_train) if use_cache else _NoWinoCache()
        if getattr(self.opt, "load_weights_folder", None):       # trainer.py:137-138
            self.load_model()

    def set_train(self):
        for m in self.models.values():
            m.train()

    def freeze_hidden_states(self):
        """trainer_gru.py:295-307: from epoch `opt.h_s_epoch` on the learned initial hidden states of the ConvGRU cells stop
        training (`h0_layer1.requires_grad = False`; they stay in Adam's parameter list and simply receive no gradient any
        more, as in the reference).  Captured hipGraphs recorded the old autograd graph and are dropped (re-captured after
        the usual warm-up)."""
        if "gru" not in self.models:
            return
        for cell in self.models["gru"].cells():
            cell.h0_layer1.requires_grad = False
            cell.h0_layer1.grad = None
        self.reset_graphs()

    def start_epoch(self, epoch):
        """Per-epoch hooks of the reference's train loops: trainer_gru.py:295 (`(epoch + 1) == h_s_epoch`)."""
        self.epoch = epoch
        if self.opt.gru and (epoch + 1) == self.opt.h_s_epoch:
            self.freeze_hidden_states()

    def set_eval(self):
        for m in self.models.values():
            m.eval()

    # ------------------------------------------------------------------ trainer.py:256-376
    def _stack_sequence(self, inputs):
        """Sequence dict (datasets/kitti_dataset_seq.py:110-140: ("color", f, s, j), ("K", s, j), ("inv_K", s, j), j = frame
        of the sequence) -> the ordinary dict with the sequence stacked along the batch, as trainer_gru.py:819-821,886-896,
        943-944 concatenate it at every use.  The GRU trainer feeds the un-augmented images to both networks."""
        n = self.opt.len_sequence
        keys = [("color", f, s) for f in (0, -1, 1) for s in self.opt.scales if ("color", f, s, 0) in inputs]
        keys += [(k, s) for s in self.opt.scales for k in ("K", "inv_K")]
        groups = [[inputs[key + (j,)] for j in range(n)] for key in keys]
        if groups[0][0].is_cuda and all(t.dtype == torch.float32 for g in groups for t in g):
            stacked = ops.stack_frames(groups)                   # one launch for the 20 concatenations
        else:
            stacked = [torch.cat(g, 0) for g in groups]
        out = dict(zip(keys, stacked))
        for key in keys:
            if key[0] == "color":
                out[("color_aug",) + key[1:]] = out[key]
        return out

    def process_batch(self, inputs):
        """`opt.nets_dtype = "bf16"`: the reduced-precision-networks policy of BASELINE configs[4] -- every convolution of the
        networks rounds its operands to bf16 for the matrix cores (fp32 accumulate); tensors, master weights, BatchNorm
        statistics and the whole photometric loss stay fp32 (depthcore.ops.matrix_precision)."""
        with ops.matrix_precision(getattr(self.opt, "nets_dtype", "f32")):
            return self._process_batch(inputs)

    def _process_batch(self, inputs):
        for key, ipt in inputs.items():
            if ipt.device != self.device:
                inputs[key] = ipt.to(self.device)
        if self.opt.gru:
            inputs = self._stack_sequence(inputs)
        if self.opt.pose_model_type == "shared":
            # trainer.py:263-279: every frame goes through the depth encoder (ONE pass, the frames stacked along the batch --
            # BatchNorm statistics over all of them, as in the reference), the pose decoder reads the per-frame features
            outputs, features = self._depth_branch_shared(inputs)
            outputs.update(self.predict_poses(inputs, features))
        elif getattr(self.opt, "overlap_streams", False) and self.device.type == "cuda":
            # The pose network (pose encoder + decoder) and the depth network are independent until the loss: run them
            # on two HIP streams so that one branch's kernels fill the other's tails and small launches.  Autograd
            # replays each backward op on its forward stream, so the backward overlaps the same way.
            main = torch.cuda.current_stream(self.device)
            if self._side_stream is None:
                self._side_stream = torch.cuda.Stream(self.device)      # (normal priority: see on_step_stream)
                self.buckets.streams = [main, self._side_stream]
            side = self._side_stream
            side.wait_stream(main)
            with torch.cuda.stream(side):
                pose_out = self.predict_poses(inputs, None)
            outputs = self._depth_branch(inputs)
            main.wait_stream(side)
            for t in pose_out.values():
                t.record_stream(main)
            outputs.update(pose_out)
        else:
            outputs = self._depth_branch(inputs)
            outputs.update(self.predict_poses(inputs, None))
        if self.opt.fused_loss:
            losses = self.fused_losses_v1(inputs, outputs) if self.opt.v1_multiscale else self.fused_losses(inputs, outputs)
        else:
            self.generate_images_pred(inputs, outputs)
            losses = self.compute_losses(inputs, outputs)
        return outputs, losses

    def _depth_branch(self, inputs):
        """trainer.py:276-310 (vanilla wiring) / trainer_fusion_v3.py:311-330 (frames [-2, -1, 0] stacked along the batch
        through encoder + decoder, then Fusion_v3: the decoder output of frame -2 is the one refined -- as in the reference)."""
        if self.opt.fusion:
            enc_input = torch.cat([inputs[("color_aug", i, 0)] for i in (-2, -1, 0)], 0)
            features = self.models["encoder"](enc_input)
            outputs = dict(self.models["fusion"](self.models["depth"](features)))
        elif self.opt.gru:                                       # trainer_gru.py:595-644, batch size 1
            features = self.models["encoder"](inputs[("color", 0, 0)])
            outputs = dict(self.models["depth"](self.models["gru"].run_sequence(features)))
        else:
            features = self.models["encoder"](inputs[("color_aug", 0, 0)])
            outputs = dict(self.models["depth"](features))
        if self.opt.predictive_mask:                             # trainer.py:307-310
            outputs["predictive_mask"] = self.models["predictive_mask"](features)
        return outputs

    def _depth_branch_shared(self, inputs):
        ids = list(self.opt.frame_ids)
        B = self.opt.batch_size
        all_features = self.models["encoder"](torch.cat([inputs[("color_aug", i, 0)] for i in ids], 0))
        features = {k: [f[i * B:(i + 1) * B] for f in all_features] for i, k in enumerate(ids)}
        outputs = dict(self.models["depth"](features[0]))
        if self.opt.predictive_mask:
            outputs["predictive_mask"] = self.models["predictive_mask"](features[0])
        return outputs, features

    # ------------------------------------------------------------------ trainer.py:378-442
    def predict_poses(self, inputs, features):
        """Default (separate_resnet, pairs): both temporally ordered pairs (-1,0), (0,+1) go through the pose encoder in ONE
        pass: they are stacked along the batch and BatchNorm keeps separate statistics per pair (`bn_groups=2`), which is
        exactly the arithmetic of the reference's two sequential passes (trainer.py:398-419) at twice the GEMM N.
        Every other pose_model_type / pose_model_input follows trainer.py:378-442 statement by statement."""
        if self.opt.pose_model_type != "separate_resnet" or self.opt.pose_model_input != "pairs":
            return self._predict_poses_modes(inputs, features)
        outputs = {}
        pose_feats = {f: inputs[("color_aug", f, 0)] for f in (-1, 0, 1)}
        B = pose_feats[0].shape[0]
        # (the pair tensors cat([f-1, f0], 1), cat([f0, f+1], 1) of trainer.py:398-412 are formed by the stem kernel's loader)
        feats = self.models["pose_encoder"].forward_pairs(pose_feats[-1], pose_feats[0], pose_feats[1])
        # rows [0, B): the pair (-1, 0), inverted (trainer.py:416-419); rows [B, 2B): the pair (0, +1); predicted frame 0 of each
        axisangle, translation, Ts = self.models["pose"].forward_poses([feats], [(0, B, 0, 1), (B, B, 0, 0)])
        for i, f in enumerate((-1, 1)):
            outputs[("axisangle", 0, f)] = axisangle[i * B:(i + 1) * B]
            outputs[("translation", 0, f)] = translation[i * B:(i + 1) * B]
            outputs[("cam_T_cam", 0, f)] = Ts[i]
        return outputs

    def _poses(self, pose, pose_inputs, groups):
        """(axisangle, translation, [cam_T_cam per group]).  The fused tail (`forward_poses`: one launch each way) hands out
        axisangle / translation as RECORDS -- the loss reaches the network through the matrices.  posecnn mode is the one place
        where the loss reads the vectors themselves (generate_images_pred rebuilds the pose from them at every scale,
        trainer.py:490-499), so there the module's plain forward (differentiable outputs) + transformation_from_parameters."""
        if self.opt.pose_model_type != "posecnn":
            return pose.forward_poses(pose_inputs, groups)
        axisangle, translation = pose(pose_inputs)
        Ts = [transformation_from_parameters(axisangle[r0:r0 + rows, slot], translation[r0:r0 + rows, slot], invert=bool(inv))
              for r0, rows, slot, inv in groups]
        return axisangle, translation, Ts

    def _predict_poses_modes(self, inputs, features):
        o, outputs = self.opt, {}
        pose = self.models["pose"]
        if self.num_pose_frames == 2:                            # trainer.py:383-419: one pass per source frame
            feats = {f: features[f] for f in o.frame_ids} if o.pose_model_type == "shared" else \
                {f: inputs[("color_aug", f, 0)] for f in (-1, 0, 1)}
            for f in (-1, 1):
                pose_inputs = [feats[f], feats[0]] if f < 0 else [feats[0], feats[f]]        # temporal order
                if o.pose_model_type == "separate_resnet":
                    pose_inputs = [self.models["pose_encoder"](torch.cat(pose_inputs, 1))]
                elif o.pose_model_type == "posecnn":
                    pose_inputs = torch.cat(pose_inputs, 1)
                B = inputs[("color_aug", 0, 0)].shape[0]
                axisangle, translation, Ts = self._poses(pose, pose_inputs, [(0, B, 0, int(f < 0))])
                outputs[("axisangle", 0, f)] = axisangle
                outputs[("translation", 0, f)] = translation
                outputs[("cam_T_cam", 0, f)] = Ts[0]
            return outputs
        ids = [i for i in o.frame_ids if i != "s"]               # trainer.py:421-440: all frames in, all poses out
        if o.pose_model_type in ("separate_resnet", "posecnn"):
            pose_inputs = torch.cat([inputs[("color_aug", i, 0)] for i in ids], 1)
            if o.pose_model_type == "separate_resnet":
                pose_inputs = [self.models["pose_encoder"](pose_inputs)]
        else:
            pose_inputs = [features[i] for i in ids]
        B = inputs[("color_aug", 0, 0)].shape[0]
        used = [(i, f) for i, f in enumerate(o.frame_ids[1:]) if f != "s"]
        axisangle, translation, Ts = self._poses(pose, pose_inputs, [(0, B, i, 0) for i, _ in used])
        for (i, f), T_ in zip(used, Ts):
            outputs[("axisangle", 0, f)] = axisangle
            outputs[("translation", 0, f)] = translation
            outputs[("cam_T_cam", 0, f)] = T_
        return outputs

    # ------------------------------------------------------------------ fused a14 + a15
    def _noise(self, B, nch):
        if not self.opt.cpu_tiebreak_noise or self.opt.disable_automasking:
            return None
        # the reference draws on the CPU generator and copies (trainer.py:594-595); v1_multiscale: at each scale's size
        v1 = self.opt.v1_multiscale
        return [torch.randn(B, nch, self.opt.height >> (s if v1 else 0), self.opt.width >> (s if v1 else 0)).to(self.device)
                for s in self.opt.scales]

    def fused_losses(self, inputs, outputs, materialize=None):
        o = self.opt
        materialize = o.materialize_logs if materialize is None else materialize
        B = inputs[("color", 0, 0)].shape[0]
        # the data step's pixel-interleaved copies of the three full-resolution frames (depthcore.data / synthetic_batch):
        # with them the loss does not repack the frames every step; a reference data loader does not provide them
        packed = tuple(inputs.get(("color_packed", f, 0)) for f in (0, -1, 1))
        packed = packed if all(t is not None for t in packed) else None
        cfg = ops.PhotoConfig(
            inputs[("color", 0, 0)], inputs[("color", -1, 0)], inputs[("color", 1, 0)],
            [inputs[("color", 0, s)] for s in o.scales], inputs[("K", 0)], inputs[("inv_K", 0)],
            noise=self._noise(B, 1 if o.avg_reprojection else 2), min_depth=o.min_depth, max_depth=o.max_depth,
            smoothness=o.disparity_smoothness, disable_automasking=o.disable_automasking,
            avg_reprojection=o.avg_reprojection, no_ssim=o.no_ssim, materialize=materialize, packed=packed,
            rng_seed=self._seed_dev if self._seed_dev is not None else self.step * 1000003 + self.rank)
        masks, bce = self._predictive_masks(outputs, full_res=True)
        if o.pose_model_type == "posecnn":                       # trainer.py:490-499: one pose per (scale, frame)
            lv = ops.photometric_loss(cfg, None, None, [outputs[("disp", s)] for s in o.scales], pred_masks=masks,
                                      T_scales=self._posecnn_T_scales(outputs))
        else:
            lv = ops.photometric_loss(cfg, outputs[("cam_T_cam", 0, -1)], outputs[("cam_T_cam", 0, 1)],
                                      [outputs[("disp", s)] for s in o.scales], pred_masks=masks)
        if masks is None:
            losses = {"loss/{}".format(s): lv[i] for i, s in enumerate(o.scales)}
            losses["loss"] = lv[len(o.scales)]
        else:           # + the masks' BCE weighting term per scale (trainer.py:579-581); total = mean over the scales
            losses = {"loss/{}".format(s): lv[i] + bce[i] for i, s in enumerate(o.scales)}
            losses["loss"] = sum(losses["loss/{}".format(s)] for s in o.scales) / self.num_scales
        ex = cfg.extras
        for i, s in enumerate(o.scales):
            outputs[("argmin", s)] = ex["argmin"][i]
            if materialize:
                outputs[("depth", 0, s)] = ex["depth"][i]
                for j, f in enumerate((-1, 1)):
                    outputs[("sample", f, s)] = ex["sample"][i][j]
                    outputs[("color", f, s)] = ex["color"][i][j]
                    if not o.disable_automasking:
                        outputs[("color_identity", f, s)] = inputs[("color", f, 0)]
                if not o.disable_automasking:
                    outputs["identity_selection/{}".format(s)] = ex["identity_selection"][i]
        return losses

    def _predictive_masks(self, outputs, full_res):
        """opt.predictive_mask (trainer.py:571-584): the mask decoder's outputs, upsampled to the loss resolution unless every
        scale works at its own (`v1_multiscale`), and their weighting term 0.2 * BCE(mask, 1) (-log clamped at -100, mean)."""
        if not self.opt.predictive_mask:
            return None, None
        o = self.opt
        masks = [outputs["predictive_mask"][("disp", s)] for s in o.scales]
        if full_res:
            masks = [interpolate_bilinear(m, [o.height, o.width]) for m in masks]
        return masks, [0.2 * (-torch.clamp(torch.log(m), min=-100.0)).mean() for m in masks]

    def fused_losses_v1(self, inputs, outputs, materialize=None):
        """`opt.v1_multiscale` (trainer.py:470-472, 541-544) on the fused kernels: every scale warps and compares at its OWN
        resolution -- its own images, intrinsics, identity losses and tie-break noise -- so the step is one fused call per
        scale, each a single-scale problem of size (H >> s, W >> s) (the kernels' scale-0 wave: the upsample is the identity),
        with the smoothness weight divided by 2^s as trainer.py:616 does; total = mean over the scales."""
        o = self.opt
        materialize = o.materialize_logs if materialize is None else materialize
        B = inputs[("color", 0, 0)].shape[0]
        noise = self._noise(B, 1 if o.avg_reprojection else 2)
        masks, bce = self._predictive_masks(outputs, full_res=False)
        losses, total = {}, 0
        for i, s in enumerate(o.scales):
            cfg = ops.PhotoConfig(
                inputs[("color", 0, s)], inputs[("color", -1, s)], inputs[("color", 1, s)], [inputs[("color", 0, s)]],
                inputs[("K", s)], inputs[("inv_K", s)], noise=None if noise is None else [noise[i]], min_depth=o.min_depth,
                max_depth=o.max_depth, smoothness=o.disparity_smoothness / (2 ** s), disable_automasking=o.disable_automasking,
                avg_reprojection=o.avg_reprojection, no_ssim=o.no_ssim, materialize=materialize,
                rng_seed=(self._seed_dev + i) if self._seed_dev is not None else self.step * 1000003 + self.rank + 7919 * i)
            if o.pose_model_type == "posecnn":                   # trainer.py:490-499 at this scale's own resolution
                _, depth = disp_to_depth(outputs[("disp", s)], o.min_depth, o.max_depth)
                lv = ops.photometric_loss(cfg, None, None, [outputs[("disp", s)]], pred_masks=None if masks is None else [masks[i]],
                                          T_scales=[tuple(self._posecnn_T(outputs, f, depth) for f in (-1, 1))])
            else:
                lv = ops.photometric_loss(cfg, outputs[("cam_T_cam", 0, -1)], outputs[("cam_T_cam", 0, 1)], [outputs[("disp", s)]],
                                          pred_masks=None if masks is None else [masks[i]])
            losses["loss/{}".format(s)] = lv[0] if masks is None else lv[0] + bce[i]
            total = total + losses["loss/{}".format(s)]
            ex = cfg.extras
            outputs[("argmin", s)] = ex["argmin"][0]
            if materialize:
                outputs[("depth", 0, s)] = ex["depth"][0]
                for j, f in enumerate((-1, 1)):
                    outputs[("sample", f, s)] = ex["sample"][0][j]
                    outputs[("color", f, s)] = ex["color"][0][j]
                    if not o.disable_automasking:
                        outputs[("color_identity", f, s)] = inputs[("color", f, s)]
                if not o.disable_automasking:
                    outputs["identity_selection/{}".format(s)] = ex["identity_selection"][0]
        losses["loss"] = total / self.num_scales
        return losses

    def _posecnn_T(self, outputs, frame_id, depth):
        """trainer.py:490-499 (`pose_model_type == "posecnn"`, after arXiv:1712.00175): the predicted translation is rescaled by
        the mean inverse depth of THIS scale's (upsampled) depth map and the pose matrix rebuilt from it."""
        inv_depth = 1 / depth
        mean_inv_depth = inv_depth.mean(3, True).mean(2, True)
        return transformation_from_parameters(
            outputs[("axisangle", 0, frame_id)][:, 0], outputs[("translation", 0, frame_id)][:, 0] * mean_inv_depth[:, 0],
            frame_id < 0)

    def _posecnn_T_scales(self, outputs):
        """The per-scale pose pairs of the fused loss in posecnn mode: depth of scale s = disp_to_depth(upsampled disp_s), as
        generate_images_pred forms it."""
        o, pairs = self.opt, []
        for s in o.scales:
            disp = interpolate_bilinear(outputs[("disp", s)], [o.height, o.width])
            _, depth = disp_to_depth(disp, o.min_depth, o.max_depth)
            pairs.append(tuple(self._posecnn_T(outputs, f, depth) for f in (-1, 1)))
        return pairs

    # ------------------------------------------------------------------ trainer.py:465-515
    def generate_images_pred(self, inputs, outputs):
        o = self.opt
        for scale in o.scales:
            if o.v1_multiscale:                                  # trainer.py:470-472: warp at each scale's own resolution
                disp, source_scale = outputs[("disp", scale)], scale
            else:
                disp = interpolate_bilinear(outputs[("disp", scale)], [o.height, o.width])
                source_scale = 0
            _, depth = disp_to_depth(disp, o.min_depth, o.max_depth)
            outputs[("depth", 0, scale)] = depth
            for frame_id in (-1, 1):
                T = outputs[("cam_T_cam", 0, frame_id)]
                if o.pose_model_type == "posecnn":                # trainer.py:490-499
                    T = self._posecnn_T(outputs, frame_id, depth)
                cam_points = self.backproject_depth[source_scale](depth, inputs[("inv_K", source_scale)])
                pix_coords = self.project_3d[source_scale](cam_points, inputs[("K", source_scale)], T)
                outputs[("sample", frame_id, scale)] = pix_coords
                outputs[("color", frame_id, scale)] = grid_sample(
                    inputs[("color", frame_id, source_scale)], pix_coords, padding_mode="border")
                if not o.disable_automasking:
                    outputs[("color_identity", frame_id, scale)] = inputs[("color", frame_id, source_scale)]

    # ------------------------------------------------------------------ trainer.py:517-529
    def compute_reprojection_loss(self, pred, target):
        l1_loss = torch.abs(target - pred).mean(1, True)
        if self.opt.no_ssim:
            return l1_loss
        return 0.85 * self.ssim(pred, target).mean(1, True) + 0.15 * l1_loss

    # ------------------------------------------------------------------ trainer.py:531-622
    def compute_losses(self, inputs, outputs):
        o = self.opt
        losses, total_loss = {}, 0
        B = inputs[("color", 0, 0)].shape[0]
        noise = self._noise(B, 1 if o.avg_reprojection else 2)
        for si, scale in enumerate(o.scales):
            source_scale = scale if o.v1_multiscale else 0       # trainer.py:541-544
            disp, color, target = outputs[("disp", scale)], inputs[("color", 0, scale)], inputs[("color", 0, source_scale)]
            reproj = torch.cat([self.compute_reprojection_loss(outputs[("color", f, scale)], target)
                                for f in (-1, 1)], 1)
            loss = 0
            if o.predictive_mask:                                # trainer.py:571-584 (automasking is off here)
                mask = outputs["predictive_mask"][("disp", scale)]
                if not o.v1_multiscale:
                    mask = interpolate_bilinear(mask, [o.height, o.width])
                reproj = reproj * mask
                # nn.BCELoss()(mask, ones): -log(mask) with the log clamped at -100, mean over all elements
                loss = loss + 0.2 * (-torch.clamp(torch.log(mask), min=-100.0)).mean()
            if o.avg_reprojection:
                reproj = reproj.mean(1, keepdim=True)
            if not o.disable_automasking:
                ident = torch.cat([self.compute_reprojection_loss(inputs[("color", f, source_scale)], target)
                                   for f in (-1, 1)], 1)
                if o.avg_reprojection:
                    ident = ident.mean(1, keepdim=True)
                nz = noise[si] if noise is not None else torch.randn(ident.shape, device=self.device)
                combined = torch.cat((ident + nz * 0.00001, reproj), dim=1)
            else:
                combined = reproj
            if combined.shape[1] == 1:
                to_optimise = combined
            else:
                to_optimise, idxs = torch.min(combined, dim=1)
                if not o.disable_automasking:
                    outputs["identity_selection/{}".format(scale)] = (idxs > ident.shape[1] - 1).float()
            loss = loss + to_optimise.mean()
            norm_disp = disp / (disp.mean(2, True).mean(3, True) + 1e-7)
            loss = loss + o.disparity_smoothness * get_smooth_loss(norm_disp, color) / (2 ** scale)
            total_loss = total_loss + loss
            losses["loss/{}".format(scale)] = loss
        losses["loss"] = total_loss / self.num_scales
        return losses

    # ------------------------------------------------------------------ trainer.py:700-763
    @property
    def log_path(self):
        return os.path.join(self.opt.log_dir, self.opt.model_name)          # trainer.py:35

    def save_opts(self, models_dir=None):
        """trainer.py:700-709: `<log_path>/models/opt.json` with every option."""
        models_dir = models_dir or os.path.join(self.log_path, "models")
        os.makedirs(models_dir, exist_ok=True)
        with open(os.path.join(models_dir, "opt.json"), "w") as f:
            json.dump({k: v for k, v in vars(self.opt).items() if isinstance(v, (int, float, str, bool, list, type(None)))},
                      f, indent=2)

    def save_model(self, folder=None):
        """Reference checkpoint layout (trainer.py:711-729): `<log_path>/models/weights_{epoch}/{name}.pth` = state_dict per
        model (the encoder's additionally carries height / width / use_stereo) + `adam.pth`.  `folder` overrides the
        location (then opt.json is written next to the weights as well)."""
        explicit = folder is not None
        folder = folder or os.path.join(self.log_path, "models", "weights_{}".format(getattr(self, "epoch", 0)))
        os.makedirs(folder, exist_ok=True)
        for name, model in self.models.items():
            to_save = model.state_dict()
            if name == "encoder":
                to_save["height"], to_save["width"] = self.opt.height, self.opt.width
                to_save["use_stereo"] = getattr(self.opt, "use_stereo", False)
            torch.save(to_save, os.path.join(folder, "{}.pth".format(name)))
        torch.save(self.model_optimizer.state_dict(), os.path.join(folder, "adam.pth"))
        self.save_opts(folder if explicit else None)
        return folder

    def load_model(self, folder=None, models_to_load=None):
        """trainer.py:731-763: `opt.load_weights_folder` / `opt.models_to_load` by default; for every model keep only the
        keys the current module has (drops the encoder's height / width / use_stereo entries) and leave the others at their
        initial values.  Names of `models_to_load` that this trainer does not have (the reference's default list also names
        `gru` and `head`) are skipped; a listed model without its file is an error, as in the reference."""
        folder = os.path.expanduser(folder or self.opt.load_weights_folder)
        if not os.path.isdir(folder):
            raise FileNotFoundError("Cannot find folder {}".format(folder))
        names = list(models_to_load if models_to_load is not None else self.opt.models_to_load)
        for name in names[::-1]:
            if name not in self.models:
                continue
            path = os.path.join(folder, "{}.pth".format(name))
            model_dict = self.models[name].state_dict()
            pretrained = torch.load(path, map_location=self.device)
            model_dict.update({k: v for k, v in pretrained.items() if k in model_dict})
            self.models[name].load_state_dict(model_dict)
        adam = os.path.join(folder, "adam.pth")
        if os.path.isfile(adam):
            self.model_optimizer.load_state_dict(torch.load(adam, map_location=self.device))
        # a captured step names the optimiser's moment buffers and step counts (re-homed by load_state_dict): re-capture
        self.reset_graphs()

    # ------------------------------------------------------------------ trainer.py:233-237
    GRAPH_WARMUP = 3      # eager steps before a capture: lazy allocations, LDS attributes, the weight cache's variants

    def _step_stream(self):
        """The HIGH-priority stream of `on_step_stream()` (eager steps with overlapping branches, self.step_priority = -1), or None."""
        if (self.device.type != "cuda" or not getattr(self.opt, "overlap_streams", False) or self.graph_enabled
                or self.step_priority >= 0):
            return None
        if self._main_stream is None:
            self._main_stream = torch.cuda.Stream(self.device, priority=-1)
        return self._main_stream

    @contextlib.contextmanager
    def on_step_stream(self):
        """Run a training LOOP on the trainer's high-priority stream:  `with trainer.on_step_stream(): for batch in loader:
        trainer.train_step(batch)`.  Inside, the depth branch -- encoder, decoder, loss and their backward, the step's longest
        dependency chain -- is enqueued at high priority while the pose branch (its side stream) and the weight-gradient lanes stay
        at normal priority: the command processor hands the chain's workgroups out first and the other two fill what it leaves
        (same-box A/B at C2, alternating runs: 10.96 / 10.96 / 11.00 -> 10.83 / 10.83 / 10.81 ms; the POSE branch at high priority
        instead: 11.13).  HIP offers two levels here (torch.cuda.Stream.priority_range() = (0, -1)) and the caller's default stream
        sits at the lower one, hence a stream of the trainer's own.  The hand-over between the caller's stream and this one is made
        ONCE, around the loop: made per step (a wait on either side of every train_step) it costs more than the priority gains
        (10.97-11.06), so train_step() outside this context simply stays on the caller's stream.  No-op (yields at once) on the CPU,
        with opt.overlap_streams = 0, a captured step (opt.hip_graph), opt.step_priority = 0, or = 2 ("auto") outside the configurations where
        it was measured to pay (see __init__)."""
        hp = self._step_stream()
        if hp is None:
            yield
            return
        cur = torch.cuda.current_stream(self.device)
        hp.wait_stream(cur)
        with torch.cuda.stream(hp):
            yield
        cur.wait_stream(hp)

    def train_step(self, inputs):
        if not self.graph_enabled:
            return self._train_step_eager(inputs)
        shapes = tuple((k, tuple(v.shape)) for k, v in inputs.items())
        key = (shapes, tuple(g["lr"] for g in self.model_optimizer.param_groups))
        if key in self._graphs and self._graphs[key] is None:      # a capture of this shape failed before: eager
            return self._eager_on_graph_stream(inputs)
        entry = self._graphs.get(key)
        if entry is None:
            # Everything a capture will see must already have happened on a NON-default stream: autograd binds each
            # parameter's AccumulateGrad node to the stream of its first use, and a node bound to the legacy default stream
            # makes the backward touch that stream inside the capture (fatal).  So the eager warm-up steps of graph mode
            # run on the stream the capture will use.
            if self._graph_stream is None:
                self._graph_stream = torch.cuda.Stream(self.device)
            gs, cur = self._graph_stream, torch.cuda.current_stream(self.device)
            # Warm-up is counted PER INPUT SHAPE: a new shape meets new kernel variants, lazy workspace sizes and weight-cache
            # variants, and all of that (allocations, attribute calls, a synchronous table upload) must happen in eager
            # steps, not inside a capture.  A learning-rate change on a known shape captures at once.
            warm = self._graph_warm.get(shapes, 0)
            if warm < self.GRAPH_WARMUP:
                self._graph_warm[shapes] = warm + 1
                return self._eager_on_graph_stream(inputs)
            # capture: records the launches of one step (both streams, backward, Adam) without running them.
            # torch.cuda.graph() runs gc.collect() + empty_cache() on entry; a model that died in a reference cycle is
            # finalised there -- INSIDE the capture, when nothing may allocate, copy or rebuild a table any more.  Collect
            # first, so that every finaliser (weight-cache owners, workspaces) has run while the stream is still eager.
            import gc
            gc.collect()
            self._evict_graphs(key)
            static_in = {k: v.clone() for k, v in inputs.items()}
            torch.cuda.synchronize(self.device)
            cap_pg = None
            if self.world_size > 1:
                # The captured collectives run on a process group of their own that has never executed anything eagerly.
                # (Cause of the round-4 abort: RCCL's watchdog thread retires finished eager collectives by querying their end
                # events, recorded on the group's internal stream; once that stream has joined a capture the query fails with
                # hipErrorCapturedEvent and the watchdog abort()s.  The warm-up steps' group is never part of a capture now, so
                # its watchdog only ever queries events of an eager stream; the capture group's watchdog has nothing to query,
                # since torch does not enqueue collectives issued during a capture.  A condition, not a sleep.)
                cap_pg = self._capture_group()
                # (the decision is COLLECTIVE: a rank that stays eager on the base group while the others replay graphs on the
                # capture group would leave both sides waiting on communicators the other never enters)
                if not self._agree(cap_pg is not None):     # no communicator without a collective on some rank: all stay eager
                    self.graph_enabled = False
                    return self._eager_on_graph_stream(inputs)
            g = torch.cuda.CUDAGraph()
            step0 = self.step
            stream0 = torch.cuda.current_stream(self.device)
            err = None
            try:
                if self._fail_capture_for_test():
                    raise RuntimeError("capture failure forced by DC_TEST_FAIL_CAPTURE_RANK")
                # (world > 1: RCCL's watchdog thread polls events while we capture -- legal, but only in thread-local mode)
                mode = {} if self.world_size == 1 else {"capture_error_mode": "thread_local"}
                pg0 = self.buckets.pg
                try:
                    if cap_pg is not None:
                        self.buckets.pg = cap_pg
                    with torch.cuda.graph(g, stream=gs, **mode):
                        static_out = self._train_step_eager(static_in)
                finally:
                    self.buckets.pg = pg0
            except Exception as e:       # a launch refused inside the capture, or the capture was invalidated
                # Nothing of a captured step has run (capture records, it does not execute): host state is rolled back, the
                # shape is marked eager-only and the step is done eagerly -- a failed capture costs speed, never the process.
                from depthcore import _lib
                err = e
                self.step = step0
                self._graph_failed = getattr(self, "_graph_failed", 0) + 1
                torch.cuda.set_stream(stream0)          # (torch.cuda.graph.__exit__ does not restore it when capture_end raises)
                # an invalidated capture leaves its origin stream capturing (every later launch on it would fail): end it
                _lib.lib().dc_abort_capture(gs.cuda_stream)
                _lib.lib().dc_clear_error()             # the failed capture's error code is not the next launch's
                # CUDAGraph.capture_end ends the stream capture BEFORE it tells the caching allocator that the capture is over:
                # when the former throws, the allocator would keep routing this stream to the graph's pool and never process
                # its deferred frees (record_stream blocks) again -- a slow leak in the eager fallback.  Close it here.
                try:
                    torch._C._cuda_endAllocateToPool(self.device.index or 0, g.pool())
                    torch._C._cuda_releasePool(self.device.index or 0, g.pool())
                except Exception:
                    pass
            # World > 1: the outcome is agreed over the ranks (on the eager base group, outside any capture).  If the capture failed
            # anywhere, every rank discards its graph for this shape and all of them run it eagerly on the base group -- the
            # replays' collectives live on the capture group, the eager steps' on the base group, and mixed ranks never meet.
            if self.world_size > 1 and err is None:
                torch.cuda.synchronize(self.device)
            if not self._agree(err is None) and err is None:
                err = RuntimeError("the capture failed on another rank")
                self.step = step0
                self._graph_agreed_off = getattr(self, "_graph_agreed_off", 0) + 1
                g.reset()
            if err is not None:
                import warnings
                self._graphs[key] = None
                warnings.warn("hip_graph: capture of the training step failed (%s: %s); this input shape runs eagerly"
                              % (type(err).__name__, err))
                try:
                    torch.cuda.synchronize(self.device)
                    return self._eager_on_graph_stream(inputs)
                except Exception as e2:
                    # an ILLEGAL call inside the capture (a synchronisation, an allocation by foreign code) can leave the HIP
                    # runtime's capture state beyond repair for this process: say so instead of failing somewhere else
                    raise RuntimeError("hip_graph: the capture failed (%s: %s) and the streams could not be recovered (%s: %s); "
                                       "run without opt.hip_graph" % (type(err).__name__, err, type(e2).__name__, e2)) from err
            entry = self._graphs[key] = (g, static_in, static_out)
            self.step -= 1                      # (the recorded step has not run yet: the replay below is that step)
        else:
            for k, v in inputs.items():
                if v is not entry[1][k]:
                    entry[1][k].copy_(v, non_blocking=True)
        self._graph = entry[0]
        entry[0].replay()
        self.step += 1
        return entry[2]                       # static tensors: rewritten by every replay of this graph

    def _agree(self, ok):
        """`ok` made collective over the gradient exchange's ranks (depthcore.ddp.agree_all on the EAGER base group); plain `ok`
        at world size 1."""
        if self.world_size == 1:
            return bool(ok)
        return agree_all(ok, self.buckets.pg, self.device)

    def _fail_capture_for_test(self):
        """tests/ddp_graph_child.py: DC_TEST_FAIL_CAPTURE_RANK=<rank> makes that rank's first capture attempt fail."""
        r = os.environ.get("DC_TEST_FAIL_CAPTURE_RANK")
        return r is not None and int(r) == self.rank and not getattr(self, "_graph_failed", 0)

    def _capture_group(self):
        """The process group of the captured steps' collectives: same ranks as the gradient exchange's group, RCCL, its
        communicator connected eagerly (no collective, hence no work item its watchdog would ever poll).  Collective over the
        ranks -- every rank reaches its first capture at the same step.  None if the communicator cannot be had that way."""
        if self._capture_pg is None:
            import torch.distributed as dist
            base = self.buckets.pg if self.buckets.pg is not None else dist.group.WORLD
            pg = dist.new_group(ranks=dist.get_process_group_ranks(base), backend="nccl", device_id=self.device)
            be = pg._get_backend(self.device)
            try:
                # (new_group(device_id=...) connects eagerly where the backend supports it; asking again is a no-op then.
                # `_is_initialized()` is not a usable signal on this stack -- it stays False for a connected communicator -- so
                # the evidence is that the connect call returns; a communicator that is still missing would be created by
                # the first captured collective, fail the capture, and the step falls back to eager: train_step's except path)
                be.eager_connect_single_device(self.device)
                ok = True
            except Exception:
                ok = False
            self._capture_pg = pg if ok else False
        return self._capture_pg or None

    def _eager_on_graph_stream(self, inputs):
        gs, cur = self._graph_stream, torch.cuda.current_stream(self.device)
        gs.wait_stream(cur)
        with torch.cuda.stream(gs):
            out = self._train_step_eager(inputs)
        cur.wait_stream(gs)
        return out

    MAX_GRAPHS = 3        # captured steps kept (each holds a private pool of a whole step's activations and gradients)

    def _evict_graphs(self, new_key):
        """Before a new capture: drop graphs that can never be replayed again (their learning rate is baked into the Adam
        launch and differs from the optimiser's current one) and keep at most MAX_GRAPHS - 1 others, oldest first."""
        lr = new_key[1]
        for k in [k for k in self._graphs if k[1] != lr]:
            self._drop_graph(k)
        live = [k for k, v in self._graphs.items() if v is not None]
        while len(live) >= self.MAX_GRAPHS:
            self._drop_graph(live.pop(0))

    def _drop_graph(self, key):
        entry = self._graphs.pop(key, None)
        if entry is not None:
            if self._graph is entry[0]:
                self._graph = None
            # the host runs ahead of the GPU: the last replay of this graph may still be executing, and reset() destroys the
            # hipGraphExec and hands its private pool back while kernels read their arguments and static buffers
            torch.cuda.synchronize(self.device)
            entry[0].reset()

    def reset_graphs(self):
        """Forget every captured step (their kernel arguments name optimiser state, weights-cache tables and the autograd
        graph of the moment of capture): after load_model(), a rebuilt optimiser, a change of `requires_grad`."""
        for k in list(self._graphs):
            self._drop_graph(k)
        self._graphs, self._graph, self._graph_warm = {}, None, {}

    def close(self):
        """Drop the captured graphs (they name the weight cache's buffers) and this trainer's cache registrations."""
        self.reset_graphs()
        self.wino_cache.close()
        if self._capture_pg:                    # the capture-only communicator (one per trainer that captured at world > 1)
            import torch.distributed as dist
            try:
                dist.destroy_process_group(self._capture_pg)
            except Exception:
                pass
            self._capture_pg = None

    def _train_step_eager(self, inputs):
        self.wino_cache.refresh()               # every 3x3 weight -> Winograd domain, one launch per step
        try:
            outputs, losses = self.process_batch(inputs)
            self.buckets.zero()                     # model_optimizer.zero_grad(set_to_none=True)
            with ops.WgradLanes.active(self.wgrad_lanes):
                losses["loss"].backward()
            ops.assert_no_dangling_sums()       # (a gradient handed to a SkipSum that nobody collected would be a lost gradient)
        finally:
            self.wino_cache.invalidate()        # the optimiser step below rewrites the weights
        self.buckets.finish()                   # RCCL all-reduce (mean) launched from the backward hooks
        self.model_optimizer.step()
        if self._seed_dev is not None:
            self._seed_dev.add_(1000003)        # next step's noise seed (device side: part of a captured step)
        self.step += 1
        return outputs, losses
