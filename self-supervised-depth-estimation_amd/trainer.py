"""The per-batch hot path of the reference's Trainer (reference trainer.py:256-622), MI355X-native.

Same method names and dict schemas as the reference so that code written against
`Trainer.process_batch / predict_poses / generate_images_pred / compute_reprojection_loss /
compute_losses` keeps working; the control plane around it (data loading, tensorboard, checkpoints,
validation) is out of scope (SURVEY 8, rows a14-a18 only).

Two equivalent loss paths, both on hand-written gfx950 kernels:
  * fused (default): generate_images_pred + compute_losses are ONE forward op and ONE backward op
    (`depthcore.ops.photometric_loss`); log tensors are produced only when asked for;
  * layer-by-layer (`opt.fused_loss = False`): the reference's own sequence of `layers.*` calls.
"""
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
from depthcore.ddp import GradBuckets, broadcast_parameters


def default_options(**kw):
    """Hot-path defaults of the reference's options.py:100-213."""
    o = types.SimpleNamespace(
        height=192, width=640, scales=[0, 1, 2, 3], min_depth=0.1, max_depth=100.0, disparity_smoothness=1e-3,
        frame_ids=[0, -1, 1], batch_size=12, learning_rate=1e-4, scheduler_step_size=15, num_layers=18,
        weights_init="scratch", pose_model_type="separate_resnet", pose_model_input="pairs",
        v1_multiscale=False, avg_reprojection=False, disable_automasking=False, predictive_mask=False,
        no_ssim=False, fused_loss=True, cpu_tiebreak_noise=False, materialize_logs=False, bucket_mb=32,
        overlap_streams=True)
    for k, v in kw.items():
        setattr(o, k, v)
    return o


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
        if self.opt.pose_model_type != "separate_resnet" or self.opt.pose_model_input != "pairs":
            raise NotImplementedError("hot path covers pose_model_type=separate_resnet, pose_model_input=pairs")
        if self.opt.predictive_mask or self.opt.v1_multiscale:
            raise NotImplementedError("predictive_mask / v1_multiscale ablations are outside the hot path")
        self.num_scales = len(self.opt.scales)
        self.num_pose_frames = 2

        torch.manual_seed(seed)                                  # same initial weights on every rank
        self.models = {}
        self.models["encoder"] = networks.ResnetEncoder(self.opt.num_layers, self.opt.weights_init == "pretrained")
        self.models["depth"] = networks.DepthDecoder(self.models["encoder"].num_ch_enc, self.opt.scales)
        self.models["pose_encoder"] = networks.ResnetEncoder(self.opt.num_layers, self.opt.weights_init == "pretrained",
                                                             num_input_images=self.num_pose_frames)
        self.models["pose"] = networks.PoseDecoder(self.models["pose_encoder"].num_ch_enc, num_input_features=1,
                                                   num_frames_to_predict_for=2)
        self.parameters_to_train = []
        for m in self.models.values():
            m.to(self.device)
            self.parameters_to_train += list(m.parameters())     # trainer.py:69-113: one Adam group
        if world_size > 1:
            broadcast_parameters(self.models.values(), 0, process_group)
        # in the order the forward runs the modules (GradBuckets exchanges in the reverse of it): with overlap_streams
        # the pose branch is issued first (process_batch), so its gradients are the last ones backward produces
        order = (("pose_encoder", "pose", "encoder", "depth") if getattr(self.opt, "overlap_streams", False)
                 else ("encoder", "depth", "pose_encoder", "pose"))
        order = list(order) + [k for k in self.models if k not in order]
        named = [(k + "." + n, p) for k in order for n, p in self.models[k].named_parameters()]
        self.buckets = GradBuckets(named, self.opt.bucket_mb, world_size, process_group)
        self.model_optimizer = optim.Adam(self.parameters_to_train, self.opt.learning_rate,
                                          fused=self.device.type == "cuda")
        self.model_lr_scheduler = optim.lr_scheduler.StepLR(self.model_optimizer, self.opt.scheduler_step_size, 0.1)

        self.ssim = SSIM()
        self.backproject_depth, self.project_3d = {}, {}
        for s in self.opt.scales:
            h, w = self.opt.height // (2 ** s), self.opt.width // (2 ** s)
            self.backproject_depth[s] = BackprojectDepth(self.opt.batch_size, h, w).to(self.device)
            self.project_3d[s] = Project3D(self.opt.batch_size, h, w).to(self.device)
        self.step = 0
        self._side_stream = None

    def set_train(self):
        for m in self.models.values():
            m.train()

    def set_eval(self):
        for m in self.models.values():
            m.eval()

    # ------------------------------------------------------------------ trainer.py:256-376
    def process_batch(self, inputs):
        for key, ipt in inputs.items():
            if ipt.device != self.device:
                inputs[key] = ipt.to(self.device)
        if getattr(self.opt, "overlap_streams", False) and self.device.type == "cuda":
            # The pose network (pose encoder + decoder) and the depth network are independent until the loss: run them
            # on two HIP streams so that one branch's kernels fill the other's tails and small launches.  Autograd
            # replays each backward op on its forward stream, so the backward overlaps the same way.
            main = torch.cuda.current_stream(self.device)
            if self._side_stream is None:
                self._side_stream = torch.cuda.Stream(self.device)
                self.buckets.streams = [main, self._side_stream]
            side = self._side_stream
            side.wait_stream(main)
            with torch.cuda.stream(side):
                pose_out = self.predict_poses(inputs, None)
            features = self.models["encoder"](inputs[("color_aug", 0, 0)])
            outputs = dict(self.models["depth"](features))
            main.wait_stream(side)
            for t in pose_out.values():
                t.record_stream(main)
            outputs.update(pose_out)
        else:
            features = self.models["encoder"](inputs[("color_aug", 0, 0)])
            outputs = dict(self.models["depth"](features))
            outputs.update(self.predict_poses(inputs, features))
        if self.opt.fused_loss:
            losses = self.fused_losses(inputs, outputs)
        else:
            self.generate_images_pred(inputs, outputs)
            losses = self.compute_losses(inputs, outputs)
        return outputs, losses

    # ------------------------------------------------------------------ trainer.py:378-442 (pairs mode)
    def predict_poses(self, inputs, features):
        """Both temporally ordered pairs (-1,0), (0,+1) go through the pose encoder in ONE pass: they are stacked
        along the batch and BatchNorm keeps separate statistics per pair (`bn_groups=2`), which is exactly the
        arithmetic of the reference's two sequential passes (trainer.py:398-419) at twice the GEMM N."""
        outputs = {}
        pose_feats = {f: inputs[("color_aug", f, 0)] for f in (-1, 0, 1)}
        pairs = [torch.cat([pose_feats[-1], pose_feats[0]], 1), torch.cat([pose_feats[0], pose_feats[1]], 1)]
        B = pairs[0].shape[0]
        feats = self.models["pose_encoder"](torch.cat(pairs, 0), bn_groups=2)
        axisangle, translation = self.models["pose"]([feats])
        for i, f in enumerate((-1, 1)):
            aa, tr = axisangle[i * B:(i + 1) * B], translation[i * B:(i + 1) * B]
            outputs[("axisangle", 0, f)] = aa
            outputs[("translation", 0, f)] = tr
            outputs[("cam_T_cam", 0, f)] = transformation_from_parameters(aa[:, 0], tr[:, 0], invert=(f < 0))
        return outputs

    # ------------------------------------------------------------------ fused a14 + a15
    def _noise(self, B, nch):
        if not self.opt.cpu_tiebreak_noise or self.opt.disable_automasking:
            return None
        # the reference draws on the CPU generator and copies (trainer.py:594-595)
        return [torch.randn(B, nch, self.opt.height, self.opt.width).to(self.device) for _ in self.opt.scales]

    def fused_losses(self, inputs, outputs, materialize=None):
        o = self.opt
        materialize = o.materialize_logs if materialize is None else materialize
        B = inputs[("color", 0, 0)].shape[0]
        cfg = ops.PhotoConfig(
            inputs[("color", 0, 0)], inputs[("color", -1, 0)], inputs[("color", 1, 0)],
            [inputs[("color", 0, s)] for s in o.scales], inputs[("K", 0)], inputs[("inv_K", 0)],
            noise=self._noise(B, 1 if o.avg_reprojection else 2), min_depth=o.min_depth, max_depth=o.max_depth,
            smoothness=o.disparity_smoothness, disable_automasking=o.disable_automasking,
            avg_reprojection=o.avg_reprojection, no_ssim=o.no_ssim, materialize=materialize,
            rng_seed=self.step * 1000003 + self.rank)
        lv = ops.photometric_loss(cfg, outputs[("cam_T_cam", 0, -1)], outputs[("cam_T_cam", 0, 1)],
                                  [outputs[("disp", s)] for s in o.scales])
        losses = {"loss/{}".format(s): lv[i] for i, s in enumerate(o.scales)}
        losses["loss"] = lv[len(o.scales)]
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

    # ------------------------------------------------------------------ trainer.py:465-515
    def generate_images_pred(self, inputs, outputs):
        o = self.opt
        for scale in o.scales:
            disp = interpolate_bilinear(outputs[("disp", scale)], [o.height, o.width])
            source_scale = 0
            _, depth = disp_to_depth(disp, o.min_depth, o.max_depth)
            outputs[("depth", 0, scale)] = depth
            for frame_id in (-1, 1):
                T = outputs[("cam_T_cam", 0, frame_id)]
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
            disp, color, target = outputs[("disp", scale)], inputs[("color", 0, scale)], inputs[("color", 0, 0)]
            reproj = torch.cat([self.compute_reprojection_loss(outputs[("color", f, scale)], target)
                                for f in (-1, 1)], 1)
            if o.avg_reprojection:
                reproj = reproj.mean(1, keepdim=True)
            if not o.disable_automasking:
                ident = torch.cat([self.compute_reprojection_loss(inputs[("color", f, 0)], target)
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
            loss = to_optimise.mean()
            norm_disp = disp / (disp.mean(2, True).mean(3, True) + 1e-7)
            loss = loss + o.disparity_smoothness * get_smooth_loss(norm_disp, color) / (2 ** scale)
            total_loss = total_loss + loss
            losses["loss/{}".format(scale)] = loss
        losses["loss"] = total_loss / self.num_scales
        return losses

    # ------------------------------------------------------------------ trainer.py:700-763
    def save_model(self, folder):
        """Reference checkpoint layout (trainer.py:711-729): `{name}.pth` = state_dict per model (the encoder's
        additionally carries height / width / use_stereo) + `adam.pth`; options next to them as opt.json."""
        os.makedirs(folder, exist_ok=True)
        for name, model in self.models.items():
            to_save = model.state_dict()
            if name == "encoder":
                to_save["height"], to_save["width"], to_save["use_stereo"] = self.opt.height, self.opt.width, False
            torch.save(to_save, os.path.join(folder, "{}.pth".format(name)))
        torch.save(self.model_optimizer.state_dict(), os.path.join(folder, "adam.pth"))
        with open(os.path.join(folder, "opt.json"), "w") as f:
            json.dump({k: v for k, v in vars(self.opt).items() if isinstance(v, (int, float, str, bool, list))}, f, indent=2)

    def load_model(self, folder, models_to_load=("encoder", "depth", "pose_encoder", "pose")):
        """trainer.py:731-763: keep only the keys the current model has (drops height/width/use_stereo)."""
        for name in models_to_load:
            path = os.path.join(folder, "{}.pth".format(name))
            model_dict = self.models[name].state_dict()
            pretrained = torch.load(path, map_location=self.device)
            model_dict.update({k: v for k, v in pretrained.items() if k in model_dict})
            self.models[name].load_state_dict(model_dict)
        adam = os.path.join(folder, "adam.pth")
        if os.path.isfile(adam):
            self.model_optimizer.load_state_dict(torch.load(adam, map_location=self.device))

    # ------------------------------------------------------------------ trainer.py:233-237
    def train_step(self, inputs):
        outputs, losses = self.process_batch(inputs)
        self.buckets.zero()                     # model_optimizer.zero_grad(set_to_none=True)
        losses["loss"].backward()
        self.buckets.finish()                   # RCCL all-reduce (mean) launched from the backward hooks
        self.model_optimizer.step()
        self.step += 1
        return outputs, losses
