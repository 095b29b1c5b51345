"""One full CPU training step of the oracle (TEST / BASELINE INFRASTRUCTURE, see oracle/__init__.py).

process_batch (depth encoder on frame 0, depth decoder, pose encoder+decoder on the pairs (-1,0),(0,+1),
8 warps, 16 reprojection terms, 4 smoothness terms) -> backward -> Adam, i.e. reference
trainer.py:233-237 + 256-622 in the vanilla Monodepth2 wiring, on functional networks
over plain state dicts.  Used as the checker of the full-step parity test and as bench.py's
`cpu_baseline` ("port").
"""
import torch

from . import fusion_ref as FR
from . import gru_ref as GR
from . import ref_cpu as R
from .kinks import ForcedKinks
from .resnet_ref import resnet_encoder_forward


class CpuTrainer:
    def __init__(self, state, opt=None, num_layers=18, lr=1e-4):
        """state: {"encoder": sd, "depth": sd, "pose_encoder": sd, "pose": sd[, "fusion": sd]} of CPU tensors (fp32, or
        fp64 for conditioning checks).  With a "fusion" entry the front-end is trainer_fusion_v3.py:311-330."""
        self.opt = opt or R.Opt()
        self.num_layers = num_layers
        self.state = {k: {n: t.detach().clone() for n, t in sd.items()} for k, sd in state.items()}
        self.params = []
        for k, sd in self.state.items():
            for n, t in sd.items():
                if t.is_floating_point() and "running_" not in n:
                    t.requires_grad_()
                    self.params.append(t)
        self.optim = torch.optim.Adam(self.params, lr)
        self.num_ch_enc = [64, 64, 128, 256, 512] if num_layers <= 34 else [64, 256, 512, 1024, 2048]

    def process_batch(self, inputs, noise, kinks=None, nets_autocast=None):
        """`kinks`: {"encoder" | "pose_encoder" | "pose": tape entries recorded by the HIP path (tests/kink_tape.py)} --
        the ReLU / max-pool decisions of those networks are then imposed (oracle/kinks.py); the imposed-vs-own disagreements
        are left in `self.kink_report`.  The HIP path stacks both pose pairs along the batch: pair i = rows [i*B, (i+1)*B).
        `nets_autocast`: a torch dtype -- the networks are evaluated under torch's own CPU autocast of that type and their
        outputs cast back to fp32 for the loss: torch's statement of "reduced-precision networks, fp32 loss" (BASELINE
        configs[4]), the yardstick of tests/test_bf16_policy_gpu.py."""
        import contextlib
        o = self.opt
        with (torch.autocast("cpu", dtype=nets_autocast) if nets_autocast is not None else contextlib.nullcontext()):
            outputs = self._networks(inputs, kinks)
        if nets_autocast is not None:
            outputs = {k: (v.float() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in outputs.items()}
        R.generate_images_pred(inputs, outputs, o)
        losses = R.compute_losses(inputs, outputs, o, noise)
        for _, k in self._used:
            k.done()
        self.kink_report = [(name,) + d for name, k in self._used for d in k.disagree]
        self.kink_objs = self._used
        return outputs, losses

    def _networks(self, inputs, kinks):
        o = self.opt
        used = self._used = []

        def forced(name, part=None, parts=1):
            if not kinks or name not in kinks:
                return None
            n = kinks[name][0][1].shape[0] // parts
            k = ForcedKinks(kinks[name], None if part is None else slice(part * n, (part + 1) * n), keep_pre=True)
            used.append((name, k))
            return k
        enc_k = forced("encoder")
        if "gru" in self.state:                      # trainer_gru.py:595-644: `inputs` = the sequence stacked along the batch
            feats = resnet_encoder_forward(self.state["encoder"], inputs[("color", 0, 0)], self.num_layers, kinks=enc_k)
            feats = GR.run_gru_v5(feats, self.state["gru"])
            outputs = R.depth_decoder_forward(self.state["depth"], feats, self.num_ch_enc, tuple(o.scales))
        elif "fusion" in self.state:
            enc_input = torch.cat([inputs[("color_aug", i, 0)] for i in (-2, -1, 0)], 0)
            feats = resnet_encoder_forward(self.state["encoder"], enc_input, self.num_layers, kinks=enc_k)
            dec = R.depth_decoder_forward(self.state["depth"], feats, self.num_ch_enc, tuple(o.scales))
            outputs = FR.fusion_v3_forward(self.state["fusion"], dec)
        else:
            feats = resnet_encoder_forward(self.state["encoder"], inputs[("color_aug", 0, 0)], self.num_layers, kinks=enc_k)
            outputs = R.depth_decoder_forward(self.state["depth"], feats, self.num_ch_enc, tuple(o.scales))
        pair = [0, 0]           # R.predict_poses runs the encoder and the decoder once per pair, (-1, 0) first

        def pose_enc(x):
            k = forced("pose_encoder", pair[0], 2)
            pair[0] += 1
            return resnet_encoder_forward(self.state["pose_encoder"], x, self.num_layers, kinks=k)

        def pose_dec(f):
            k = forced("pose", pair[1], 2)
            pair[1] += 1
            return R.pose_decoder_forward(self.state["pose"], f, 2, kinks=k)
        outputs.update(R.predict_poses(inputs, pose_enc, pose_dec))
        return outputs

    def train_step(self, inputs, noise):
        outputs, losses = self.process_batch(inputs, noise)
        self.optim.zero_grad()
        losses["loss"].backward()
        self.optim.step()
        return outputs, losses
