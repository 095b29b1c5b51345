#!/usr/bin/env python3
"""Workload for the HBM-traffic PMC passes: a calibration kernel with a KNOWN dword-access byte count
(dc_disp_to_depth_fwd on 64 Mi floats: reads 256 MiB, writes 512 MiB) followed by the fused photometric
forward + backward at the BASELINE config given by DC_B / DC_H / DC_W / DC_LAYERS (default C2) and two full training steps (Winograd convolution kernels).  Run under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE`
(separate passes); tools/pmc_traffic.sh prints calibrated bytes per launch."""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "self-supervised-depth-estimation_amd"))
sys.path.insert(0, REPO)
from depthcore import ops  # noqa: E402
from depthcore.synthetic import synthetic_batch  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    big = torch.rand(64 << 20, device=dev)
    for _ in range(3):
        ops.disp_to_depth(big, 0.1, 100.0)
    torch.cuda.synchronize()
    B, H, W, NL = (int(os.environ.get(k, d)) for k, d in (("DC_B", 12), ("DC_H", 192), ("DC_W", 640), ("DC_LAYERS", 18)))
    fusion = os.environ.get("DC_FRONT", "") == "fusion"             # BASELINE configs[4] wiring
    dtype = os.environ.get("DC_DTYPE", "f32")                       # networks' matrix-core precision (f32 | bf16)
    inp = synthetic_batch(B, H, W, dev, seed=0, frame_ids=(0, -2, -1, 1) if fusion else (0, -1, 1), packed=True)    # as bench.py
    g = torch.Generator(device=dev).manual_seed(0)
    disps = []
    for s in range(4):
        lo = torch.rand(B, 1, H // 32, W // 32, device=dev, generator=g)
        disps.append(torch.nn.functional.interpolate(lo, size=(H >> s, W >> s), mode="bicubic").clamp(0.01, 0.99)
                     .contiguous().requires_grad_())
    T = []
    for f in range(2):
        t = torch.eye(4, device=dev).repeat(B, 1, 1)
        t[:, :3, 3] = 0.01 * torch.randn(B, 3, device=dev, generator=g)
        T.append(t.requires_grad_())
    cfg = ops.PhotoConfig(inp[("color", 0, 0)], inp[("color", -1, 0)], inp[("color", 1, 0)],
                          [inp[("color", 0, s)] for s in range(4)], inp[("K", 0)], inp[("inv_K", 0)],
                          packed=tuple(inp[("color_packed", f, 0)] for f in (0, -1, 1)))
    for _ in range(5):
        # flush the 256 MiB Infinity Cache between steps so that reads come from HBM, as in a training step
        big.mul_(1.0)
        l = ops.photometric_loss(cfg, T[0], T[1], disps)
        l[4].backward()
    torch.cuda.synchronize()
    # two full training steps: the Winograd convolution kernels (mean HBM bytes per launch over all layers)
    import trainer as T
    kw = dict(fusion="v3", frame_ids=[0, -2, -1, 1]) if fusion else {}
    tr = T.Trainer(T.default_options(batch_size=B, height=H, width=W, num_layers=NL, overlap_streams=False, nets_dtype=dtype, **kw),
                   device=dev)
    tr.set_train()
    for _ in range(2):
        tr.train_step(inp)
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
