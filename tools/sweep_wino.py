#!/usr/bin/env python3
"""wino_ps_kernel tile variant sweep: every (MR, NR, reduction split) on the convolution shapes of a training step
(DC_WINO_FORCE, read per launch by csrc/wino.hip).  Prints one line per shape with the time of each variant, so that the
host-side choice in wino_launch can be checked against measurement.   usage: sweep_wino.py [B,Ci,Co,H,W ...]"""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "self-supervised-depth-estimation_amd"))
from depthcore import _lib  # noqa: E402
from depthcore.ops import ptr  # noqa: E402

VARIANTS = ["", "1,2,1", "1,2,2", "2,2,1", "2,2,2", "2,4,1", "2,4,2"]
# C2 (B = 12 depth, 24 pose trunk) : trunk layers, then the depth decoder's upconvs (Ci, Co at the map they run on)
SHAPES = [(12, 64, 64, 48, 160), (24, 64, 64, 48, 160), (12, 128, 128, 24, 80), (24, 128, 128, 24, 80),
          (12, 256, 256, 12, 40), (24, 256, 256, 12, 40), (12, 512, 512, 6, 20), (24, 512, 512, 6, 20),
          (12, 512, 256, 6, 20), (12, 512, 256, 12, 40), (12, 256, 128, 12, 40), (12, 256, 128, 24, 80),
          (12, 128, 64, 24, 80), (12, 128, 64, 48, 160), (12, 64, 32, 48, 160), (12, 96, 32, 96, 320),
          (12, 32, 16, 96, 320), (12, 16, 16, 192, 640),
          # data gradients of the same (roles of Ci / Co swapped)
          (12, 256, 512, 12, 40), (12, 128, 256, 24, 80), (12, 64, 128, 48, 160), (12, 32, 96, 96, 320)]


def timed(fn, iters=60):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    L = _lib.lib()
    shapes = SHAPES
    if len(sys.argv) > 1:
        shapes = [tuple(int(v) for v in s.split(",")) for s in sys.argv[1:]]
    print("shape | " + " | ".join(v or "default" for v in VARIANTS))
    for B, Ci, Co, H, W in shapes:
        x = torch.randn(B, Ci, H, W, device="cuda")
        w = torch.randn(Co, Ci, 3, 3, device="cuda") * 0.05
        y = torch.empty(B, Co, H, W, device="cuda")
        ws = torch.empty(L.dc_wino3x3_workspace(B, Ci, Co, H, W), dtype=torch.uint8, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        out, ref = [], None
        for v in VARIANTS:
            if v:
                os.environ["DC_WINO_FORCE"] = v
            else:
                os.environ.pop("DC_WINO_FORCE", None)
            t = timed(lambda: L.dc_wino3x3_fwd(ptr(x), ptr(w), ptr(y), ws.data_ptr(), B, Ci, Co, H, W, st))
            if ref is None:
                ref = y.clone()
            err = float((y - ref).abs().max() / ref.abs().max())
            out.append("%6.1f%s" % (t, "" if err < 1e-5 else " ERR %.1e" % err))
        os.environ.pop("DC_WINO_FORCE", None)
        print("B=%d %d->%d %dx%d | " % (B, Ci, Co, H, W) + " | ".join(out), flush=True)


if __name__ == "__main__":
    main()
