#!/usr/bin/env python3
"""1x1 convolutions of the ResNet-50 trunks at BASELINE configs[2] (B=8 depth / 16 pose, 320x1024): depthcore's tiled
fp32-MFMA GEMMs (dc_conv1x1_*) vs the library path (aten::convolution / convolution_backward), per pass, with the
fraction of the fp32 matrix peak (157.3 TFLOP/s) and of 8 TB/s HBM each depthcore kernel reaches."""
import os
import sys

import torch
import torch.nn.functional as F

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "self-supervised-depth-estimation_amd"))
from depthcore import _lib  # noqa: E402
from depthcore._lib import ptr  # noqa: E402


def timed(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    L = _lib.lib()
    only_dc = "--dc-only" in sys.argv
    cases = []
    for B in ((8,) if "--b8" in sys.argv else (8, 16)):
        cases += [("l1.conv1", B, 256, 64, 80, 256, 1), ("l1.conv3", B, 64, 256, 80, 256, 1),
                  ("l2.0.conv1", B, 256, 128, 80, 256, 1), ("l2.0.down", B, 256, 512, 80, 256, 2),
                  ("l2.conv1", B, 512, 128, 40, 128, 1), ("l2.conv3", B, 128, 512, 40, 128, 1),
                  ("l3.0.down", B, 512, 1024, 40, 128, 2), ("l3.conv1", B, 1024, 256, 20, 64, 1),
                  ("l3.conv3", B, 256, 1024, 20, 64, 1), ("l4.0.down", B, 1024, 2048, 20, 64, 2),
                  ("l4.conv1", B, 2048, 512, 10, 32, 1), ("l4.conv3", B, 512, 2048, 10, 32, 1)]
    tot_lib = tot_dc = 0.0
    for name, B, Ci, Co, H, W, s in cases:
        x = torch.randn(B, Ci, H, W, device="cuda")
        w = torch.randn(Co, Ci, 1, 1, device="cuda") * 0.05
        y = torch.empty(B, Co, H // s, W // s, device="cuda")
        gy = torch.randn_like(y)
        dx, dw = torch.empty_like(x), torch.empty_like(w)
        ws = torch.empty(max(16, L.dc_conv1x1_wgrad_workspace(B, Ci, Co, H, W, s)), dtype=torch.uint8, device="cuda")
        st = _lib.stream(x)
        m = [timed(lambda: L.dc_conv1x1_fwd(ptr(x), ptr(w), ptr(y), B, Ci, Co, H, W, s, st)),
             timed(lambda: L.dc_conv1x1_dgrad(ptr(gy), ptr(w), ptr(dx), B, Ci, Co, H, W, s, st)),
             timed(lambda: L.dc_conv1x1_wgrad(ptr(x), ptr(gy), ptr(dw), ws.data_ptr(), B, Ci, Co, H, W, s, st))]
        flop = 2.0 * B * Co * Ci * (H // s) * (W // s)
        byts = 4.0 * (x.numel() / (s * s if s == 2 else 1) + y.numel())        # activations each once (weights negligible)
        if only_dc:
            lib = [0.0, 0.0, 0.0]
        else:
            lib = [timed(lambda: F.conv2d(x, w, None, s)),
                   timed(lambda: torch.ops.aten.convolution_backward(gy, x, w, None, [s, s], [0, 0], [1, 1], False, [0, 0], 1,
                                                                     [True, False, False])),
                   timed(lambda: torch.ops.aten.convolution_backward(gy, x, w, None, [s, s], [0, 0], [1, 1], False, [0, 0], 1,
                                                                     [False, True, False]))]
        tot_lib += sum(lib)
        tot_dc += sum(m)
        print("%-10s B=%2d %4d->%4d %3dx%3d s%d | dc fwd %6.1f dgrad %6.1f wgrad %6.1f us (%.0f/%.0f/%.0f %% mfma, %.0f/%.0f %% hbm) | lib %6.1f %6.1f %6.1f"
              % (name, B, Ci, Co, H, W, s, m[0], m[1], m[2], *(100 * flop / (t * 1e-6) / 157.3e12 for t in m),
                 *(100 * byts / (t * 1e-6) / 8e12 for t in m[:2]), lib[0], lib[1], lib[2]), flush=True)
    print("sum over cases: depthcore %.1f us, library %.1f us" % (tot_dc, tot_lib))


if __name__ == "__main__":
    main()
