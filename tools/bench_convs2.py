#!/usr/bin/env python3
"""Strided convolutions of the ResNet trunks at BASELINE configs[1] (B = 12 depth / 24 pose, 192x640): depthcore's
implicit-GEMM kernels (dc_convs2_*), per pass, with TFLOP/s (2 MAC of the direct convolution) and the fraction of the fp32
matrix peak (157.3).  --lib adds the library path (aten::convolution / convolution_backward) for comparison."""
import os
import sys

import torch
import torch.nn.functional as F

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "self-supervised-depth-estimation_amd"))
from depthcore import _lib  # noqa: E402
from depthcore._lib import ptr  # noqa: E402


def timed(fn, n=20):
    for _ in range(3):
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
    lib = "--lib" in sys.argv
    only = [a for a in sys.argv[1:] if not a.startswith("--")]
    cases = [("stem.depth", 12, 3, 64, 192, 640, 7), ("stem.pose", 24, 6, 64, 192, 640, 7),
             ("l2.depth", 12, 64, 128, 48, 160, 3), ("l2.pose", 24, 64, 128, 48, 160, 3),
             ("l3.depth", 12, 128, 256, 24, 80, 3), ("l3.pose", 24, 128, 256, 24, 80, 3),
             ("l4.depth", 12, 256, 512, 12, 40, 3), ("l4.pose", 24, 256, 512, 12, 40, 3)]
    tot = 0.0
    for name, B, Ci, Co, H, W, k in cases:
        if only and not any(o in name for o in only):
            continue
        x = torch.randn(B, Ci, H, W, device="cuda")
        w = torch.randn(Co, Ci, k, k, device="cuda") * 0.05
        y = torch.empty(B, Co, H // 2, W // 2, device="cuda")
        gy = torch.randn_like(y)
        dx, dw = torch.empty_like(x), torch.empty_like(w)
        st = _lib.stream(x)
        wsf = torch.empty(max(16, L.dc_convs2_fwd_workspace(B, Ci, Co, H, W, k)), dtype=torch.uint8, device="cuda")
        wsw = torch.empty(max(16, L.dc_convs2_wgrad_workspace(B, Ci, Co, H, W, k)), dtype=torch.uint8, device="cuda")
        t = [timed(lambda: _lib.check(L.dc_convs2_fwd(ptr(x), ptr(w), ptr(y), wsf.data_ptr(), B, Ci, Co, H, W, k, st), "fwd")), 0.0,
             timed(lambda: _lib.check(L.dc_convs2_wgrad(ptr(x), ptr(gy), ptr(dw), wsw.data_ptr(), B, Ci, Co, H, W, k, st), "wgrad"))]
        if k == 3:
            wsd = torch.empty(max(16, L.dc_convs2_dgrad_workspace(B, Ci, Co, H, W, k)), dtype=torch.uint8, device="cuda")
            t[1] = timed(lambda: _lib.check(L.dc_convs2_dgrad(ptr(gy), ptr(w), ptr(dx), wsd.data_ptr(), B, Ci, Co, H, W, k, st), "dgrad"))
        flop = 2.0 * B * Co * Ci * k * k * (H // 2) * (W // 2)
        line = "%-11s B=%2d %3d->%3d @%dx%d k%d  %5.2f GF |" % (name, B, Ci, Co, H, W, k, flop / 1e9)
        for nm, us in zip(("fwd", "dgrad", "wgrad"), t):
            if us:
                line += " %s %6.1f us %5.1f TF (%.2f)" % (nm, us, flop / us / 1e6, flop / us / 1e6 / 157.3)
        if lib:
            lt = [timed(lambda: F.conv2d(x, w, None, 2, k // 2)),
                  timed(lambda: torch.ops.aten.convolution_backward(gy, x, w, None, [2, 2], [k // 2] * 2, [1, 1], False, [0, 0], 1,
                                                                    [k == 3, True, False]))]
            line += " | lib fwd %6.1f bwd %6.1f" % tuple(lt)
        tot += sum(t)
        print(line, flush=True)
    print("sum of depthcore passes: %.1f us" % tot)


if __name__ == "__main__":
    main()
