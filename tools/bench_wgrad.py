#!/usr/bin/env python3
"""Weight-gradient of 3x3 stride-1 zero-pad convs at the ResNet-18 encoder shapes: depthcore's Winograd-domain
dc_wino3x3_wgrad and direct split-K dc_conv3x3_bwd (weight only) vs MIOpen (aten.convolution_backward, weight only)."""
import ctypes
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "self-supervised-depth-estimation_amd"))
from depthcore import _lib  # noqa: E402
from depthcore._lib import ptr, stream, check  # noqa: E402


def timeit(fn, n=100):
    for _ in range(20):
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
    dev = torch.device("cuda:0")
    for (B, C, H, W) in [(12, 64, 48, 160), (24, 64, 48, 160), (12, 128, 24, 80), (24, 128, 24, 80), (12, 256, 12, 40),
                         (24, 256, 12, 40), (12, 512, 6, 20), (24, 512, 6, 20)]:
        x = torch.randn(B, C, H, W, device=dev)
        w = torch.randn(C, C, 3, 3, device=dev) * 0.05
        gy = torch.randn(B, C, H, W, device=dev)
        y = torch.empty_like(gy)
        dw = torch.empty_like(w)
        ws = torch.empty(L.dc_conv3x3_bwd_workspace(C, 0, B, C, H, W), dtype=torch.uint8, device=dev)

        def mine():
            check(L.dc_conv3x3_bwd(ptr(x), C, 0, None, 0, ptr(w), ptr(y), ptr(gy), None, None, ptr(dw), None, ws.data_ptr(),
                                   B, C, H, W, 0, 1, stream()), "bwd")

        def miopen():
            return torch.ops.aten.convolution_backward(gy, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
                                                       [False, True, False])[1]
        wws = torch.empty(L.dc_wino3x3_wgrad_workspace(B, C, C, H, W), dtype=torch.uint8, device=dev)
        dww = torch.empty_like(w)

        def wino():
            check(L.dc_wino3x3_wgrad(ptr(x), ptr(gy), ptr(dww), wws.data_ptr(), B, C, C, H, W, stream()), "wino wgrad")

        ref = miopen()
        mine()
        wino()
        err = float((dw - ref).abs().max() / ref.abs().max())
        errw = float((dww - ref).abs().max() / ref.abs().max())
        print("B=%d C=%d %dx%d: wino wgrad %.1f us (err %.1e, ws %.0f MB) | direct wgrad %.1f us (err %.1e) | MIOpen %.1f us" % (
            B, C, H, W, timeit(wino), errw, wws.numel() / 1e6, timeit(mine), err, timeit(miopen)), flush=True)


if __name__ == "__main__":
    main()
