#!/usr/bin/env python3
"""dc_wino3x3_fwd / dgrad vs the library convolution (MIOpen) at the ResNet-18 trunk shapes of the bench config."""
import os
import sys

import torch
import torch.nn.functional as F

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "self-supervised-depth-estimation_amd"))
from depthcore import _lib  # noqa: E402
from depthcore.ops import ptr  # noqa: E402


def timed(fn, iters=100):
    for _ in range(20):
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
    if os.environ.get("DC_WINO_F4"):                 # A/B: F(4x4,3x3) for the plain trunk convolutions (csrc/wino4.hip)
        L.dc_set_wino_f4(int(os.environ["DC_WINO_F4"]))
    shapes = [(12, 64, 48, 160), (24, 64, 48, 160), (12, 128, 24, 80), (24, 128, 24, 80), (12, 256, 12, 40),
              (24, 256, 12, 40), (12, 512, 6, 20), (24, 512, 6, 20)]
    if len(sys.argv) > 1:
        shapes = [tuple(int(v) for v in s.split(",")) for s in sys.argv[1:]]
    for shp in shapes:
        if len(shp) == 5:           # B, Ci, Co, H, W: forward only (kernel scaling experiments)
            B, Ci, Co, H, W = shp
            x = torch.randn(B, Ci, H, W, device="cuda")
            w = torch.randn(Co, Ci, 3, 3, device="cuda") * 0.05
            y = torch.empty(B, Co, H, W, device="cuda")
            ws = torch.empty(L.dc_wino3x3_workspace(B, Ci, Co, H, W), dtype=torch.uint8, device="cuda")
            st = torch.cuda.current_stream().cuda_stream
            t_f = timed(lambda: L.dc_wino3x3_fwd(ptr(x), ptr(w), ptr(y), ws.data_ptr(), B, Ci, Co, H, W, st))
            t_fl = timed(lambda: F.conv2d(x, w, None, 1, 1))
            print("B=%d %d->%d %dx%d: fwd %.1f us (lib %.1f)" % (B, Ci, Co, H, W, t_f, t_fl), flush=True)
            continue
        B, C, H, W = shp
        x = torch.randn(B, C, H, W, device="cuda")
        w = torch.randn(C, C, 3, 3, device="cuda") * 0.05
        gy = torch.randn(B, C, H, W, device="cuda")
        y = torch.empty_like(x)
        ws = torch.empty(L.dc_wino3x3_workspace(B, C, C, H, W), dtype=torch.uint8, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        t_f = timed(lambda: L.dc_wino3x3_fwd(ptr(x), ptr(w), ptr(y), ws.data_ptr(), B, C, C, H, W, st))
        ref = F.conv2d(x, w, None, 1, 1)
        err_f = float((y - ref).abs().max() / ref.abs().max())
        t_fl = timed(lambda: F.conv2d(x, w, None, 1, 1))
        t_d = timed(lambda: L.dc_wino3x3_dgrad(ptr(gy), ptr(w), ptr(y), ws.data_ptr(), B, C, C, H, W, st))
        refd = torch.ops.aten.convolution_backward(gy, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
                                                   [True, False, False])[0]
        err_d = float((y - refd).abs().max() / refd.abs().max())
        t_dl = timed(lambda: torch.ops.aten.convolution_backward(gy, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0],
                                                                 1, [True, False, False]))
        gflop = 2.0 * B * C * C * 9 * H * W / 1e9
        # weight gradient (kernel + slab reduce) on the same shape
        wws = torch.empty(L.dc_wino3x3_wgrad_workspace(B, C, C, H, W), dtype=torch.uint8, device="cuda")
        dw = torch.empty_like(w)
        t_w = timed(lambda: L.dc_wino3x3_wgrad(ptr(x), ptr(gy), ptr(dw), wws.data_ptr(), B, C, C, H, W, st))
        # issued = the 16 Winograd-domain GEMMs actually sent to the matrix cores = 16/36 of the direct convolution's MACs
        iss = [gflop * 16.0 / 36.0 / t * 1e3 for t in (t_f, t_d, t_w)]
        print("B=%d C=%d %dx%d: fwd %.1f us (lib %.1f) err %.1e | dgrad %.1f us (lib %.1f) err %.1e | wgrad+reduce %.1f us | direct-equiv "
              "%.0f / %.0f / %.0f TFLOP/s | issued %.0f / %.0f / %.0f TFLOP/s = %.2f / %.2f / %.2f of the fp32 matrix peak (157.3)"
              % (B, C, H, W, t_f, t_fl, err_f, t_d, t_dl, err_d, t_w, gflop / t_f * 1e3, gflop / t_d * 1e3, gflop / t_w * 1e3,
                 iss[0], iss[1], iss[2], iss[0] / 157.3, iss[1] / 157.3, iss[2] / 157.3), flush=True)


if __name__ == "__main__":
    main()
