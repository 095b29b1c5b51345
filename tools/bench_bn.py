#!/usr/bin/env python3
"""dc_bn_relu_fwd / dc_bn_relu_bwd at the trunk shapes of a training step: time and bytes moved per second
(forward: x read twice + y written [+ residual read]; backward: gy and x read twice, dx written [+ dres written])."""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "self-supervised-depth-estimation_amd"))
from depthcore import _lib  # noqa: E402
from depthcore._lib import ptr, stream, check  # noqa: E402


def timeit(fn, n=100):
    for _ in range(10):
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
    shapes = [(12, 64, 96, 320, 0), (24, 64, 96, 320, 0), (12, 64, 48, 160, 1), (24, 64, 48, 160, 1), (24, 128, 24, 80, 1),
              (24, 256, 12, 40, 1), (24, 512, 6, 20, 1), (8, 256, 80, 256, 1), (8, 64, 80, 256, 0), (8, 512, 40, 128, 1)]
    for (N, C, H, W, with_res) in shapes:
        HW = H * W
        x = torch.randn(N, C, H, W, device=dev)
        res = torch.randn_like(x) if with_res else None
        gy = torch.randn_like(x)
        y, dx, dres = torch.empty_like(x), torch.empty_like(x), (torch.empty_like(x) if with_res else None)
        g, b = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
        sm, si = torch.empty(C, device=dev), torch.empty(C, device=dev)
        rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        dg, db = torch.empty(C, device=dev), torch.empty(C, device=dev)
        ws = torch.empty(L.dc_bn_workspace(N, C, HW), dtype=torch.uint8, device=dev)
        mask = torch.empty(max(1, L.dc_bn_mask_bytes(N, C, HW)), dtype=torch.uint8, device=dev)

        def fwd():
            check(L.dc_bn_relu_fwd(ptr(x), ptr(res) if with_res else None, ptr(g), ptr(b), ptr(y), ptr(sm), ptr(si), ptr(rm), ptr(rv),
                                   ws.data_ptr(), mask.data_ptr(), N, C, HW, 1e-5, 0.1, 1, 1, stream()), "fwd")

        def bwd():
            check(L.dc_bn_relu_bwd(ptr(x), ptr(y), ptr(gy), ptr(g), ptr(sm), ptr(si), ptr(dx), ptr(dres) if with_res else None, ptr(dg),
                                   ptr(db), ws.data_ptr(), mask.data_ptr(), N, C, HW, 1, 1, stream()), "bwd")
        tf, tb = timeit(fwd), timeit(bwd)
        nb = x.numel() * 4
        fb, bb = nb * (3 + with_res), nb * (5 + with_res)
        print("N=%d C=%d %dx%d res=%d (%.0f MB): fwd %.1f us %.2f TB/s | bwd %.1f us %.2f TB/s" % (N, C, H, W, with_res, nb / 1e6, tf, fb / tf / 1e6,
                                                                                               tb, bb / tb / 1e6), flush=True)


if __name__ == "__main__":
    main()
