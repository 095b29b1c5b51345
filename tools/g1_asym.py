#!/usr/bin/env python3
"""Asymptotic efficiency of the 1x1 GEMM kernels: large shapes (many chunks per block, many rounds of blocks) per tile choice
(DC_G1_TILE), against the step's 5.4 GFLOP shapes -- separates the inner loop from prologue / epilogue / round quantisation."""
import os
import sys

import torch

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
    cases = [("big", 8, 1024, 1024, 40, 128), ("bigK", 8, 2048, 512, 40, 128), ("l3.conv1", 8, 1024, 256, 20, 64), ("l3.conv3", 8, 256, 1024, 20, 64),
             ("l2.conv3", 8, 128, 512, 40, 128), ("l1.conv3", 8, 64, 256, 80, 256)]
    for name, B, Ci, Co, H, W in cases:
        x = torch.randn(B, Ci, H, W, device="cuda")
        w = torch.randn(Co, Ci, 1, 1, device="cuda") * 0.05
        y = torch.empty(B, Co, H, W, device="cuda")
        gy = torch.randn_like(y)
        dx = torch.empty_like(x)
        st = _lib.stream(x)
        flop = 2.0 * B * H * W * Ci * Co
        line = "%-9s %4d->%4d N=%6d %6.1f GF |" % (name, Ci, Co, B * H * W, flop / 1e9)
        for tile in ("4,4", "2,4", "2,2"):
            os.environ["DC_G1_TILE"] = tile
            tf = timed(lambda: _lib.check(L.dc_conv1x1_bias_act_fwd(ptr(x), ptr(w), None, ptr(y), B, Ci, Co, H, W, 1, 0, st), "fwd"))
            td = timed(lambda: _lib.check(L.dc_conv1x1_dgrad_add(ptr(gy), ptr(w), ptr(dx), None, B, Ci, Co, H, W, 1, st), "dgrad"))
            line += " %s: fwd %7.1f us %.2f dgrad %7.1f us %.2f |" % (tile, tf, flop / tf / 1e6 / 157.3, td, flop / td / 1e6 / 157.3)
        os.environ.pop("DC_G1_TILE", None)
        print(line, flush=True)


if __name__ == "__main__":
    main()
