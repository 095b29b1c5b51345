#!/usr/bin/env python3
"""dc_conv1x1_fwd / dc_conv1x1_dgrad at the ResNet-50 shapes of BASELINE configs[2] against the tile (DC_G1_TILE, read by
g1_pick in csrc/gemm1x1.hip): checks the tile model's pick against measurement."""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "self-supervised-depth-estimation_amd"))
from depthcore import _lib  # noqa: E402
from depthcore._lib import ptr  # noqa: E402

TILES = ["", "4,4", "2,4", "2,2"]


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
    cases = []
    for B in (8, 16):
        cases += [("l1.conv1", B, 256, 64, 80, 256, 1), ("l1.conv3", B, 64, 256, 80, 256, 1), ("l1.0.conv1", B, 64, 64, 80, 256, 1),
                  ("l2.0.conv1", B, 256, 128, 80, 256, 1), ("l2.0.down", B, 256, 512, 80, 256, 2),
                  ("l2.conv1", B, 512, 128, 40, 128, 1), ("l2.conv3", B, 128, 512, 40, 128, 1),
                  ("l3.0.down", B, 512, 1024, 40, 128, 2), ("l3.conv1", B, 1024, 256, 20, 64, 1),
                  ("l3.conv3", B, 256, 1024, 20, 64, 1), ("l4.0.down", B, 1024, 2048, 20, 64, 2),
                  ("l4.conv1", B, 2048, 512, 10, 32, 1), ("l4.conv3", B, 512, 2048, 10, 32, 1)]
    print("shape | fwd: " + " / ".join(t or "model" for t in TILES) + " | dgrad: " + " / ".join(t or "model" for t in TILES))
    tot = [[0.0] * len(TILES), [0.0] * len(TILES)]
    for name, B, Ci, Co, H, W, s in cases:
        x = torch.randn(B, Ci, H, W, device="cuda")
        w = torch.randn(Co, Ci, 1, 1, device="cuda") * 0.05
        y = torch.empty(B, Co, H // s, W // s, device="cuda")
        gy = torch.randn_like(y)
        dx = torch.empty_like(x)
        st = _lib.stream(x)
        cols = [[], []]
        for i, t in enumerate(TILES):
            if t:
                os.environ["DC_G1_TILE"] = t
            else:
                os.environ.pop("DC_G1_TILE", None)
            a = timed(lambda: L.dc_conv1x1_fwd(ptr(x), ptr(w), ptr(y), B, Ci, Co, H, W, s, st))
            b = timed(lambda: L.dc_conv1x1_dgrad(ptr(gy), ptr(w), ptr(dx), B, Ci, Co, H, W, s, st))
            tot[0][i] += a
            tot[1][i] += b
            cols[0].append("%6.1f" % a)
            cols[1].append("%6.1f" % b)
        os.environ.pop("DC_G1_TILE", None)
        print("%-10s B=%2d %4d->%4d %3dx%3d s%d | " % (name, B, Ci, Co, H, W, s) + " ".join(cols[0]) + " | " + " ".join(cols[1]), flush=True)
    print("sum | " + " ".join("%.0f" % t for t in tot[0]) + " | " + " ".join("%.0f" % t for t in tot[1]))


if __name__ == "__main__":
    main()
