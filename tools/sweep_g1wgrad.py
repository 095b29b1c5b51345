#!/usr/bin/env python3
"""dc_conv1x1_wgrad at the ResNet-50 shapes of BASELINE configs[2] against the split target, tile and block order
(DC_G1_WBLOCKS, DC_G1_WTILE, DC_G1_WXCD, read by csrc/gemm1x1.hip): one line per shape, one column per choice."""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "self-supervised-depth-estimation_amd"))
from depthcore import _lib  # noqa: E402
from depthcore._lib import ptr  # noqa: E402

CHOICES = [(0, 0, 0), (0, 0, 1), (256, 0, 1), (384, 0, 1), (768, 0, 1), (512, 2, 1), (1024, 2, 1), (512, 4, 1), (256, 4, 1)]      # (blocks, tile, XCD-local splits)


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
    print("shape | " + " | ".join("default" if c == (0, 0, 0) else "%d blocks, %dx%d tile, xcd %d" % (c[0], 32 * c[1], 32 * c[1], c[2]) for c in CHOICES))
    tot = [0.0] * len(CHOICES)
    for name, B, Ci, Co, H, W, s in cases:
        x = torch.randn(B, Ci, H, W, device="cuda")
        gy = torch.randn(B, Co, H // s, W // s, device="cuda")
        dw = torch.empty(Co, Ci, 1, 1, device="cuda")
        st = _lib.stream(x)
        out = []
        for i, (blocks, tile, xcd) in enumerate(CHOICES):
            for k, v in (("DC_G1_WBLOCKS", blocks), ("DC_G1_WTILE", tile), ("DC_G1_WXCD", xcd)):
                if v:
                    os.environ[k] = str(v)
                else:
                    os.environ.pop(k, None)
            ws = torch.empty(max(16, L.dc_conv1x1_wgrad_workspace(B, Ci, Co, H, W, s)), dtype=torch.uint8, device="cuda")
            t = timed(lambda: L.dc_conv1x1_wgrad(ptr(x), ptr(gy), ptr(dw), ws.data_ptr(), B, Ci, Co, H, W, s, st))
            tot[i] += t
            out.append("%6.1f" % t)
        print("%-10s B=%2d %4d->%4d %3dx%3d s%d | " % (name, B, Ci, Co, H, W, s) + " | ".join(out), flush=True)
    print("sum | " + " | ".join("%.0f" % t for t in tot))


if __name__ == "__main__":
    main()
