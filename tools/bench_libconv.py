#!/usr/bin/env python3
"""Library (MIOpen) times of the trunk convolutions that are NOT on depthcore kernels: 7x7/2 stem, stride-2 3x3,
1x1/2 downsample.  fwd / dgrad / wgrad separately."""
import os
import sys

import torch
import torch.nn.functional as F

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "self-supervised-depth-estimation_amd"))
from depthcore import ops  # noqa: E402


def timed(fn, n=50):
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
    cases = []
    for B in (12, 24):
        cin0 = 3 if B == 12 else 6
        cases += [("stem7x7s2", B, cin0, 64, 192, 640, 7, 2, 3, False),
                  ("l2.0.conv1", B, 64, 128, 48, 160, 3, 2, 1, True), ("l2.0.down", B, 64, 128, 48, 160, 1, 2, 0, True),
                  ("l3.0.conv1", B, 128, 256, 24, 80, 3, 2, 1, True), ("l3.0.down", B, 128, 256, 24, 80, 1, 2, 0, True),
                  ("l4.0.conv1", B, 256, 512, 12, 40, 3, 2, 1, True), ("l4.0.down", B, 256, 512, 12, 40, 1, 2, 0, True)]
    tot = 0.0
    for name, B, Ci, Co, H, W, k, s, p, need_dx in cases:
        x = torch.randn(B, Ci, H, W, device="cuda")
        w = torch.randn(Co, Ci, k, k, device="cuda") * 0.05
        y = F.conv2d(x, w, None, s, p)
        gy = torch.randn_like(y)
        t_f = timed(lambda: F.conv2d(x, w, None, s, p))
        t_d = timed(lambda: torch.ops.aten.convolution_backward(gy, x, w, None, [s, s], [p, p], [1, 1], False, [0, 0], 1,
                                                                [True, False, False])) if need_dx else 0.0
        t_w = timed(lambda: torch.ops.aten.convolution_backward(gy, x, w, None, [s, s], [p, p], [1, 1], False, [0, 0], 1,
                                                                [False, True, False]))
        tot += t_f + t_d + t_w
        gmac = B * Co * Ci * k * k * y.shape[2] * y.shape[3] / 1e9
        mine = ""
        if k == 1:
            xg, wg = x.clone().requires_grad_(True), w.clone().requires_grad_(False)
            yg = ops.conv1x1(xg, wg, s)
            m_f = timed(lambda: ops.conv1x1(x, w, s))
            m_d = timed(lambda: yg.backward(gy, retain_graph=True))
            xg2, wg2 = x.clone().requires_grad_(False), w.clone().requires_grad_(True)
            yg2 = ops.conv1x1(xg2, wg2, s)
            m_w = timed(lambda: yg2.backward(gy, retain_graph=True))
            mine = "  | depthcore fwd %5.1f dgrad %5.1f wgrad %5.1f" % (m_f, m_d, m_w)
        print("%-11s B=%2d %3d->%3d %3dx%3d k%d s%d: fwd %6.1f  dgrad %6.1f  wgrad %6.1f us  (%.2f GMAC)%s"
              % (name, B, Ci, Co, H, W, k, s, t_f, t_d, t_w, gmac, mine), flush=True)
    print("total %.1f us per step" % tot)


if __name__ == "__main__":
    main()
