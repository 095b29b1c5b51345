#!/usr/bin/env python3
"""Classic vs persistent wino_ps launches (dc_set_wino_persist) on the step's multi-round trunk shapes: forward and data
gradient, us per launch, same process, alternating."""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "self-supervised-depth-estimation_amd"))
from depthcore import _lib  # noqa: E402
from depthcore._lib import ptr  # noqa: E402


def timed(fn, n=40):
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
    shapes = [(12, 64, 64, 48, 160), (24, 64, 64, 48, 160), (12, 128, 128, 24, 80), (24, 128, 128, 24, 80), (24, 256, 256, 12, 40),
              (8, 64, 64, 80, 256), (16, 64, 64, 80, 256), (8, 128, 128, 40, 128), (16, 128, 128, 40, 128), (36, 64, 64, 48, 160)]
    tot = [0.0, 0.0]
    for B, Ci, Co, H, W in shapes:
        x = torch.randn(B, Ci, H, W, device="cuda")
        w = torch.randn(Co, Ci, 3, 3, device="cuda") * 0.05
        y = torch.empty(B, Co, H, W, device="cuda")
        ws = torch.empty(L.dc_wino3x3_workspace(B, Ci, Co, H, W), dtype=torch.uint8, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        r = []
        for rep in range(2):
            for mode in (0, 1):
                L.dc_set_wino_persist(mode)
                tf = timed(lambda: L.dc_wino3x3_fwd(ptr(x), ptr(w), ptr(y), ws.data_ptr(), B, Ci, Co, H, W, st))
                td = timed(lambda: L.dc_wino3x3_dgrad(ptr(x), ptr(w), ptr(y), ws.data_ptr(), B, Ci, Ci, H, W, st)) if Ci == Co else 0.0
                r.append((tf, td))
        L.dc_set_wino_persist(0)
        c = [min(r[0][0], r[2][0]), min(r[0][1], r[2][1])]
        p = [min(r[1][0], r[3][0]), min(r[1][1], r[3][1])]
        tot[0] += sum(c); tot[1] += sum(p)
        print("B=%2d %3d->%3d %3dx%3d | classic fwd %6.1f dgrad %6.1f | persistent fwd %6.1f dgrad %6.1f | %+5.1f %% / %+5.1f %%"
              % (B, Ci, Co, H, W, c[0], c[1], p[0], p[1], 100 * (p[0] / c[0] - 1), 100 * (p[1] / max(c[1], 1e-9) - 1)), flush=True)
    print("sum: classic %.0f us, persistent %.0f us (%+.1f %%)" % (tot[0], tot[1], 100 * (tot[1] / tot[0] - 1)))


if __name__ == "__main__":
    main()
