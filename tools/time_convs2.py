#!/usr/bin/env python3
"""3x3 / 2 trunk convolutions: the split-operand kernels (gemm1x1_x3.hip, S = 3 loaders) against the fp32-MFMA kernels (convgemm.hip
cg_fwd3 / cg_wgrad3) -- time of the forward and of the weight gradient per shape, dc_set_gemm_split(0) against (3).

    python tools/time_convs2.py [--config c2|c3]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "self-supervised-depth-estimation_amd"))
from depthcore import _lib  # noqa: E402
from depthcore._lib import check, ptr, stream  # noqa: E402

SHAPES = {"c2": [(12, 64, 128, 48, 160), (12, 128, 256, 24, 80), (12, 256, 512, 12, 40)],
          "c3": [(8, 128, 128, 80, 256), (8, 256, 256, 40, 128), (8, 512, 512, 20, 64)]}


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c2")
    args = ap.parse_args()
    L = _lib.lib()
    dev = torch.device("cuda:0")
    prev = L.dc_get_gemm_split()
    try:
        for B, Ci, Co, Hi, Wi in SHAPES[args.config]:
            x = torch.relu(torch.randn(B, Ci, Hi, Wi, device=dev)); w = torch.randn(Co, Ci, 3, 3, device=dev) * (2.0 / (9 * Ci)) ** 0.5
            gy = torch.randn(B, Co, Hi // 2, Wi // 2, device=dev); y = torch.empty_like(gy); dw = torch.empty_like(w)
            ws = torch.empty(max(16, L.dc_convs2_fwd_workspace(B, Ci, Co, Hi, Wi, 3), L.dc_convs2_wgrad_workspace(B, Ci, Co, Hi, Wi, 3)),
                             dtype=torch.uint8, device=dev)
            st = stream(x)
            t = {}
            for mode in (0, 3):
                L.dc_set_gemm_split(mode)
                t["f%d" % mode] = timeit(lambda: check(L.dc_convs2_fwd(ptr(x), ptr(w), ptr(y), ws.data_ptr(), B, Ci, Co, Hi, Wi, 3, st), "fwd"))
                t["w%d" % mode] = timeit(lambda: check(L.dc_convs2_wgrad(ptr(x), ptr(gy), ptr(dw), ws.data_ptr(), B, Ci, Co, Hi, Wi, 3, st), "wgrad"))
            print("B=%d %4d->%4d %3dx%-4d | fwd f32 %6.1f us, split %6.1f us (%.2fx) | wgrad f32 %6.1f us, split %6.1f us (%.2fx)"
                  % (B, Ci, Co, Hi, Wi, t["f0"], t["f3"], t["f0"] / t["f3"], t["w0"], t["w3"], t["w0"] / t["w3"]), flush=True)
    finally:
        L.dc_set_gemm_split(prev)


if __name__ == "__main__":
    main()
