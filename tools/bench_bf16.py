#!/usr/bin/env python3
"""Per-layer time of the 3x3 convolutions of one network pass, fp32 kernels (Winograd / direct) next to the bf16 matrix-core
kernels (csrc/conv_bf16.hip):  python tools/bench_bf16.py [--batch 36] [--only dec|enc]

Prints forward / data gradient / weight gradient in microseconds and the HBM rate of the bf16 kernels (each fp32 operand and
the result once)."""
import argparse
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "self-supervised-depth-estimation_amd"))
from depthcore import ops  # noqa: E402

# name, C0, up0, C1, Co, H, W, act, pad   (192 x 640 input)
DEC = [("upconv_4_0", 512, 0, 0, 256, 6, 20, 1, 0), ("upconv_4_1", 256, 1, 256, 256, 12, 40, 1, 0),
       ("upconv_3_0", 256, 0, 0, 128, 12, 40, 1, 0), ("upconv_3_1", 128, 1, 128, 128, 24, 80, 1, 0),
       ("upconv_2_0", 128, 0, 0, 64, 24, 80, 1, 0), ("upconv_2_1", 64, 1, 64, 64, 48, 160, 1, 0),
       ("upconv_1_0", 64, 0, 0, 32, 48, 160, 1, 0), ("upconv_1_1", 32, 1, 64, 32, 96, 320, 1, 0),
       ("upconv_0_0", 32, 0, 0, 16, 96, 320, 1, 0), ("upconv_0_1", 16, 1, 0, 16, 192, 640, 1, 0)]
ENC = [("layer1", 64, 0, 0, 64, 48, 160, 0, 1), ("layer2", 128, 0, 0, 128, 24, 80, 0, 1),
       ("layer3", 256, 0, 0, 256, 12, 40, 0, 1), ("layer4", 512, 0, 0, 512, 6, 20, 0, 1)]


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
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=36)
    ap.add_argument("--only", default="")
    ap.add_argument("--layer", default="", help="comma-separated layer names")
    ap.add_argument("--prec", default="f32,bf16")
    a = ap.parse_args()
    B = a.batch
    layers = (DEC if a.only != "enc" else []) + (ENC if a.only != "dec" else [])
    if a.layer:
        layers = [l for l in layers if l[0] in a.layer.split(",")]
    tot = {"f32": 0.0, "bf16": 0.0}
    for name, C0, up0, C1, Co, H, W, act, pad in layers:
        h0, w0 = (H // 2, W // 2) if up0 else (H, W)
        gy = torch.randn(B, Co, H, W, device="cuda")
        line = "%-11s %3d%s+%3d -> %3d @%3dx%3d:" % (name, C0, "^" if up0 else " ", C1, Co, H, W)
        nbytes = 4.0 * (B * C0 * h0 * w0 + B * C1 * H * W + B * Co * H * W)
        for prec in a.prec.split(","):
            res = []
            for need_x, need_w in ((True, False), (False, True)):
                x0 = torch.randn(B, C0, h0, w0, device="cuda", requires_grad=need_x)
                x1 = torch.randn(B, C1, H, W, device="cuda", requires_grad=need_x) if C1 else None
                w = (torch.randn(Co, C0 + C1, 3, 3, device="cuda") * 0.05).requires_grad_(need_w)
                b = torch.zeros(Co, device="cuda", requires_grad=need_w) if act else None

                def fwd():
                    with ops.matrix_precision(prec):
                        return ops.conv3x3_block(x0, x1, w, b, up0=bool(up0), act=act, pad=pad)
                y = fwd()
                if need_x:
                    res.append(timed(fwd))
                res.append(timed(lambda: y.backward(gy, retain_graph=True)))
            f, dx, dw = res
            tot[prec] += f + dx + dw
            line += "  %s fwd %6.1f dx %6.1f dw %6.1f" % (prec, f, dx, dw)
            if prec == "bf16":
                line += "  (%.2f / %.2f / %.2f TB/s)" % (nbytes / f / 1e6, nbytes / dx / 1e6, nbytes / dw / 1e6)
        print(line, flush=True)
    print("total us: f32 %.0f, bf16 %.0f (B=%d)" % (tot["f32"], tot["bf16"], B))


if __name__ == "__main__":
    main()
