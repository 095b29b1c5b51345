#!/usr/bin/env python3
"""Split-operand 1x1 GEMMs (csrc/gemm1x1_x3.hip: fp32 through three bf16 pieces on the bf16 matrix cores) against the fp32-MFMA
kernels of csrc/gemm1x1.hip on the resnet50 shapes of BASELINE configs[2]: error of both against an fp64 GEMM, time of both.

    python tools/bench_g1x3.py [--batch 8] [--height 320 --width 1024]
"""
import argparse
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "self-supervised-depth-estimation_amd"))
from depthcore import _lib  # noqa: E402
from depthcore._lib import check, ptr, stream  # noqa: E402


def shapes(B, H, W):
    h4, w4 = H // 4, W // 4
    yield "l1.conv1", 256, 64, h4, w4, 1
    yield "l1.conv3", 64, 256, h4, w4, 1
    yield "l2.0.conv1", 256, 128, h4, w4, 1
    yield "l2.0.down", 256, 512, h4, w4, 2
    yield "l2.conv1", 512, 128, h4 // 2, w4 // 2, 1
    yield "l2.conv3", 128, 512, h4 // 2, w4 // 2, 1
    yield "l3.0.down", 512, 1024, h4 // 2, w4 // 2, 2
    yield "l3.conv1", 1024, 256, h4 // 4, w4 // 4, 1
    yield "l3.conv3", 256, 1024, h4 // 4, w4 // 4, 1
    yield "l4.0.down", 1024, 2048, h4 // 4, w4 // 4, 2
    yield "l4.conv1", 2048, 512, h4 // 8, w4 // 8, 1
    yield "l4.conv3", 512, 2048, h4 // 8, w4 // 8, 1


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


def rel(a, ref):
    return float((a.double() - ref).norm() / ref.norm())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--height", type=int, default=320)
    ap.add_argument("--width", type=int, default=1024)
    args = ap.parse_args()
    L = _lib.lib()
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    tot = [0.0, 0.0]
    for name, Ci, Co, Hi, Wi, s in shapes(args.batch, args.height, args.width):
        B = args.batch
        Ho, Wo = Hi // s, Wi // s
        x = torch.relu(torch.randn(B, Ci, Hi, Wi, device=dev, generator=g))
        w = torch.randn(Co, Ci, device=dev, generator=g) * (2.0 / Ci) ** 0.5
        gy = torch.randn(B, Co, Ho, Wo, device=dev, generator=g)
        y0, y1 = torch.empty(B, Co, Ho, Wo, device=dev), torch.empty(B, Co, Ho, Wo, device=dev)
        dx0, dx1 = torch.empty(B, Ci, Hi, Wi, device=dev), torch.empty(B, Ci, Hi, Wi, device=dev)
        ws = torch.empty(L.dc_gemm1x1x3_workspace(Ci, Co), dtype=torch.uint8, device=dev)
        st = stream(x)
        f32 = lambda: check(L.dc_conv1x1_fwd(ptr(x), ptr(w), ptr(y0), B, Ci, Co, Hi, Wi, s, st), "fwd")
        x3 = lambda: check(L.dc_gemm1x1x3_fwd(ptr(x), ptr(w), None, ptr(y1), ws.data_ptr(), B, Ci, Co, Hi, Wi, s, 0, st), "x3 fwd")
        ok = L.dc_gemm1x1x3_fwd_ok(B, Ci, Co, Hi, Wi, s)
        line = "%-10s B=%d %4d->%4d %3dx%-4d s%d |" % (name, B, Ci, Co, Hi, Wi, s)
        if ok:
            t0, t1 = timeit(f32), timeit(x3)
            xs = x[:, :, ::s, ::s].double()
            ref = torch.einsum("mk,bkhw->bmhw", w.double(), xs)
            line += " fwd f32 %6.1f us err %.1e | x3 %6.1f us err %.1e (%.2fx) |" % (t0, rel(y0, ref), t1, rel(y1, ref), t0 / t1)
            tot[0] += t0; tot[1] += t1
            del ref, xs
        if L.dc_gemm1x1x3_dgrad_ok(B, Ci, Co, Hi, Wi, s):
            d32 = lambda: check(L.dc_conv1x1_dgrad(ptr(gy), ptr(w), ptr(dx0), B, Ci, Co, Hi, Wi, s, st), "dgrad")
            d3 = lambda: check(L.dc_gemm1x1x3_dgrad(ptr(gy), ptr(w), ptr(dx1), ws.data_ptr(), None, None, B, Ci, Co, Hi, Wi, s, st), "x3 dgrad")
            t0, t1 = timeit(d32), timeit(d3)
            ref = torch.einsum("mk,bmhw->bkhw", w.double(), gy.double())
            if s == 2:                          # the data gradient of a stride-2 1x1: the values at the even cells, zeros elsewhere
                full = torch.zeros(B, Ci, Hi, Wi, dtype=torch.float64, device=dev)
                full[:, :, ::2, ::2] = ref
                ref = full
            line += " dgrad f32 %6.1f us err %.1e | x3 %6.1f us err %.1e (%.2fx)" % (t0, rel(dx0, ref), t1, rel(dx1, ref), t0 / t1)
            tot[0] += t0; tot[1] += t1
            del ref
        if L.dc_gemm1x1x3_wgrad_ok(B, Ci, Co, Hi, Wi, s):
            dw0, dw1 = torch.empty(Co, Ci, device=dev), torch.empty(Co, Ci, device=dev)
            w0 = torch.empty(max(16, L.dc_conv1x1_wgrad_workspace(B, Ci, Co, Hi, Wi, s)), dtype=torch.uint8, device=dev)
            w1 = torch.empty(max(16, L.dc_gemm1x1x3_wgrad_workspace(B, Ci, Co, Hi, Wi, s)), dtype=torch.uint8, device=dev)
            g32 = lambda: check(L.dc_conv1x1_wgrad(ptr(x), ptr(gy), ptr(dw0), w0.data_ptr(), B, Ci, Co, Hi, Wi, s, st), "wgrad")
            g3 = lambda: check(L.dc_gemm1x1x3_wgrad(ptr(x), ptr(gy), ptr(dw1), w1.data_ptr(), B, Ci, Co, Hi, Wi, s, st), "x3 wgrad")
            t0, t1 = timeit(g32), timeit(g3)
            ref = torch.einsum("bmhw,bkhw->mk", gy.double(), x[:, :, ::s, ::s].double())
            line += " | wgrad f32 %6.1f us err %.1e | x3 %6.1f us err %.1e (%.2fx)" % (t0, rel(dw0, ref), t1, rel(dw1, ref), t0 / t1)
            tot[0] += t0; tot[1] += t1
            del ref
        print(line, flush=True)
    print("sum: fp32-MFMA %.1f us, split %.1f us (%.2fx)" % (tot[0], tot[1], tot[0] / max(tot[1], 1e-9)))


if __name__ == "__main__":
    main()
