#!/usr/bin/env python3
"""Bias of the split-operand weight gradient (csrc/gemm1x1_x3.hip x3_chunk): error of dc_gemm1x1x3_wgrad and of the fp32-MFMA
dc_conv1x1_wgrad against an fp64 GEMM on resnet50's layer1 conv1 shape, for B = 1, 2, 8 (the reduction length), with fp32 inputs, with
inputs that are exact in bf16 (only the a0 b0 product is non-zero: no inexact partial sums) and with an all-positive gradient; and the
MEAN of the signed error over its rms.  A matrix instruction that truncated toward -inf shows as a mean/rms near -1 growing with B
(measured before the chunk totals got alternating signs: -0.81, -0.88, -0.96; fp32-MFMA: 0.00).

    python tools/diag_x3_bias.py
"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "self-supervised-depth-estimation_amd"))
from depthcore import _lib
from depthcore._lib import check, ptr, stream
L = _lib.lib(); dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
def rel(a, ref): return float((a.double() - ref).norm() / ref.norm())
Ci, Co, Hi, Wi = 256, 64, 80, 256
for B in (1, 2, 8):
    for kind in ("f32", "bf16in", "pos"):
        x = torch.relu(torch.randn(B, Ci, Hi, Wi, device=dev, generator=g)); gy = torch.randn(B, Co, Hi, Wi, device=dev, generator=g)
        if kind == "bf16in": x = x.bfloat16().float(); gy = gy.bfloat16().float()
        if kind == "pos": gy = gy.abs()
        dw0, dw1 = torch.empty(Co, Ci, device=dev), torch.empty(Co, Ci, device=dev)
        w0 = torch.empty(max(16, L.dc_conv1x1_wgrad_workspace(B, Ci, Co, Hi, Wi, 1)), dtype=torch.uint8, device=dev)
        w1 = torch.empty(max(16, L.dc_gemm1x1x3_wgrad_workspace(B, Ci, Co, Hi, Wi, 1)), dtype=torch.uint8, device=dev)
        st = stream(x)
        check(L.dc_conv1x1_wgrad(ptr(x), ptr(gy), ptr(dw0), w0.data_ptr(), B, Ci, Co, Hi, Wi, 1, st), "wgrad")
        check(L.dc_gemm1x1x3_wgrad(ptr(x), ptr(gy), ptr(dw1), w1.data_ptr(), B, Ci, Co, Hi, Wi, 1, st), "x3 wgrad")
        ref = torch.einsum("bmhw,bkhw->mk", gy.double(), x.double())
        d1 = (dw1.double() - ref); d0 = (dw0.double() - ref)
        print("B=%d %-6s f32 %.2e x3 %.2e | mean signed err / rms: f32 %+.2f x3 %+.2f | slab bytes %d" % (B, kind, rel(dw0, ref), rel(dw1, ref),
              float(d0.mean() / d0.pow(2).mean().sqrt()), float(d1.mean() / d1.pow(2).mean().sqrt()), w1.numel()))
