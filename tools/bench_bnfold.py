#!/usr/bin/env python3
"""Cost of every piece of the BatchNorm fold against the stand-alone passes it replaces, per trunk shape (kernel time on the
GPU: 50 launches captured in a hipGraph and replayed, so the host is not in the measurement).
usage: bench_bnfold.py [B,C,H,W ...]   (3x3 C -> C on the Winograd kernels and 1x1 C -> 4C / 4C -> C on the tiled GEMMs)"""
import ctypes
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "self-supervised-depth-estimation_amd"))
from depthcore import _lib  # noqa: E402
from depthcore._lib import BnFold  # noqa: E402
from depthcore.ops import ptr  # noqa: E402

N = 50


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.graph(g, stream=s):
        for _ in range(N):
            fn()
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (4 * N) * 1e3


def st():
    return torch.cuda.current_stream().cuda_stream


def main():
    L = _lib.lib()
    shapes = [(12, 64, 48, 160), (24, 64, 48, 160), (12, 128, 24, 80), (24, 128, 24, 80), (8, 64, 80, 256), (8, 128, 40, 128)]
    if len(sys.argv) > 1:
        shapes = [tuple(int(v) for v in s.split(",")) for s in sys.argv[1:]]
    dev = "cuda"
    for B, C, H, W in shapes:
        groups = 1
        x = torch.randn(B, C, H, W, device=dev)
        gy = torch.randn(B, C, H, W, device=dev)
        y = torch.empty_like(x)
        y2 = torch.empty_like(x)
        add = torch.randn(B, C, H, W, device=dev)
        w = torch.randn(C, C, 3, 3, device=dev) * 0.05
        dw = torch.empty_like(w)
        gamma, beta = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
        rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        tab = torch.empty(4, C, device=dev)
        ws = torch.empty(L.dc_wino3x3_workspace(B, C, C, H, W), dtype=torch.uint8, device=dev)
        wws = torch.empty(L.dc_wino3x3_wgrad_workspace(B, C, C, H, W), dtype=torch.uint8, device=dev)
        bws = torch.empty(L.dc_bn_workspace(B, C, H * W), dtype=torch.uint8, device=dev)
        mask = torch.empty(L.dc_bn_mask_bytes(B, C, H * W), dtype=torch.uint8, device=dev)
        mean, invstd = torch.empty(C, device=dev), torch.empty(C, device=dev)
        dg, db = torch.empty(C, device=dev), torch.empty(C, device=dev)
        coef = torch.empty(4 * C, device=dev)
        ppg = ctypes.c_int(0)
        np_f = L.dc_wino3x3_stat_parts(B, C, C, H, W, groups, ctypes.byref(ppg)); ppg_f = ppg.value
        np_b = L.dc_wino3x3_bwd_parts(B, C, C, H, W, groups, ctypes.byref(ppg)); ppg_b = ppg.value
        np_s = L.dc_bn_stat_parts(B, C, H * W, groups, ctypes.byref(ppg)); ppg_s = ppg.value
        part = torch.empty(max(np_f, np_b, np_s, 1) * C * 2, device=dev)
        cnt = float(B * H * W)
        r = {}
        # ---- stand-alone BatchNorm passes (the old chain)
        r["bn fwd (stats+apply)"] = timed(lambda: L.dc_bn_relu_fwd(ptr(x), None, ptr(gamma), ptr(beta), ptr(y), ptr(mean), ptr(invstd), ptr(rm), ptr(rv),
                                                                  bws.data_ptr(), mask.data_ptr(), B, C, H * W, 1e-5, 0.1, 1, 1, st()))
        r["bn bwd (stats+apply)"] = timed(lambda: L.dc_bn_relu_bwd(ptr(x), None, ptr(gy), ptr(gamma), ptr(mean), ptr(invstd), ptr(y), None, ptr(dg), ptr(db),
                                                                  bws.data_ptr(), mask.data_ptr(), B, C, H * W, 1, 1, st()))
        r["bn_stats"] = timed(lambda: L.dc_bn_stats(ptr(x), ptr(part), B, C, H * W, groups, st()))
        r["finalize"] = timed(lambda: L.dc_bn_finalize(ptr(part), np_s, ppg_s, cnt, ptr(gamma), ptr(beta), ptr(rm), ptr(rv), tab[0].data_ptr(), tab[1].data_ptr(),
                                                       tab[2].data_ptr(), tab[3].data_ptr(), C, groups, 1e-5, 0.1, st()))
        r["bn_apply2 (+res)"] = timed(lambda: L.dc_bn_apply(ptr(x), ptr(add), tab[2].data_ptr(), tab[3].data_ptr(), ptr(y), mask.data_ptr(), B, C, H * W, 1, groups, st()))
        r["bwd_finalize"] = timed(lambda: L.dc_bn_bwd_finalize(ptr(part), np_s, ppg_s, cnt, ptr(gamma), tab[0].data_ptr(), tab[1].data_ptr(), ptr(coef), ptr(dg), ptr(db), C, groups, st()))
        r["bwd_apply2"] = timed(lambda: L.dc_bn_bwd_apply(ptr(x), ptr(gy), ptr(coef), ptr(y), B, C, H * W, groups, st()))
        # ---- Winograd 3x3 C -> C
        f0 = BnFold(); f0.groups = groups
        fs = BnFold(); fs.groups = groups; fs.stat_part = part.data_ptr()
        fi = BnFold(); fi.groups = groups; fi.in_scale, fi.in_shift = tab[2].data_ptr(), tab[3].data_ptr()
        fis = BnFold(); fis.groups = groups; fis.in_scale, fis.in_shift = tab[2].data_ptr(), tab[3].data_ptr(); fis.stat_part = part.data_ptr()
        fb2 = BnFold(); fb2.groups = groups; fb2.in_scale, fb2.in_shift = tab[2].data_ptr(), tab[3].data_ptr()
        fb2.bn_x, fb2.bn_mean, fb2.bwd_part = x.data_ptr(), tab[0].data_ptr(), part.data_ptr()
        fb3 = BnFold(); fb3.groups = groups; fb3.bn_x, fb3.bn_mean, fb3.bwd_part, fb3.bn_mask = x.data_ptr(), tab[0].data_ptr(), part.data_ptr(), mask.data_ptr()
        r["wino fwd"] = timed(lambda: L.dc_wino3x3_fwd(ptr(x), ptr(w), ptr(y2), ws.data_ptr(), B, C, C, H, W, st()))
        if np_f:
            r["wino fwd +stats"] = timed(lambda: L.dc_wino3x3_fwd_bn(ptr(x), ptr(w), ptr(y2), ws.data_ptr(), B, C, C, H, W, ctypes.byref(fs), st()))
            r["wino fwd +fold"] = timed(lambda: L.dc_wino3x3_fwd_bn(ptr(x), ptr(w), ptr(y2), ws.data_ptr(), B, C, C, H, W, ctypes.byref(fi), st()))
            r["wino fwd +fold+stats"] = timed(lambda: L.dc_wino3x3_fwd_bn(ptr(x), ptr(w), ptr(y2), ws.data_ptr(), B, C, C, H, W, ctypes.byref(fis), st()))
        r["wino dgrad"] = timed(lambda: L.dc_wino3x3_dgrad(ptr(gy), ptr(w), ptr(y2), ws.data_ptr(), B, C, C, H, W, st()))
        r["wino dgrad +add"] = timed(lambda: L.dc_wino3x3_dgrad_add(ptr(gy), ptr(w), ptr(y2), ptr(add), ws.data_ptr(), B, C, C, H, W, st()))
        if np_b:
            r["wino dgrad +bn(re-derive)"] = timed(lambda: L.dc_wino3x3_dgrad_bn(ptr(gy), ptr(w), ptr(y2), None, ws.data_ptr(), B, C, C, H, W, ctypes.byref(fb2), st()))
            r["wino dgrad +add+bn(bits)"] = timed(lambda: L.dc_wino3x3_dgrad_bn(ptr(gy), ptr(w), ptr(y2), ptr(add), ws.data_ptr(), B, C, C, H, W, ctypes.byref(fb3), st()))
        r["wino wgrad"] = timed(lambda: L.dc_wino3x3_wgrad(ptr(x), ptr(gy), ptr(dw), wws.data_ptr(), B, C, C, H, W, st()))
        r["wino wgrad +fold"] = timed(lambda: L.dc_wino3x3_wgrad_bn(ptr(x), ptr(gy), ptr(dw), wws.data_ptr(), B, C, C, H, W, ctypes.byref(fi), st()))
        # ---- 1x1: C -> 4C (conv3, input folded) and 4C -> C (conv1 of the next block: link epilogue)
        C4 = 4 * C
        if B * C4 * H * W * 4 < 2 ** 31:
            w3 = torch.randn(C4, C, device=dev) * 0.05
            dw3 = torch.empty_like(w3)
            y3 = torch.empty(B, C4, H, W, device=dev)
            g3 = torch.randn(B, C4, H, W, device=dev)
            np3 = L.dc_conv1x1_stat_parts(B, C, C4, H, W, 1, groups, ctypes.byref(ppg))
            np3b = L.dc_conv1x1_bwd_parts(B, C, C4, H, W, groups, ctypes.byref(ppg))
            part3 = torch.empty(max(np3, np3b, 1) * C4 * 2, device=dev)
            wws3 = torch.empty(L.dc_conv1x1_wgrad_workspace(B, C, C4, H, W, 1), dtype=torch.uint8, device=dev)
            f3s = BnFold(); f3s.groups = groups; f3s.stat_part = part3.data_ptr()
            f3is = BnFold(); f3is.groups = groups; f3is.stat_part = part3.data_ptr(); f3is.in_scale, f3is.in_shift = tab[2].data_ptr(), tab[3].data_ptr()
            f3b = BnFold(); f3b.groups = groups; f3b.in_scale, f3b.in_shift = tab[2].data_ptr(), tab[3].data_ptr()
            f3b.bn_x, f3b.bn_mean, f3b.bwd_part = x.data_ptr(), tab[0].data_ptr(), part3.data_ptr()
            r["g1 C->4C fwd"] = timed(lambda: L.dc_conv1x1_fwd(ptr(x), ptr(w3), ptr(y3), B, C, C4, H, W, 1, st()))
            r["g1 C->4C fwd +stats"] = timed(lambda: L.dc_conv1x1_fwd_bn(ptr(x), ptr(w3), ptr(y3), B, C, C4, H, W, 1, ctypes.byref(f3s), st()))
            r["g1 C->4C fwd +fold+stats"] = timed(lambda: L.dc_conv1x1_fwd_bn(ptr(x), ptr(w3), ptr(y3), B, C, C4, H, W, 1, ctypes.byref(f3is), st()))
            r["g1 C->4C dgrad"] = timed(lambda: L.dc_conv1x1_dgrad(ptr(g3), ptr(w3), ptr(y2), B, C, C4, H, W, 1, st()))
            r["g1 C->4C dgrad +bn(re-derive)"] = timed(lambda: L.dc_conv1x1_dgrad_bn(ptr(g3), ptr(w3), ptr(y2), None, B, C, C4, H, W, 1, ctypes.byref(f3b), st()))
            r["g1 C->4C wgrad"] = timed(lambda: L.dc_conv1x1_wgrad(ptr(x), ptr(g3), ptr(dw3), wws3.data_ptr(), B, C, C4, H, W, 1, st()))
            r["g1 C->4C wgrad +fold"] = timed(lambda: L.dc_conv1x1_wgrad_bn(ptr(x), ptr(g3), ptr(dw3), wws3.data_ptr(), B, C, C4, H, W, 1, ctypes.byref(f3is), st()))
        print("B=%d C=%d %dx%d  (%.1f MB per tensor)" % (B, C, H, W, B * C * H * W * 4 / 1e6))
        for k, v in r.items():
            print("   %-34s %8.1f us" % (k, v))
        sys.stdout.flush()


if __name__ == "__main__":
    main()
