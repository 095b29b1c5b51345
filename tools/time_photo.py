#!/usr/bin/env python3
"""Quick timing of the fused photometric kernels at BASELINE config 2 (B=12, 192x640, 4 scales)."""
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "self-supervised-depth-estimation_amd"))
sys.path.insert(0, REPO)
from depthcore import ops  # noqa: E402


def main():
    B, H, W = 12, 192, 640
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    imgs = [torch.rand(B, 3, H, W, device=dev, generator=g) for _ in range(3)]
    imgs = [torch.nn.functional.avg_pool2d(torch.nn.functional.pad(i, (2, 2, 2, 2), mode="reflect"), 5, 1) for i in imgs]
    color_s = [imgs[0] if s == 0 else torch.nn.functional.avg_pool2d(imgs[0], 2 ** s) for s in range(4)]
    K = torch.tensor([[0.58 * W, 0, 0.5 * W, 0], [0, 1.92 * H, 0.5 * H, 0], [0, 0, 1, 0], [0, 0, 0, 1]], device=dev)
    invK = torch.linalg.pinv(K)
    K, invK = K.expand(B, 4, 4).contiguous(), invK.expand(B, 4, 4).contiguous()
    T = []
    for f in range(2):
        t = torch.eye(4, device=dev).repeat(B, 1, 1)
        t[:, :3, 3] = 0.01 * torch.randn(B, 3, device=dev, generator=g)
        T.append(t.requires_grad_())
    # smooth disparity fields (a network's output is smooth; per-pixel white noise would scatter every
    # bilinear gather over 64 cache lines and benchmark the texture path instead of the kernel)
    def smooth_disp(h, w):
        lo = torch.rand(B, 1, 6, 20, device=dev, generator=g)
        return torch.nn.functional.interpolate(lo, size=(h, w), mode="bicubic", align_corners=False).clamp(0.01, 0.99)
    if os.environ.get("DC_WHITE_NOISE_DISP"):
        disps = [torch.rand(B, 1, H >> s, W >> s, device=dev, generator=g).requires_grad_() for s in range(4)]
    else:
        disps = [smooth_disp(H >> s, W >> s).contiguous().requires_grad_() for s in range(4)]
    noise = [torch.randn(B, 2, H, W, device=dev, generator=g) for _ in range(4)]
    kw = {}
    for k in os.environ.get("DC_PHOTO_FLAGS", "").split(","):
        if k:
            kw[k] = True
    for mode, nz in (("external-noise", noise), ("device-rng", None)):
        cfg = ops.PhotoConfig(imgs[0], imgs[1], imgs[2], color_s, K, invK, noise=nz, **kw)
        for _ in range(3):
            l = ops.photometric_loss(cfg, T[0], T[1], disps)
            l[4].backward()
        torch.cuda.synchronize()
        n = 20
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        tf = tb = 0.0
        for _ in range(n):
            e[0].record()
            l = ops.photometric_loss(cfg, T[0], T[1], disps)
            e[1].record()
            l[4].backward()
            e[2].record()
            torch.cuda.synchronize()
            tf += e[0].elapsed_time(e[1])
            tb += e[1].elapsed_time(e[2])
        fb = ops.photo_algorithmic_bytes(cfg, T[0], T[1], disps, False)
        bb = ops.photo_algorithmic_bytes(cfg, T[0], T[1], disps, True)
        print("%s: fwd %.1f us (%.2f TB/s algorithmic)  bwd %.1f us (%.2f TB/s)  loss %.6f" % (
            mode, tf / n * 1e3, fb / (tf / n * 1e-3) / 1e12, tb / n * 1e3, bb / (tb / n * 1e-3) / 1e12, float(l[4])))


if __name__ == "__main__":
    main()
