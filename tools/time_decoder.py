#!/usr/bin/env python3
"""DepthDecoder forward+backward at BASELINE config 2 shapes (B=12, resnet18 features of a 192x640 image);
run under tools/prof_decoder.sh for per-kernel times.  DC_MIN_PIXELS overrides layers.FUSED_CONV_MIN_PIXELS."""
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "self-supervised-depth-estimation_amd"))
import layers  # noqa: E402
import networks  # noqa: E402


def main():
    if "DC_MIN_PIXELS" in os.environ:
        layers.FUSED_CONV_MIN_PIXELS = int(os.environ["DC_MIN_PIXELS"])
    dev = torch.device("cuda:0")
    nce = np.array([64, 64, 128, 256, 512])
    torch.manual_seed(0)
    dec = networks.DepthDecoder(nce).to(dev)
    g = torch.Generator(device=dev).manual_seed(1)
    feats = [torch.randn(12, int(c), 96 >> i, 320 >> i, device=dev, generator=g, requires_grad=True)
             for i, c in enumerate(nce)]
    def step():
        o = dec(feats)
        tot = sum(o[("disp", s)].sum() for s in range(4))
        torch.autograd.grad(tot, feats + list(dec.parameters()))
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    print("decoder fwd+bwd: %.3f ms per step (min_pixels=%d)" % ((time.perf_counter() - t0) / n * 1e3, layers.FUSED_CONV_MIN_PIXELS))


if __name__ == "__main__":
    main()
