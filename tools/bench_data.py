#!/usr/bin/env python3
"""Row f4 timing: the device data step for one BASELINE-config batch (B items x 3 frames of native 375 x 1242 uint8 ->
4-scale "color" + "color_aug" float tensors), HIP events around the whole call, frames already resident in HBM.
Algorithmic bytes = native bytes read once + the float tensors written once (intermediate uint8 levels are not counted)."""
import argparse
import json
import os
import random
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "self-supervised-depth-estimation_amd"))
from depthcore.data import GpuPreprocessor, sample_item  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=12)
    ap.add_argument("--height", type=int, default=192)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    native = torch.randint(0, 256, (3, a.batch, 375, 1242, 3), dtype=torch.uint8, device=dev)
    pre = GpuPreprocessor(a.height, a.width, device=dev)
    rng = random.Random(0)
    out = {}
    for mode in ("all_jitter", "sampled", "no_jitter"):
        if mode == "sampled":
            items = [sample_item(True, rng) for _ in range(a.batch)]
        else:
            items = [(True, ((2, 0, 3, 1), (1.1, 0.9, 1.15, 0.05)) if mode == "all_jitter" else None) for _ in range(a.batch)]
        flips, jit = [i[0] for i in items], [i[1] for i in items]
        for _ in range(3):
            pre(native, flips, jit)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            pre(native, flips, jit)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.iters
        n_out = sum(3 * a.batch * 3 * (a.height >> s) * (a.width >> s) * 4 for s in range(4)) * (2 if mode != "no_jitter" else 1)
        alg = native.numel() + n_out
        out[mode] = {"ms_per_batch": round(ms, 4), "items_per_s": round(a.batch / ms * 1e3, 1), "algorithmic_bytes": alg,
                     "GBps": round(alg / ms / 1e6, 1), "frac_of_8TBps": round(alg / ms / 1e6 / 8000, 4)}
    print(json.dumps({"workload": "data step B=%d 375x1242 -> %dx%d x4 scales x3 frames" % (a.batch, a.height, a.width), **out}))


if __name__ == "__main__":
    main()
