#!/usr/bin/env python3
"""Per-bucket budget of the gradient exchange (DESIGN 4 "exchange budget"), measured on ONE GPU: a 1-rank RCCL group with the
Trainer told world = 2, so that the bucket machinery, the communication stream and the collectives are all live while the step
keeps its single-GPU timing (tests/ddp_graph_child.py uses the same arrangement).  For every bucket in launch order: bytes, the
GPU time after the start of the backward at which its last gradient has landed and its all-reduce can start, and what a ring /
a direct (all-to-all reduce-scatter + all-gather) exchange over xGMI would cost for N = 8 at the link rate of
MI355X_MICROARCH / the task statement (7 links x 153 GB/s per GPU, point-to-point) -- hence the slack before the step's last
backward kernel ends.

    python tools/ddp_budget.py [--num-layers 50 --height 320 --width 1024 --batch 8] [--bucket-mb 32]
"""
import argparse
import os
import socket
import sys

import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "self-supervised-depth-estimation_amd"))
import trainer as T  # noqa: E402
from depthcore.ddp import bucket_timeline  # noqa: E402
from depthcore.synthetic import synthetic_batch  # noqa: E402

LINK_GBS = 153.0      # one xGMI link, one direction
LINKS = 7


def model_ms(nbytes, n=8):
    """ring: 2 (n-1)/n of the bytes leave every GPU over the ring's ONE outgoing link (RCCL can stripe a collective over several
    rings on distinct links: the k-ring time is this / k, k <= 7); direct: reduce-scatter + all-gather as all-to-all, every GPU
    sends 1/n of the bytes to each of its n-1 peers over the n-1 direct links at once, twice."""
    ring1 = 2.0 * (n - 1) / n * nbytes / (LINK_GBS * 1e9) * 1e3
    direct = 2.0 * (nbytes / n) / (LINK_GBS * 1e9) * 1e3
    return ring1, ring1 / LINKS, direct


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--num-layers", type=int, default=18)
    ap.add_argument("--height", type=int, default=192)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--batch", type=int, default=12)
    ap.add_argument("--bucket-mb", type=float, default=0.0)
    ap.add_argument("--steps", type=int, default=6)
    args = ap.parse_args()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(port))
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        opt = T.default_options(batch_size=args.batch, height=args.height, width=args.width, num_layers=args.num_layers)
        if args.bucket_mb > 0:
            opt.bucket_mb = args.bucket_mb
        tr = T.Trainer(opt, device=dev, rank=0, world_size=2)
        tr.set_train()
        inputs = synthetic_batch(args.batch, args.height, args.width, dev, seed=100)
        for _ in range(3):
            tr.train_step(inputs)
        rows, bwd_ms = None, []
        tr.buckets.timing = True
        for _ in range(args.steps):
            outputs, losses = tr.process_batch(inputs)
            e1 = torch.cuda.Event(enable_timing=True)
            tr.buckets.zero()
            losses["loss"].backward()
            e1.record()                      # (current stream: behind the last backward kernel of the main branch)
            tr.buckets.finish()
            tr.model_optimizer.step()
            tr.step += 1
            torch.cuda.synchronize()
            tl = bucket_timeline(tr.buckets)
            bwd_ms.append(tr.buckets._t0.elapsed_time(e1))
            rows = tl                        # (the last step's timeline)
            del outputs, losses
        # (the backward runs on several streams: its end is the later of the main stream's last kernel and the last bucket's readiness)
        bwd = max(sorted(bwd_ms)[len(bwd_ms) // 2], max(t0 for _, _, t0, _ in rows))
        total = sum(nb for _, nb, _, _ in rows)
        print("# %s resnet%d %dx%d B=%d: %d buckets, %.1f MB per step; backward %.2f ms on this GPU (median of %d steps)"
              % (torch.cuda.get_device_name(dev), args.num_layers, args.height, args.width, args.batch, len(rows), total / 1e6, bwd, args.steps))
        print("# bucket | MB | last gradient lands (ms after backward start) | ring, 1 link (ms) | ring striped over 7 links | direct, 7 links | "
              "slack to the end of the backward with 1 ring / 7 rings (ms; queued behind the earlier buckets)")
        free1 = free7 = 0.0
        for b, nb, t0, _ in rows:
            r1, r7, dr = model_ms(nb)
            s1, s7 = max(t0, free1), max(t0, free7)
            free1, free7 = s1 + r1, s7 + r7
            print("%6d | %6.1f | %8.2f | %6.3f | %6.3f | %6.3f | %+7.2f / %+7.2f" % (b, nb / 1e6, t0, r1, r7, dr, bwd - free1, bwd - free7))
        print("# exposed after the backward: 1 ring %.2f ms, 7 rings %.2f ms (negative slack of the last bucket); whole exchange: 1 ring %.2f ms"
              % (max(0.0, free1 - bwd), max(0.0, free7 - bwd), sum(model_ms(nb)[0] for _, nb, _, _ in rows)))
        tr.close()
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
