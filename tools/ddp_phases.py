#!/usr/bin/env python3
"""Per-phase wall time of Trainer.train_step under torch.distributed (rehearsal / debugging of depthcore/ddp.py).
Launch with torch.distributed.run; DC_DIST_BACKEND selects the backend (gloo for several ranks on one GPU)."""
import os
import sys
import time

import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "self-supervised-depth-estimation_amd"))
import trainer as T  # noqa: E402
from depthcore.synthetic import synthetic_batch  # noqa: E402


def main():
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    if world > 1:
        dist.init_process_group(os.environ.get("DC_DIST_BACKEND", "nccl"), rank=rank, world_size=world)
    opt = T.default_options(batch_size=12, overlap_streams=os.environ.get("OVERLAP", "1") == "1")
    tr = T.Trainer(opt, device=dev, rank=rank, world_size=world)
    tr.set_train()
    inputs = synthetic_batch(12, 192, 640, dev, seed=100 + rank)
    for it in range(5):
        ts = [time.perf_counter()]

        def mark():
            torch.cuda.synchronize()
            ts.append(time.perf_counter())
        outputs, losses = tr.process_batch(inputs); mark()
        tr.buckets.zero()
        losses["loss"].backward(); mark()
        tr.buckets.finish(); mark()
        tr.model_optimizer.step(); mark()
        if rank == 0:
            print("it %d: fwd %.1f  bwd %.1f  finish %.1f  adam %.1f ms" % ((it,) + tuple((ts[i + 1] - ts[i]) * 1e3 for i in range(4))), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
