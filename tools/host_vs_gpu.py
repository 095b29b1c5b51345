#!/usr/bin/env python3
"""Is the training step bound by host-side launch work or by the GPU?  Host time per step (enqueue only) vs wall."""
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "self-supervised-depth-estimation_amd"))
import trainer as T  # noqa: E402
from depthcore.synthetic import synthetic_batch  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    tr = T.Trainer(T.default_options(batch_size=12, overlap_streams=os.environ.get("OVERLAP", "1") == "1"), device=dev)
    tr.set_train()
    inputs = synthetic_batch(12, 192, 640, dev, seed=1)
    for _ in range(5):
        tr.train_step(inputs)
    torch.cuda.synchronize()
    n = 30
    t0 = time.perf_counter()
    for _ in range(n):
        tr.train_step(inputs)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):                     # queue empty: no back-pressure on the host
        t1 = time.perf_counter()
        tr.train_step(inputs)
        ts.append((time.perf_counter() - t1) * 1e3)
    torch.cuda.synchronize()
    print("host time of single steps on an empty queue: %s ms" % ", ".join("%.2f" % t for t in ts))
    print("host enqueue %.2f ms/step, wall %.2f ms/step (GPU drains %.2f ms after the last enqueue)"
          % (t_host / n * 1e3, t_all / n * 1e3, (t_all - t_host) * 1e3))


if __name__ == "__main__":
    main()
