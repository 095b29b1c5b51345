#!/usr/bin/env python3
"""Soak: N training steps at the BASELINE config on changing synthetic batches; reports loss trend, NaNs and allocator growth."""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "self-supervised-depth-estimation_amd"))
import trainer as T  # noqa: E402
from depthcore.synthetic import synthetic_batch  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    dev = torch.device("cuda:0")
    B, H, W, NL = (int(os.environ.get(k, d)) for k, d in (("DC_B", 12), ("DC_H", 192), ("DC_W", 640), ("DC_LAYERS", 18)))
    if os.environ.get("DC_FRONT") == "gru":        # BASELINE configs[3]: sequences of 3 frames, batch size 1
        from depthcore.synthetic import synthetic_sequence_batch
        tr = T.Trainer(T.default_options(batch_size=1, height=H, width=W, num_layers=NL, gru="v5", len_sequence=3), device=dev)
        batches = [synthetic_sequence_batch(3, H, W, dev, seed=s) for s in range(4)]
    else:
        tr = T.Trainer(T.default_options(batch_size=B, height=H, width=W, num_layers=NL), device=dev)
        batches = [synthetic_batch(B, H, W, dev, seed=s) for s in range(4)]
    tr.set_train()
    losses = []
    mem0 = None
    trace = []
    ctx = tr.on_step_stream()          # the loop on the trainer's step stream (high priority where opt.step_priority says so), as bench.py runs it
    ctx.__enter__()
    for i in range(n):
        _, l = tr.train_step(batches[i % 4])
        if i % 25 == 0 or i == n - 1:
            v = float(l["loss"].detach())
            losses.append(v)
            if not v == v:
                raise SystemExit("NaN at step %d" % i)
        if i == 20:
            torch.cuda.synchronize()
            mem0 = torch.cuda.memory_reserved()
        if i % 100 == 0:
            trace.append("%.2f" % (torch.cuda.memory_reserved() / 2**30))
    torch.cuda.synchronize()
    print("steps %d: loss %.5f -> %.5f (every 25th: %s)" % (n, losses[0], losses[-1], " ".join("%.4f" % v for v in losses)))
    print("reserved GB every 100 steps: " + " ".join(trace) + "  (weight-gradient lanes %s)" % ("on" if tr.wgrad_lanes else "off"))
    print("reserved memory after 20 steps %.2f GB, at the end %.2f GB, peak allocated %.2f GB"
          % (mem0 / 2**30, torch.cuda.memory_reserved() / 2**30, torch.cuda.max_memory_allocated() / 2**30))


if __name__ == "__main__":
    main()
