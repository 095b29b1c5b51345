#!/usr/bin/env python3
"""Experiment: the whole training step (forward on two streams, backward, fused Adam) captured in ONE hipGraph and replayed.
Prints eager vs replay ms/step.  (The tie-break seed is passed by value, so a replay repeats step 0's noise field: a timing
probe, not a training mode.)"""
import os
import sys
import time

import torch
import torch.optim as optim

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "self-supervised-depth-estimation_amd"))
import trainer as T  # noqa: E402
from depthcore.synthetic import synthetic_batch  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    B = int(os.environ.get("DC_B", 12))
    tr = T.Trainer(T.default_options(batch_size=B), device=dev)
    tr.set_train()
    tr.model_optimizer = optim.Adam(tr.parameters_to_train, tr.opt.learning_rate, fused=True, capturable=True)
    inputs = synthetic_batch(B, 192, 640, dev, seed=1)

    def timed(fn, n=20):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(4):
            tr.train_step(dict(inputs))
    torch.cuda.current_stream().wait_stream(s)
    print("eager: %.3f ms/step" % timed(lambda: tr.train_step(dict(inputs))), flush=True)

    g = torch.cuda.CUDAGraph()
    static = {k: v.clone() for k, v in inputs.items()}
    print("capturing ...", flush=True)
    with torch.cuda.graph(g):
        _, losses = tr.train_step(static)
    print("captured", flush=True)
    g.replay()
    torch.cuda.synchronize()
    l0 = float(losses["loss"])
    print("replay: %.3f ms/step (loss after first replay %.6f)" % (timed(g.replay), l0), flush=True)
    ls = []
    for _ in range(5):
        g.replay()
        torch.cuda.synchronize()
        ls.append(float(losses["loss"]))
    print("losses over replays:", ["%.5f" % v for v in ls])


if __name__ == "__main__":
    main()
