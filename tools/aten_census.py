#!/usr/bin/env python3
"""Which lines of the host code still launch framework (ATen) kernels in a training step: a TorchDispatchMode over a few steps,
every ATen op that reaches the dispatcher with a device tensor (views and metadata ops excluded) grouped by the innermost Python
frame inside this repository (the autograd engine's own sums have none).

    python tools/aten_census.py            # C2 (DC_B / DC_FRONT=gru|fusion / DC_LAYERS / DC_H / DC_W as tools/host_profile.py)
"""
import collections
import os
import sys

import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode
from torch.utils._pytree import tree_flatten

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "self-supervised-depth-estimation_amd")
sys.path.insert(0, PKG)
import trainer as T  # noqa: E402
from depthcore.synthetic import synthetic_batch, synthetic_sequence_batch  # noqa: E402

SKIP = {"view", "reshape", "slice", "select", "expand", "permute", "transpose", "t", "unsqueeze", "squeeze", "alias", "detach",
        "as_strided", "empty", "empty_like", "empty_strided", "new_empty", "unbind", "split", "split_with_sizes", "narrow", "_unsafe_view",
        "_local_scalar_dense", "lift_fresh", "unfold", "chunk", "record_stream", "resize_", "set_", "is_pinned", "is_same_size",
        "_reshape_alias", "view_as", "expand_as", "flatten", "unflatten", "movedim", "diagonal", "real", "imag", "_has_compatible_shallow_copy_type"}


class Census(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.count = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        out = func(*args, **(kwargs or {}))
        if name.split(".")[1] not in SKIP:
            on_gpu = any(isinstance(a, torch.Tensor) and a.is_cuda for a in tree_flatten((args, kwargs or {}, out))[0])
            if on_gpu:
                where = "(no frame of this repository: the autograd engine's own sums / torch internals)"
                for fr in reversed(traceback.extract_stack()):
                    if fr.filename.startswith(PKG):
                        where = "%s:%d %s" % (fr.filename[len(PKG) + 1:], fr.lineno, fr.name)
                        break
                self.count[(name, where)] += 1
        return out


def main():
    dev = torch.device("cuda:0")
    front = os.environ.get("DC_FRONT", "")
    H, W = int(os.environ.get("DC_H", 192)), int(os.environ.get("DC_W", 640))
    if front == "gru":
        opt = T.default_options(batch_size=1, height=H, width=W, gru="v5", len_sequence=3)
        inputs = synthetic_sequence_batch(3, H, W, dev)
    else:
        B = int(os.environ.get("DC_B", 12))
        kw = dict(fusion="v3") if front == "fusion" else {}
        opt = T.default_options(batch_size=B, height=H, width=W, num_layers=int(os.environ.get("DC_LAYERS", 18)), **kw)
        inputs = synthetic_batch(B, H, W, dev, seed=1, frame_ids=(0, -1, 1, -2) if front == "fusion" else (0, -1, 1))
    tr = T.Trainer(opt, device=dev)
    tr.set_train()
    for _ in range(4):
        tr.train_step(inputs)
    torch.cuda.synchronize()
    steps = 3
    with Census() as cen:
        for _ in range(steps):
            tr.train_step(inputs)
        torch.cuda.synchronize()
    tot = 0
    for (name, where), c in sorted(cen.count.items(), key=lambda kv: -kv[1]):
        print("%6.2f/step  %-28s %s" % (c / steps, name, where))
        tot += c
    print("total %.1f ATen calls on device tensors per step" % (tot / steps))


if __name__ == "__main__":
    main()
