"""Child process of tests/test_train_gpu.py::test_illegal_call_inside_a_capture_raises_and_does_not_abort: an ILLEGAL call (a
device synchronisation) inside the captured training step.  It may leave the HIP runtime's capture state beyond repair for the
process, which is why this runs in a process of its own.  Prints `outcome=raised` (RuntimeError from the Trainer) or
`outcome=recovered` (the steps went on eagerly); anything else -- an abort, a hang -- fails the parent test."""
import os
import sys
import warnings

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "self-supervised-depth-estimation_amd")]


def main():
    import trainer as T
    from depthcore.synthetic import synthetic_batch
    dev = torch.device("cuda:0")
    batches = [synthetic_batch(2, 64, 128, dev, seed=s) for s in (2, 3)]
    a = T.Trainer(T.default_options(height=64, width=128, batch_size=2, hip_graph=True), device="cuda:0", seed=5)
    a.set_train()
    orig = a._train_step_eager

    def sabotaged(inputs):
        out = orig(inputs)
        if torch.cuda.is_current_stream_capturing():
            torch.cuda.synchronize()              # illegal during capture: invalidates it
        return out
    a._train_step_eager = sabotaged
    outcome = []
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        try:
            for i in range(5):
                outcome.append(float(a.train_step(dict(batches[i % 2]))[1]["loss"].detach()))
        except RuntimeError as e:
            assert "capture" in str(e), e
            outcome.append("raised")
    assert a._graph is None and getattr(a, "_graph_failed", 0) == 1
    assert outcome[-1] == "raised" or (len(outcome) == 5 and all(np.isfinite(v) for v in outcome)), outcome
    print("outcome=%s" % ("raised" if outcome[-1] == "raised" else "recovered"), flush=True)
    os._exit(0)             # (a poisoned runtime may not tear down cleanly: the verdict is already printed)


if __name__ == "__main__":
    main()
