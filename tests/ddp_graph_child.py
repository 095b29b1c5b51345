"""Child process of tests/test_ddp_gpu.py::test_graph_mode_with_bucketed_exchange_over_rccl (its own process: a collective that
cannot be captured on some stack must fail this test, not take the whole test session down).

1-rank RCCL group, Trainer told world_size = 2 (the bucket machinery, the communication stream and the all-reduces are all
live; the mean over one rank is the local gradient).  An eager DDP trainer and a graph-mode DDP trainer start from the same
state and see the same batches: the captured step -- both compute streams, the backward with its hooks' all-reduces recorded on
the communication stream in bucket order, finish(), Adam -- must replay to the eager trainer's losses and weights.
DC_TEST_FAIL_CAPTURE_RANK=0 forces the graph trainer's first capture attempt to fail: the outcome is agreed over the ranks
(depthcore.ddp.agree_all on the eager base group) and the steps go on eagerly on that group.
Prints one JSON line."""
import json
import os
import sys
import time

import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "self-supervised-depth-estimation_amd")]


def main():
    import trainer as T
    from depthcore.synthetic import synthetic_batch
    dev = torch.device("cuda:0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    B, H, W, N = 2, 64, 128, 8
    batches = [synthetic_batch(B, H, W, dev, seed=s) for s in (2, 3)]
    out = {}
    try:
        runs = {}
        for name, graph in (("eager", False), ("graph", True)):
            tr = T.Trainer(T.default_options(batch_size=B, height=H, width=W, hip_graph=graph), device="cuda:0", rank=0,
                           world_size=2, seed=5)
            tr.set_train()
            assert tr.buckets.world == 2 and len(tr.buckets.flat) >= 1
            losses, host = [], []
            for i in range(N):
                t0 = time.perf_counter()
                _, l = tr.train_step(dict(batches[i % 2]))
                host.append(time.perf_counter() - t0)
                losses.append(float(l["loss"].detach()))
            torch.cuda.synchronize()
            w = torch.cat([p.detach().flatten() for p in tr.parameters_to_train])
            runs[name] = dict(losses=losses, w=w.clone(), captured=tr._graph is not None, failed=getattr(tr, "_graph_failed", 0),
                              agreed_off=getattr(tr, "_graph_agreed_off", 0), had_capture_pg=bool(tr._capture_pg),
                              host_ms=[round(h * 1e3, 3) for h in host], launch_order=list(tr.buckets.launch_order),
                              packed=tr.buckets.packed)
            tr.close()
        e, g = runs["eager"], runs["graph"]
        out = dict(eager_losses=e["losses"], graph_losses=g["losses"], captured=g["captured"], capture_failed=g["failed"],
                   eager_captured=e["captured"], agreed_off=g["agreed_off"], had_capture_pg=g["had_capture_pg"], host_ms_eager=e["host_ms"], host_ms_graph=g["host_ms"],
                   weight_rel_diff=float((e["w"] - g["w"]).norm() / e["w"].norm()),
                   weight_max_abs_diff=float((e["w"] - g["w"]).abs().max()), launch_order=g["launch_order"], packed=g["packed"])
    finally:
        dist.destroy_process_group()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
