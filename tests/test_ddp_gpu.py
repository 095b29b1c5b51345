"""The gradient exchange on the real stack: RCCL (torch.distributed "nccl"), the hooks firing on the two compute streams
of Trainer.overlap_streams, the dedicated communication stream, finish() and the optimiser step.

One GPU box has one GPU, so the process group of the first test has a single rank; the Trainer is told world_size = 2,
which turns the bucket machinery on (gradients produced inside their bucket slices -> ReduceOp.AVG over the 1-rank group
-> .grad = slices).  The mean over a 1-rank group is the local gradient, so every gradient must equal the plain
single-GPU trainer's from the same state bit for bit (same kernels, fixed-order reductions) -- any lost, stale or misrouted
gradient would show.  At 64 x 128 every convolution is a depthcore launch, so NO convolution / BatchNorm gradient needs a pack copy.
The last test runs by itself the moment a box shows two GPUs: `bench.py --gpus 2` over RCCL."""
import os
import socket

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("front", ["vanilla", "fusion", "gru"])
def test_bucketed_exchange_over_rccl_with_stream_overlap(front):
    """`fusion`: the Fusion_v3 parameters (3-element rel_h / rel_w, 1-element biases) sit between the others in the flat
    bucket -- every slice must still start on a 16-byte boundary for the kernels' vector stores.  `gru`: the ConvGRU level
    nodes write the cells' shared weights' gradients (one pass over the sequence's frames) straight into their slices; only
    the five learned initial states are packed."""
    import trainer as T
    from depthcore.synthetic import synthetic_batch, synthetic_sequence_batch
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
    try:
        B, H, W = 2, 64, 128
        kw = dict(fusion="v3", frame_ids=[0, -2, -1, 1]) if front == "fusion" else {}
        inputs = synthetic_batch(B, H, W, torch.device(DEV), seed=5, frame_ids=(0, -2, -1, 1) if front == "fusion" else (0, -1, 1))
        if front == "gru":
            B, kw = 1, dict(gru="v5", len_sequence=3)
            inputs = synthetic_sequence_batch(3, H, W, torch.device(DEV), seed=5)
        ref = T.Trainer(T.default_options(batch_size=B, height=H, width=W, **kw), device=DEV, seed=11)
        ddp = T.Trainer(T.default_options(batch_size=B, height=H, width=W, **kw), device=DEV, rank=0, world_size=2, seed=11)
        for k in ref.models:
            ddp.models[k].load_state_dict(ref.models[k].state_dict())
        assert len(ddp.buckets.flat) >= 1 and ddp.buckets.world == 2
        for views in ddp.buckets.views:
            assert all(v.data_ptr() % 16 == 0 for v in views)
        ref.set_train()
        ddp.set_train()
        for step in range(2):
            ref.step = ddp.step = step                     # same on-device tie-break noise stream
            for k in ref.models:
                ddp.models[k].load_state_dict(ref.models[k].state_dict())
            grads = []
            for tr in (ref, ddp):
                _, losses = tr.process_batch(dict(inputs))
                tr.buckets.zero()
                losses["loss"].backward()
                tr.buckets.finish()                        # ddp: wait for RCCL, .grad = views into the flat buckets
                torch.cuda.synchronize()
                grads.append({n: p.grad.detach().clone() for k, m in tr.models.items() for n, p in
                              ((k + "." + n_, p_) for n_, p_ in m.named_parameters()) if p.grad is not None})
                grads[-1]["__loss__"] = float(losses["loss"].detach())
            assert grads[0]["__loss__"] == grads[1]["__loss__"]
            # a parameter without a gradient (Fusion_v3's unused last `upscale`) stays None on one GPU and contributes zeros to
            # the exchange (depthcore/ddp.py, same as torch's DistributedDataParallel)
            extra = set(grads[1]) - set(grads[0])
            assert set(grads[0]) <= set(grads[1]) and all("fusion_block_4.upscale" in n and not grads[1][n].any() for n in extra), extra
            for n in grads[0]:
                if n == "__loss__":
                    continue
                g0, g1 = grads[0][n], grads[1][n]
                # RCCL's AVG over the (1-rank) group = the local gradient, written by the same deterministic kernels
                err = float((g1 - g0).norm() / (g0.norm() + 1e-30))
                assert err <= 1e-6, (n, err, float(g0.norm()), float(g1.norm()), step)
            # every convolution / BatchNorm gradient of the step was written straight into its bucket slice: nothing was packed.
            # (Fusion_v3: the 192 AttentionConv parameter tensors -- 2 to 16 floats each -- leave dc_attnconv_bwd as one packed
            # vector per unit and reach their slices through the bucket's single multi-tensor copy.)
            n_attn = sum(1 for n, _ in ddp.models["fusion"].named_parameters() if ".atten" in n) if front == "fusion" else 0
            if front == "gru":
                n_attn = 5                                 # the learned initial states h0_layer1 of the five levels
            assert ddp.buckets.packed == n_attn and n_attn in (0, 5, 192), (ddp.buckets.packed, n_attn)
            assert ddp.buckets.launch_order == sorted(ddp.buckets.launch_order)
            # move on to another point of weight space for the next round (the ddp copy is re-synchronised there)
            ref.model_optimizer.step()
            ddp.model_optimizer.step()                     # also exercises the optimiser on the bucket-view gradients
        assert ddp.buckets.comm is not None                # the exchange really went through the communication stream
    finally:
        dist.destroy_process_group()


def test_two_ranks_on_one_gpu_full_bench_step_over_gloo():
    """The whole N > 1 path on the real kernels: `bench.py --gpus 2` launches two rank processes that share this box's single
    GPU (`--oversubscribe`, gloo instead of RCCL because two ranks cannot share a device under RCCL): initial broadcast,
    gradients written into the flat bucket slices by the HIP kernels, bucketed exchange overlapped with the backward, Adam --
    and bench.py's own replica check (all weights of all ranks agree after the timed steps) must pass.  One JSON line,
    n_gpus = 2, whole-job throughput."""
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DC_DIST_BACKEND="gloo", MASTER_PORT=str(_free_port()))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--oversubscribe", "--steps", "3", "--warmup", "1",
                        "--batch", "2", "--height", "64", "--width", "96", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["scaling"] == "weak"
    assert d["config"]["global_batch"] == 4 and d["config"]["parallelism"] == "dp2"
    assert d["value"] > 0


def test_graph_mode_with_bucketed_exchange_over_rccl():
    """hipGraph mode at world > 1 (round 4): the whole data-parallel step -- both compute streams, the backward whose hooks
    record the buckets' all-reduces on the communication stream in fixed bucket order, finish(), Adam -- is ONE captured
    graph.  Validated on the one GPU a box has: 1-rank RCCL group, Trainer told world = 2 (tests/ddp_graph_child.py, its own
    process).  The replays must follow the eager data-parallel trainer (same kernels, deterministic reductions: losses to
    1e-5, weights after 8 Adam steps to 1e-4 of their norm -- the captured Adam forms its bias corrections on the device), and
    the host's work per replayed step must be a small fraction of the eager launch work (<= 3 ms)."""
    import json
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(here, "ddp_graph_child.py")], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.returncode, r.stderr[:4000], r.stderr[-2000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["captured"] and d["capture_failed"] == 0 and not d["eager_captured"], d
    el, gl = d["eager_losses"], d["graph_losses"]
    assert all(abs(a - b) <= 1e-5 * abs(a) for a, b in zip(el[:3], gl[:3])), (el, gl)       # eager warm-up steps of both
    assert all(abs(a - b) <= 5e-4 * abs(a) for a, b in zip(el, gl)), (el, gl)               # captured step + replays
    assert len(set(round(v, 7) for v in gl[3:])) == len(gl[3:])                              # every replay is a new step
    assert d["weight_rel_diff"] <= 1e-4, d["weight_rel_diff"]
    assert d["launch_order"] == sorted(d["launch_order"]) and d["packed"] == 0
    replay_ms = sorted(d["host_ms_graph"][4:])                                               # steps after the capture
    assert replay_ms[len(replay_ms) // 2] <= 3.0, d["host_ms_graph"]
    print("host ms per step: eager DDP %s | graph DDP %s" % (d["host_ms_eager"], d["host_ms_graph"]))


def test_graph_mode_capture_failure_is_agreed_over_the_ranks_and_falls_back_to_eager():
    """with the captured collectives on a group of their own, a capture that fails on ONE rank must turn every
    rank eager (the decision is an all-reduce(MIN) on the eager base group), or the ranks wait on different communicators.  Here
    on the one GPU a box has: the graph trainer's capture is forced to fail (DC_TEST_FAIL_CAPTURE_RANK=0), the flag travels
    through the 1-rank RCCL group, the shape is marked eager-only, and the run follows the eager trainer step for step; the
    capture-only communicator is destroyed by close().  (Two ranks, one failing: tests/test_ddp_cpu.py on gloo for the
    agreement, test_two_gpus_capture_failure_on_one_rank below for the whole path on a multi-GPU box.)"""
    import json
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), DC_TEST_FAIL_CAPTURE_RANK="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(here, "ddp_graph_child.py")], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.returncode, r.stderr[:4000], r.stderr[-2000:])
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert not d["captured"] and d["capture_failed"] == 1 and d["had_capture_pg"], d
    el, gl = d["eager_losses"], d["graph_losses"]
    assert all(abs(a - b) <= 1e-5 * abs(a) for a, b in zip(el, gl)), (el, gl)     # eager all the way: the same kernels
    assert d["weight_rel_diff"] <= 1e-6, d["weight_rel_diff"]


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs: runs by itself on a multi-GPU box")
def test_two_gpus_capture_failure_on_one_rank():
    """Two real ranks over RCCL, graph mode, rank 1's capture forced to fail: both ranks must finish (no hang on mismatched
    communicators), both eager, replicas identical (bench.py's own divergence guard)."""
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_PORT=str(_free_port()), DC_TEST_FAIL_CAPTURE_RANK="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "DC_DIST_BACKEND"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "5", "--graph",
                        "--batch", "2", "--height", "64", "--width", "128", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert d["n_gpus"] == 2 and d["value"] > 0


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs: runs by itself on a multi-GPU box")
def test_two_gpus_bench_step_over_rccl():
    """`bench.py --gpus 2` on two real GPUs over RCCL/xGMI (fresh child processes, one rank per GPU): one JSON line, the
    replica-divergence guard inside bench.py passes (it exits non-zero otherwise), all gradient bytes went through the
    exchange, and the exposed wait is reported next to DESIGN's prediction."""
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_PORT=str(_free_port()))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "DC_DIST_BACKEND"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["scaling"] == "weak"
    assert d["config"]["global_batch"] == 24 and d["config"]["parallelism"] == "dp2"
    assert d["grad_bytes_allreduced_per_step"] > 100e6          # 26.8 M fp32 parameters
    assert d["value"] > 0


def test_wgrad_lanes_with_the_bucketed_exchange_change_nothing():
    """Weight-gradient lanes together with the bucketed exchange (1-rank RCCL group, Trainer told world = 2): the lanes' kernels
    write into the bucket slices, the communication stream waits for them, and three training steps end in bitwise the same
    weights as with the lanes off."""
    import trainer as T
    from depthcore.synthetic import synthetic_batch
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
    try:
        B, H, W = 2, 64, 128
        batches = [synthetic_batch(B, H, W, torch.device(DEV), seed=s) for s in (5, 6)]
        out = {}
        for lanes in (0, 1):
            tr = T.Trainer(T.default_options(batch_size=B, height=H, width=W, wgrad_lanes=lanes), device=DEV, rank=0, world_size=2, seed=11)
            assert tr.wgrad_lanes == bool(lanes) and tr.buckets.world == 2
            tr.set_train()
            losses = []
            for i in range(3):
                _, l = tr.train_step(dict(batches[i % 2]))
                losses.append(float(l["loss"].detach()))
            torch.cuda.synchronize()
            assert tr.buckets.packed == 0
            out[lanes] = (losses, torch.cat([p.detach().flatten() for p in tr.parameters_to_train]).clone())
            tr.close()
        assert out[0][0] == out[1][0], (out[0][0], out[1][0])
        assert torch.equal(out[0][1], out[1][1])
    finally:
        dist.destroy_process_group()
