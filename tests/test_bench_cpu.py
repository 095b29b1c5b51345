"""bench.py's multi-rank plumbing on a machine without GPUs: the self-launcher (fresh child per rank, rendezvous on
127.0.0.1, rank 0's single JSON line relayed), its refusal to print a mislabelled line, and the bucketed exchange of
depthcore/ddp.py over gloo inside it.  The GPU step itself is replaced by the `--rehearse` stand-in (flagged in the line)."""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(REPO, "bench.py")


def _run(args, env=None, timeout=240):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH] + args, env=e, capture_output=True, text=True, timeout=timeout)


def test_gpus_2_without_gpus_refuses_cleanly():
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("needs a machine with fewer than 2 GPUs")
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "1"])
    assert r.returncode == 2, (r.returncode, r.stderr[-400:])
    assert r.stdout.strip() == ""                       # no mislabelled n_gpus:1 line
    assert "--gpus 2" in r.stderr and "GPU" in r.stderr


def test_world_size_mismatch_is_an_error():
    r = _run(["--gpus", "2", "--rehearse"], env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and r.stdout.strip() == ""
    assert "WORLD_SIZE" in r.stderr


def test_self_launch_two_ranks_rehearsal_over_gloo():
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "2", "--rehearse"], env={"DC_DIST_BACKEND": "gloo"})
    assert r.returncode == 0, r.stderr[-800:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1                              # ONE line, from rank 0
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["rehearsal"] is True
    assert d["config"]["global_batch"] == 4 and d["config"]["parallelism"] == "dp2"
    assert d["grad_bytes_allreduced_per_step"] > 0 and d["value"] > 0
