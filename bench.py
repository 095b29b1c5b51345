#!/usr/bin/env python3
"""Headline benchmark: training images/s of the self-supervised depth step on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N = 1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = reference trainer.py:233-237 on one synthetic KITTI-shaped batch: process_batch
(resnet18 depth encoder + decoder on frame 0, pose encoder + decoder on the pairs (-1,0),(0,+1),
4-scale fused warp + SSIM/L1 + automask + smoothness loss) -> backward -> (RCCL all-reduce) -> Adam.
Workload = BASELINE.json configs[1]: resnet18, 192x640, per-GPU batch 12, fp32.  Weak scaling.

Rank 0 prints ONE JSON line (contract in the task statement) with two extra objects:
  roofline     -- the dominant kernel of the step, dc::wino_ps_kernel (Winograd F(2x2,3x3) forward / data gradient of
                  the trunk and decoder convolutions, ~30 % of the step): SURVEY 8d algorithmic FLOPs (2 MAC of the
                  direct convolution) per launch / its mean duration from hipEvents recorded on the launch stream
                  inside the timed region, against the fp32 matrix peak.  The HBM-bound fused photometric kernels
                  (BASELINE metric 2) are reported the same way under roofline.photometric;
  cpu_baseline -- the CPU oracle's full training step timed on this host's cores (N=1 only).
"""
import argparse
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(REPO, "self-supervised-depth-estimation_amd"))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s measured copy ceiling)
MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: dense fp32 matrix peak (v_mfma_f32_16x16x4_f32, 64 FLOP/clk/SIMD @2.4 GHz)


def host_cores():
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, os.cpu_count() or n, 64))


def cpu_baseline(opt, trainer, seconds_budget=25.0):
    """Time the CPU oracle (oracle/train_step.py, "port") on a BOUNDED sample of the same workload:
    a B=1 paging step, a timed B=2 probe step, then one timed step at the largest batch <= opt.batch_size
    that the probe predicts fits in ~seconds_budget.  images/s = sample batch / step time."""
    from oracle import ref_cpu as R
    from oracle.train_step import CpuTrainer
    cores = host_cores()
    torch.set_num_threads(cores)
    state = {k: {n: t.detach().cpu() for n, t in m.state_dict().items()} for k, m in trainer.models.items()}
    ct = CpuTrainer(state, R.Opt(height=opt.height, width=opt.width), opt.num_layers, opt.learning_rate)
    H, W = opt.height, opt.width

    def step(b, seed):
        inputs = R.synthetic_inputs(b, H, W, seed=seed)
        noise = R.tiebreak_noise(b, H, W)
        t0 = time.perf_counter()
        ct.train_step(inputs, noise)
        return time.perf_counter() - t0

    step(1, 2)
    t2 = step(2, 1)
    print("cpu_baseline: %d threads, B=2 probe step %.2f s" % (cores, t2), file=sys.stderr, flush=True)
    bs = int(max(1, min(opt.batch_size, seconds_budget / (t2 / 2.0))))
    t = step(bs, 0) if bs != 2 else t2
    print("cpu_baseline: B=%d step %.2f s" % (bs, t), file=sys.stderr, flush=True)
    return {"value": round(bs / t, 4), "unit": "images/s", "cores": cores, "kind": "port",
            "sample": "one full training step (fwd+bwd+Adam) of the CPU oracle at B=%d (of %d), %dx%d, resnet%d, "
                      "fp32, torch %d threads, after a B=1 and a B=2 step" % (bs, opt.batch_size, H, W,
                                                                             opt.num_layers, cores)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=12)
    ap.add_argument("--height", type=int, default=192)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--num-layers", type=int, default=18)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-noise", action="store_true", help="reference-style CPU randn tie-break noise + H2D copy")
    ap.add_argument("--miopen-find", action="store_true", help="torch.backends.cudnn.benchmark=True (MIOpen find)")
    ap.add_argument("--channels-last", action="store_true", help="run the ResNet trunks in NHWC")
    ap.add_argument("--no-overlap", action="store_true", help="pose and depth networks on one stream (A/B of overlap_streams)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    local = local % torch.cuda.device_count()          # (rehearsals with several ranks on one GPU)
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("DC_DIST_BACKEND", "nccl")     # "nccl" is RCCL on ROCm; gloo only for rehearsal
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    if args.gpus != world and rank == 0:
        print("warning: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world), file=sys.stderr)

    from depthcore import ops
    from depthcore.synthetic import synthetic_batch
    import trainer as T

    opt = T.default_options(batch_size=args.batch, height=args.height, width=args.width, num_layers=args.num_layers,
                            cpu_tiebreak_noise=args.cpu_noise, overlap_streams=not args.no_overlap)
    torch.backends.cudnn.benchmark = bool(args.miopen_find)
    tr = T.Trainer(opt, device=device, rank=rank, world_size=world)
    if args.channels_last:
        for k in ("encoder", "pose_encoder"):
            tr.models[k].to(memory_format=torch.channels_last)
    tr.set_train()
    inputs = synthetic_batch(args.batch, args.height, args.width, device, seed=100 + rank)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    loss0 = None
    for _ in range(args.warmup):
        _, losses = tr.train_step(inputs)
        loss0 = losses["loss"] if loss0 is None else loss0
    ops.profile_enable(args.steps + 8)
    # every 7th conv launch carries an event pair (7 is coprime with the 78 + 32 launches per step, so all layers are
    # sampled over the timed region); bracketing every launch costs ~4 % of the step
    ops.conv_profile_enable((args.steps + 2) * 24, 7)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        _, losses = tr.train_step(inputs)
    sync()
    dt = time.perf_counter() - t0
    prof = ops.profile_collect()
    ops.profile_enable(0)
    cprof_c, wprof_c = ops.conv_profile_collect(0), ops.conv_profile_collect(1)
    ops.conv_profile_enable(0, 1)
    # In the timed region the pose and the depth network run on two HIP streams, so a launch's event pair also spans
    # the time it shares the GPU with a kernel of the other stream.  The kernel's own duration (what rocprofv3, which
    # serialises dispatches, reports) is taken from SERIAL_STEPS single-stream steps run right after the timed region
    # with an event pair around every launch.
    SERIAL_STEPS = 3
    if tr.opt.overlap_streams:
        tr.opt.overlap_streams = False
        ops.profile_enable(SERIAL_STEPS + 2)
        ops.conv_profile_enable((SERIAL_STEPS + 1) * 160, 1)
        for _ in range(SERIAL_STEPS):
            tr.train_step(inputs)
        torch.cuda.synchronize()
        prof = ops.profile_collect()
        cprof, wprof = ops.conv_profile_collect(0), ops.conv_profile_collect(1)
        ops.profile_enable(0)
        ops.conv_profile_enable(0, 1)
        tr.opt.overlap_streams = True
        roof_src = "%d single-stream steps after the timed region, every launch" % SERIAL_STEPS
    else:
        cprof, wprof = cprof_c, wprof_c
        roof_src = "timed region, every 7th launch"
    tmax = torch.tensor([dt], dtype=torch.float64, device=device)
    lossv = losses["loss"].detach().clone().reshape(1)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        # replicas must stay in lock-step: after K identical Adam steps on averaged gradients the weights agree
        wsum = torch.stack([p.detach().double().sum() for p in tr.parameters_to_train[:8]])
        wmax, wmin = wsum.clone(), wsum.clone()
        dist.all_reduce(wmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(wmin, op=dist.ReduceOp.MIN)
        if float((wmax - wmin).abs().max()) > 1e-6 * float(wmax.abs().max() + 1):
            raise SystemExit("replicas diverged: %r vs %r" % (wmax.tolist(), wmin.tolist()))
    dt = float(tmax.item())
    loss_last = float(losses["loss"].detach())
    if not (loss_last == loss_last):
        raise SystemExit("loss is NaN")

    if rank == 0:
        N = args.batch * args.height * args.width
        bytes_fwd = sum(36.0 * N + 16.0 * (N >> (2 * s)) for s in range(4))
        bytes_bwd = sum(36.0 * N + 20.0 * (N >> (2 * s)) for s in range(4))
        bwd_ms = prof["bwd_ms"] / max(prof["bwd_launches"], 1)
        fwd_ms = prof["fwd_ms"] / max(prof["fwd_launches"], 1)
        ach = bytes_bwd / (bwd_ms * 1e-3) / 1e9 if bwd_ms > 0 else 0.0
        # HBM bytes per launch from the PMC counters: collected in separate `rocprofv3 --pmc FETCH_SIZE` /
        # `--pmc WRITE_SIZE` passes by tools/pmc_traffic.sh (FETCH_SIZE x2.0 per the gfx950 calibration on a
        # known-byte dword kernel in the same run, WRITE_SIZE x1.0) and committed under profiles/.
        traffic = {}
        tf = os.path.join(REPO, "profiles", "round1_traffic.json")
        if os.path.exists(tf) and (args.batch, args.height, args.width, args.num_layers) == (12, 192, 640, 18):
            try:
                tj = json.load(open(tf))
                traffic = {k: round(v["hbm_bytes_calibrated"], 0) for k, v in tj.items() if "hbm_bytes_calibrated" in v}
            except Exception:
                traffic = {}
        c_ms = cprof["ms"] / max(cprof["launches"], 1)
        c_tf = cprof["flops"] / (cprof["ms"] * 1e-3) / 1e12 if cprof["ms"] > 0 else 0.0
        c_ex = cprof["executed_flops"] / (cprof["ms"] * 1e-3) / 1e12 if cprof["ms"] > 0 else 0.0
        w_tf = wprof["flops"] / (wprof["ms"] * 1e-3) / 1e12 if wprof["ms"] > 0 else 0.0
        out = {
            "metric": "training images/sec at %dx%d bs%d (resnet%d depth+pose, 4-scale photometric+smoothness)"
                      % (args.height, args.width, args.batch, args.num_layers),
            "value": round(world * args.batch * args.steps / dt, 3),
            "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%sresnet%d depth+pose, %dx%d, per-GPU batch %d, 4 scales, "
                                   "frames {0,-1,+1}, automasking, Adam lr 1e-4; random-init weights"
                                   % ("BASELINE configs[1]: " if (args.num_layers, args.height, args.width, args.batch)
                                      == (18, 192, 640, 12) else
                                      "BASELINE configs[2] (per-rank): " if (args.num_layers, args.height, args.width,
                                                                             args.batch) == (50, 320, 1024, 8) else "",
                                      args.num_layers, args.height, args.width, args.batch),
                       "global_batch": world * args.batch, "parallelism": "dp%d" % world,
                       "tiebreak_noise": "cpu-randn+h2d" if args.cpu_noise else "on-device counter RNG"},
            "roofline": {"kernel": "dc::wino_ps_kernel (Winograd F(2x2,3x3) fp32-MFMA convolution: forward + data gradient of "
                                   "the trunk and decoder 3x3 convolutions)",
                         "bound": "mfma", "achieved": round(c_tf, 2), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(c_tf / MFMA_F32_PEAK_TFLOPS, 4),
                         "traffic": traffic.get("dc::wino_ps_kernel"), "traffic_source": "profiles/round1_traffic.json (mean per launch)"
                         if traffic.get("dc::wino_ps_kernel") else None,
                         "flops_definition": "SURVEY 8d algorithmic: 2 MAC of the direct 3x3 convolution, summed over the launches",
                         "algorithmic_flops_per_launch": round(cprof["flops"] / max(cprof["launches"], 1), 0),
                         "algorithmic_bytes_per_launch": round(cprof["bytes"] / max(cprof["launches"], 1), 0),
                         "avg_kernel_ms": round(c_ms, 4), "launches_timed": cprof["launches"], "measured_in": roof_src,
                         "avg_kernel_ms_in_timed_region": round(cprof_c["ms"] / max(cprof_c["launches"], 1), 4),
                         "timed_region_note": "two-stream overlap: a launch shares the GPU with the other branch's kernels "
                                              "(sampled every 7th launch); rocprofv3 serialises dispatches and matches avg_kernel_ms",
                         "note": "Winograd issues 16/36 of the algorithmic MACs to the matrix cores: `frac` follows the SURVEY 8d "
                                 "definition, `issued_frac_of_peak` is the fraction of the fp32 MFMA peak actually used",
                         "issued_to_matrix_cores_tflops": round(c_ex, 2),
                         "issued_frac_of_peak": round(c_ex / MFMA_F32_PEAK_TFLOPS, 4),
                         "wgrad_kernel": {"kernel": "dc::wino_wgrad_kernel", "achieved": round(w_tf, 2),
                                          "frac": round(w_tf / MFMA_F32_PEAK_TFLOPS, 4), "launches_timed": wprof["launches"],
                                          "issued_frac_of_peak": round(wprof["executed_flops"] / (wprof["ms"] * 1e-3) / 1e12 /
                                                                       MFMA_F32_PEAK_TFLOPS, 4) if wprof["ms"] > 0 else 0.0,
                                          "avg_kernel_ms": round(wprof["ms"] / max(wprof["launches"], 1), 4)},
                         "photometric": {"kernel": "dc::photo_bwd_kernel (fused warp+SSIM+L1+automask backward, 4 scales x 2 frames)",
                                         "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                         "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic.get("dc::photo_bwd_kernel"),
                                         "limiter": "VALU issue, not HBM: 92 M wave-level VALU instructions per launch (SQ_INSTS_VALU)",
                                         "algorithmic_bytes_per_launch": bytes_bwd, "avg_kernel_ms": round(bwd_ms, 4),
                                         "launches": prof["bwd_launches"],
                                         "fwd_kernel": {"kernel": "dc::photo_fwd_kernel", "algorithmic_bytes_per_launch": bytes_fwd,
                                                        "avg_kernel_ms": round(fwd_ms, 4),
                                                        "achieved": round(bytes_fwd / (fwd_ms * 1e-3) / 1e9, 1) if fwd_ms > 0 else 0.0}}},
            "loss_first": round(float(loss0.detach()), 6), "loss_last": round(loss_last, 6),
            "grad_bytes_allreduced_per_step": tr.buckets.nbytes if world > 1 else 0,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(opt, tr)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
