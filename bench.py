#!/usr/bin/env python3
"""Headline benchmark: training images/s of the self-supervised depth step on MI355X.

    python bench.py --gpus N --steps K --warmup W

N = 1 runs in this process.  N > 1 and no WORLD_SIZE in the environment: this process becomes a launcher that never
touches the GPU; it starts N fresh child processes of this script (one per GPU: RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_ADDR=127.0.0.1 / MASTER_PORT set), relays rank 0's JSON line and exits with the worst child exit code.  Under
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` (WORLD_SIZE already set) it is a rank.  If the
box has fewer than N GPUs the launcher refuses (exit code 2) instead of printing a mislabelled 1-GPU line.

One "step" = reference trainer.py:233-237 on one synthetic KITTI-shaped batch: process_batch (ResNet depth encoder +
decoder on frame 0, pose encoder + decoder on the pairs (-1,0),(0,+1), 4-scale fused warp + SSIM/L1 + automask +
smoothness loss) -> backward -> (RCCL all-reduce of the gradient buckets) -> Adam.  Default workload = BASELINE.json
configs[1]: resnet18, 192x640, per-GPU batch 12, fp32; `--num-layers 50 --height 320 --width 1024 --batch 8` is
configs[2] per rank.  Weak scaling: every rank processes its own batch.

Rank 0 prints ONE JSON line (contract in the task statement) with extra objects:
  roofline     -- the dominant kernel of the step (Winograd fp32-MFMA convolution): SURVEY 8d algorithmic FLOPs per launch /
                  its mean duration from hipEvents on the launch stream, against the fp32 matrix peak; the HBM-bound fused
                  photometric kernels (BASELINE metric 2) the same way under roofline.photometric, backward chain as a whole;
  phases_ms    -- forward / backward / exposed gradient exchange / Adam wall split of a step (N > 1: what scaling costs);
  cpu_baseline -- the CPU oracle's full training step timed on this host's cores (N=1 only; SURVEY 8d protocol).

Rehearsal switches (not measurements, flagged in the line): DC_DIST_BACKEND=gloo exchanges over gloo instead of RCCL;
`--oversubscribe` lets several ranks share one GPU; `--rehearse` replaces the GPU step by a CPU stand-in so that the
launcher, the rendezvous, the bucketed exchange and the max-over-ranks timing can be exercised on a machine without GPUs.
"""
import argparse
import json
import os
import socket
import statistics
import subprocess
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(REPO, "self-supervised-depth-estimation_amd"))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s measured copy ceiling)
MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: dense fp32 matrix peak (v_mfma_f32_16x16x4_f32, 64 FLOP/clk/SIMD @2.4 GHz)
MFMA_BF16_PEAK_TFLOPS = 2500.0 # MI355X_MICROARCH.md: dense bf16 matrix peak (~2.5 PF)
# kernel families that carry hipEvent pairs (include/depthcore.h: dc_conv_profile_*): id -> (name, what bounds it)
FAMILIES = {
    0: ("dc::wino_ps_kernel (Winograd F(2x2,3x3) fp32-MFMA convolution: forward + data gradient of the trunk and decoder 3x3 "
        "convolutions)", "mfma"),
    1: ("dc::wino_wgrad_kernel (Winograd-domain 3x3 weight gradient, fp32 MFMA)", "mfma"),
    2: ("dc::c3b_conv_kernel (direct 3x3 convolution on the bf16 matrix cores, fp32 tensors in HBM: forward + data gradient)", "hbm"),
    3: ("dc::c3b_wgrad_kernel (3x3 weight gradient on the bf16 matrix cores, transposed LDS reads, fp32 tensors in HBM)", "hbm"),
    4: ("dc::g1_* (1x1 convolutions as NCHW fp32-MFMA GEMMs: forward, data gradient, weight gradient)", "mfma"),
    5: ("dc::cg_* (3x3 stride-2 convolutions as implicit fp32-MFMA GEMMs: forward, data gradient split by output parity, weight "
        "gradient; main kernels, without the slab sums)", "mfma"),
    6: ("dc::stem_* (7x7 stride-2 stem, patch-staged, incl. the input normalisation and the pose pairs' concat in the loader: "
        "fp32-MFMA weight gradient; forward on bf16x3 split operands by default, counted as the fp32 GEMM it stands for)", "mfma"),
    7: ("dc::g1x3_* (1x1 convolutions as fp32-accurate GEMMs on the bf16 matrix cores: three bf16 pieces per fp32 operand, six "
        "partial products, fp32 accumulation; forward, data gradient, weight gradient)", "mfma_bf16"),
}
VALU_LANE_OPS_PEAK = 78.6e12   # 256 CUs x 4 SIMDs x 32 lanes x 2.4 GHz: wave-level VALU instructions x 64 lanes per second
# Issue slots per counted VALU instruction of the photometric kernels' main loops (tools/isa_mix.py on the gfx950 ISA of THIS
# tree's kernels, kept in profiles/round6_photo_isa_mix.txt): SQ_INSTS_VALU counts a packed-fp32 or a transcendental instruction
# once, the SIMD-32 issues them over twice the cycles (MI355X_MICROARCH.md "vector-instruction ISSUE cost").  Training forward,
# the all-the-way instantiation that is timed (photo_fwdg_kernel<false, 4, true>, two rows per loop trip): 1235 plain + 168 DPP +
# 18 lane + 2 x (342 packed + 28 transcendental) = 2161 slots per 1791 instructions; backward chain (disp_grad_kernel):
# 143 plain + 2 x (11 packed + 2 transcendental) = 169 per 156.
VALU_SLOTS_PER_INST = {"bwd": 169.0 / 156.0, "fwd": 2161.0 / 1791.0}


# ------------------------------------------------------------------------------------------------ launcher (N > 1)
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch(args, argv):
    """Parent of an N-rank run.  Must not initialise the GPU: torch.cuda.device_count() does not (task statement)."""
    n = args.gpus
    if not args.rehearse:
        have = torch.cuda.device_count()
        if have < n and not args.oversubscribe:
            print("bench.py: --gpus %d but this machine exposes %d GPU(s); refusing to print a mislabelled line "
                  "(rehearsal on fewer GPUs: DC_DIST_BACKEND=gloo python bench.py --gpus %d --oversubscribe)" % (n, have, n),
                  file=sys.stderr)
            return 2
    env = dict(os.environ)
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
    procs = []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=e,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out0, _ = procs[0].communicate()
    rcs = [procs[0].returncode]
    deadline = time.time() + 120
    for p in procs[1:]:
        try:
            rcs.append(p.wait(timeout=max(1.0, deadline - time.time())))
        except subprocess.TimeoutExpired:
            p.kill()                       # exactly the child we started
            rcs.append(p.wait())
    for line in out0.decode().splitlines():       # libraries (gloo) chat on stdout: only the JSON line is relayed there
        (sys.stdout if line.startswith("{") else sys.stderr).write(line + "\n")
    sys.stdout.flush()
    bad = [rc for rc in rcs if rc != 0]
    if bad:
        print("bench.py: rank exit codes %r" % rcs, file=sys.stderr)
        return bad[0] if bad[0] > 0 else 1
    return 0


# ------------------------------------------------------------------------------------------------ CPU baseline
def host_info():
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 1
    # a container's CPU share can be smaller than its affinity mask (cgroup quota): that, not the mask, is what "all cores" means
    quota = None
    try:
        a, b = open("/sys/fs/cgroup/cpu.max").read().split()[:2]                      # cgroup v2: "<quota|max> <period>"
        quota = None if a == "max" else float(a) / float(b)
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            quota = None if q <= 0 else q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        except (OSError, ValueError):
            pass
    if quota is not None:
        usable = max(1, min(usable, int(quota + 0.999)))
    return model, os.cpu_count() or usable, usable


def cpu_probe(threads, H, W, num_layers):
    """`bench.py --cpu-probe T`: one B=1 oracle training step at T torch threads after a B=1 warm-up (child of cpu_baseline)."""
    import trainer as T
    from oracle import ref_cpu as R
    from oracle.train_step import CpuTrainer
    torch.set_num_threads(threads)
    tr = T.Trainer(T.default_options(batch_size=1, height=H, width=W, num_layers=num_layers), device="cpu", seed=0)
    ct = CpuTrainer({k: {n: t.detach() for n, t in m.state_dict().items()} for k, m in tr.models.items()},
                    R.Opt(height=H, width=W), num_layers, 1e-4)
    for seed in (2, 8):
        inputs, noise = R.synthetic_inputs(1, H, W, seed=seed), R.tiebreak_noise(1, H, W)
        t0 = time.perf_counter()
        ct.train_step(inputs, noise)
        sec = time.perf_counter() - t0
    print("cpu_probe_seconds=%.4f" % sec, flush=True)
    return 0


def cpu_baseline(opt, trainer, timed_steps=10):
    """SURVEY 8d protocol: the CPU oracle (oracle/train_step.py, kind "port") runs the same step (fwd + bwd + Adam) from the
    GPU trainer's weights on the same kind of synthetic batch at the FULL per-rank batch: warm-up (B=1, B=2, one full batch),
    then `timed_steps` (>= 10) timed steps at 64 torch threads -> median (the headline `value`: torch's intra-op pool stops
    scaling long before 100+ threads on these convolutions, so 64 is the faster setting on the GPU boxes' hosts), then
    `allcore_steps` timed steps with one thread per usable core (reported beside it, even when slower), then one B=1 step on
    one thread.  About 60-80 s of CPU work on the GPU box's host."""
    from oracle import ref_cpu as R
    from oracle.train_step import CpuTrainer
    model, logical, usable = host_info()
    threads = max(1, min(usable, 64))
    state = {k: {n: t.detach().cpu() for n, t in m.state_dict().items()} for k, m in trainer.models.items()}
    H, W = opt.height, opt.width
    ct = CpuTrainer(state, R.Opt(height=H, width=W), opt.num_layers, opt.learning_rate)

    def step(b, seed):
        inputs = R.synthetic_inputs(b, H, W, seed=seed)
        noise = R.tiebreak_noise(b, H, W)
        t0 = time.perf_counter()
        ct.train_step(inputs, noise)
        return time.perf_counter() - t0

    def note(msg):
        print("cpu_baseline: " + msg, file=sys.stderr, flush=True)     # one line per step: a silent minute reads as a hang

    torch.set_num_threads(threads)
    step(1, 2)                                      # pages the oracle in
    t2 = step(2, 1)
    bs = opt.batch_size
    note("%s, %d logical / %d usable cores, %d threads; B=2 probe %.2f s" % (model, logical, usable, threads, t2))
    tw = step(bs, 3)                                # warm-up at the timed size; also sizes the sample
    # >= 10 timed steps wherever they fit ~75 s; a host much slower than the GPU boxes' keeps the run bounded
    nsteps = timed_steps if tw * timed_steps <= 75.0 else max(3, int(75.0 / tw))
    note("B=%d warm-up step %.2f s -> %d timed steps" % (bs, tw, nsteps))
    ts = []
    for i in range(nsteps):
        ts.append(step(bs, 10 + i))
        note("step %d/%d %.2f s" % (i + 1, nsteps, ts[-1]))
    allc = None
    if usable > threads:
        # One thread per usable core, reported beside the headline even when slower.  torch's intra-op pool stops scaling long
        # before 100+ threads on these convolutions, and where the CPU share is smaller than the affinity mask an oversubscribed
        # pool can take MINUTES for one step: the probe runs in a child process with a time limit (`--cpu-probe`), one B=1 step
        # after a B=1 warm-up, and is compared with the same B=1 step at the headline's thread count
        tb1 = step(1, 8)
        note("B=1 step at %d threads %.2f s; probing %d threads in a child process (limit 60 s)" % (threads, tb1, usable))
        allc = {"cores": usable, "unit": "images/s", "sample_batch": 1, "timed_steps": 1,
                "same_step_at_%d_threads_images_per_s" % threads: round(1.0 / tb1, 4)}
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-probe", str(usable), "--height", str(H), "--width", str(W),
                                "--num-layers", str(opt.num_layers)], capture_output=True, text=True, timeout=60)
            sec = float([ln for ln in r.stdout.splitlines() if ln.startswith("cpu_probe_seconds=")][-1].split("=")[1])
            allc.update({"value": round(1.0 / sec, 4), "step_seconds_median": round(sec, 3)})
        except subprocess.TimeoutExpired:
            allc.update({"value": None, "note": "one B=1 step at %d threads did not finish within 60 s (oversubscribed pool): the "
                                                "%d-thread figure is the faster setting on this host" % (usable, threads)})
        except Exception as e:                       # noqa: BLE001 -- the baseline must not fail the bench line
            allc.update({"value": None, "note": "probe failed: %s" % e})
        note("all %d cores: %s" % (usable, allc))
    torch.set_num_threads(1)
    t1 = step(1, 4)
    torch.set_num_threads(threads)
    med, best = statistics.median(ts), min(ts)
    note("steps %s s; all %d cores: %s; 1 thread B=1 %.2f s" % (["%.2f" % t for t in ts], usable, allc, t1))
    return {"value": round(bs / med, 4), "unit": "images/s", "cores": threads, "kind": "port",
            "value_best": round(bs / best, 4), "step_seconds_median": round(med, 3), "step_seconds_min": round(best, 3),
            "timed_steps": nsteps, "sample_batch": bs, "cpu_model": model, "host_logical_cores": logical,
            "host_usable_cores": usable, "all_usable_cores": allc,
            "one_thread": {"value": round(1.0 / t1, 4), "unit": "images/s", "cores": 1, "sample": "one B=1 step"},
            "sample": "%d timed full training steps (fwd+bwd+Adam; median) of the CPU oracle at B=%d (the full per-rank batch), "
                      "%dx%d, resnet%d, fp32, torch intra-op threads = %d, after a B=1, a B=2 and one B=%d warm-up step; "
                      "`all_usable_cores`: one B=1 step at one thread per usable core (child process, 60 s limit) beside the same "
                      "step at %d threads" % (nsteps, bs, H, W, opt.num_layers, threads, bs, threads)}


# ------------------------------------------------------------------------------------------------ rehearsal stand-in
def rehearse(args, world, rank):
    """CPU stand-in for the GPU step: same process layout, rendezvous, bucketed exchange (depthcore/ddp.py over gloo),
    barrier + max-over-ranks timing and JSON relay -- NOT a measurement of anything."""
    import torch.distributed as dist
    import torch.nn as nn
    from depthcore.ddp import GradBuckets, broadcast_parameters
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    net = nn.Sequential(nn.Conv2d(3, 16, 3, padding=1), nn.ReLU(), nn.Conv2d(16, 16, 3, padding=1), nn.ReLU(),
                        nn.Conv2d(16, 1, 3, padding=1))
    if world > 1:
        broadcast_parameters([net], 0)
    gb = GradBuckets([("net." + n, p) for n, p in net.named_parameters()], 0.001, world)
    optim = torch.optim.Adam(net.parameters(), 1e-4)
    x = torch.rand(args.batch, 3, 32, 64, generator=torch.Generator().manual_seed(100 + rank))

    def step():
        loss = net(x).square().mean()
        gb.zero()
        loss.backward()
        gb.finish()
        optim.step()
        return loss

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    if world > 1:
        dist.barrier()
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    wsum = torch.stack([p.detach().double().sum() for p in net.parameters()])
    if world > 1:
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
        lo, hi = wsum.clone(), wsum.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        if float((hi - lo).abs().max()) > 1e-9 * float(hi.abs().max() + 1):
            raise SystemExit("replicas diverged")
    if rank == 0:
        print(json.dumps({"metric": "REHEARSAL (CPU stand-in step, not a measurement)", "value": round(world * args.batch * args.steps / float(dt), 3),
                          "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(float(dt) / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "f32", "data": "synthetic", "rehearsal": True,
                          "config": {"workload": "rehearsal stand-in", "global_batch": world * args.batch, "parallelism": "dp%d" % world},
                          "grad_bytes_allreduced_per_step": gb.nbytes if world > 1 else 0, "loss_last": round(float(loss), 6)}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


# ------------------------------------------------------------------------------------------------ a rank
def _traffic_for(cfg_key):
    """HBM bytes per launch from the PMC passes kept under profiles/ (separate `rocprofv3 --pmc FETCH_SIZE` / `--pmc
    WRITE_SIZE` runs, tools/pmc_traffic.sh; FETCH_SIZE x2.0 per the gfx950 calibration, WRITE_SIZE x1.0).  They are
    CITED, not measured in this run: a PMC pass serialises the step and cannot share a process with the timed region."""
    for name in ("round6_traffic_%s.json" % cfg_key, "round5_traffic_%s.json" % cfg_key, "round4_traffic_%s.json" % cfg_key):
        if not name:
            continue
        tf = os.path.join(REPO, "profiles", name)
        if os.path.exists(tf):
            try:
                tj = json.load(open(tf))
                return {k: v for k, v in tj.items() if isinstance(v, dict)}, "cited from profiles/%s" % name
            except Exception:
                pass
    return {}, None


def run_rank(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch `python bench.py --gpus N` or torchrun with "
                         "--nproc-per-node N)" % (args.gpus, world))
    if args.rehearse:
        return rehearse(args, world, rank)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    ndev = torch.cuda.device_count()
    if world > ndev and not args.oversubscribe:
        raise SystemExit("bench.py: %d ranks but %d GPU(s) (add --oversubscribe for a rehearsal)" % (world, ndev))
    local = local % ndev
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    import torch.distributed as dist
    backend = os.environ.get("DC_DIST_BACKEND", "nccl")         # "nccl" is RCCL on ROCm; gloo only for rehearsal
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from depthcore import ops
    from depthcore.synthetic import synthetic_batch
    import trainer as T

    front = {}
    if args.front == "gru":          # BASELINE configs[3]: one sequence of --len-sequence frames per rank (batch size 1)
        front = dict(gru="v5", len_sequence=args.len_sequence)
        args.batch = 1
    elif args.front == "fusion":     # BASELINE configs[4]: frames [-2, -1, 0] stacked through encoder + decoder, then Fusion_v3
        front = dict(fusion="v3", frame_ids=[0, -2, -1, 1])
    opt = T.default_options(batch_size=args.batch, height=args.height, width=args.width, num_layers=args.num_layers,
                            nets_dtype=args.nets_dtype,
                            cpu_tiebreak_noise=args.cpu_noise, overlap_streams=not args.no_overlap, step_priority=int(args.step_priority),
                            wino_weight_cache=not args.no_wino_cache, hip_graph=bool(args.graph),
                            wgrad_lanes=int(args.wgrad_lanes), **front)
    if args.bucket_mb > 0:
        opt.bucket_mb = args.bucket_mb
    tr = T.Trainer(opt, device=device, rank=rank, world_size=world)
    tr.set_train()
    # the training loop -- everything below -- runs on the trainer's step stream (high priority for the depth branch: opt.step_priority);
    # entered once and left at process exit (the per-loop hand-over of Trainer.on_step_stream)
    _step_ctx = tr.on_step_stream()
    _step_ctx.__enter__()
    dist_info = None
    if world > 1:        # who is really in the group: every rank reports (rank, device ordinal, device name)
        mine = (rank, local, torch.cuda.get_device_name(device))
        everyone = [None] * world
        dist.all_gather_object(everyone, mine)
        dist_info = {"ranks": dist.get_world_size(), "devices": [list(e) for e in sorted(everyone)]}
    if args.front == "gru":
        from depthcore.synthetic import synthetic_sequence_batch
        inputs = synthetic_sequence_batch(args.len_sequence, args.height, args.width, device, seed=100 + rank)
    elif args.front == "fusion":
        inputs = synthetic_batch(args.batch, args.height, args.width, device, seed=100 + rank, frame_ids=(0, -2, -1, 1),
                                 packed=not args.no_packed_inputs)
    else:
        inputs = synthetic_batch(args.batch, args.height, args.width, device, seed=100 + rank, packed=not args.no_packed_inputs)
    imgs_per_step = args.len_sequence if args.front == "gru" else args.batch        # target frames that get a loss per rank and step

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # torch warns (once) when an AccumulateGrad node meets a gradient produced on another stream than the one the node was
    # created on: record in WHICH phase of this run that first happens -- inside the timed region it would mean cross-stream
    # waits in the measured step; in the diagnostic phases after it (which switch the stream layout on purpose) it is expected
    import warnings
    phase = ["warmup"]
    accgrad_phase = []
    _show = warnings.showwarning

    def _showwarning(message, category, filename, lineno, file=None, line=None):
        if "AccumulateGrad" in str(message):
            accgrad_phase.append(phase[0])
        _show(message, category, filename, lineno, file, line)
    warnings.showwarning = _showwarning
    loss0 = None
    # set-up steps BEFORE the contract's W warm-up steps (untimed, like building the trainer): the first steps of a process size
    # the workspaces, fill the allocator's pools and meet the weight-cache variants (refreshed from the second step on); with a
    # small W they would otherwise reach into the timed region (and the clocks of a GPU that was idle a moment ago are still ramping:
    # the first 20-step window after 3 + 5 steps read 11.26 / 11.58 ms where the following ones read 11.20-11.25).  config.setup_steps.
    # Default 0 since round 6: the headline follows the contract's protocol -- W warm-up steps, then K timed ones --
    # as BENCH_r01..r04 did; the steady state is reported beside it (`steady_state_ms_per_step`: median of the later windows).
    SETUP_STEPS = int(os.environ.get("DC_BENCH_SETUP", "0"))
    for _ in range(SETUP_STEPS):
        _, losses = tr.train_step(inputs)
        loss0 = losses["loss"].detach().clone() if loss0 is None else loss0
    for _ in range(args.warmup):
        _, losses = tr.train_step(inputs)
        loss0 = losses["loss"].detach().clone() if loss0 is None else loss0
    if not tr.graph_enabled:                    # (event pairs cannot be part of a captured step)
        ops.profile_enable(args.steps + 8)
        # every 7th conv launch carries an event pair (7 is coprime with the launches per step, so all layers are
        # sampled over the timed region); bracketing every launch costs ~4 % of the step
        ops.conv_profile_enable((args.steps * args.windows + 2) * 60, 7)
    sync()
    phase[0] = "timed"
    t0 = time.perf_counter()
    host_s = 0.0                                # time spent INSIDE train_step(): launch work of an eager step, one replay of a captured one
    for _ in range(args.steps):
        th = time.perf_counter()
        _, losses = tr.train_step(inputs)
        host_s += time.perf_counter() - th
    sync()
    dt = time.perf_counter() - t0
    loss_last = float(losses["loss"].detach())
    # run-to-run spread: more windows of the same K steps right after the timed one (reported, not the headline)
    windows = [dt / args.steps * 1e3]
    for _w in range(args.windows - 1):
        sync()
        tw = time.perf_counter()
        for _ in range(args.steps):
            _, losses = tr.train_step(inputs)
        sync()
        windows.append((time.perf_counter() - tw) / args.steps * 1e3)
    phase[0] = "diagnostics-after-the-timed-region"
    del losses, _                       # drop the autograd graph before the stream layout changes below
    graphed = tr.graph_enabled and tr._graph is not None
    tr.graph_enabled = False            # the diagnostic steps below are eager
    prof = ops.profile_collect()
    ops.profile_enable(0)
    fam_c = {k: ops.conv_profile_collect(k) for k in FAMILIES}
    ops.conv_profile_enable(0, 1)

    # ---- wall split of a step: forward / backward / exposed exchange / Adam (host-synchronised, after the timed region)
    # per-bucket timeline of the exchange (world > 1): one pipelined eager step with event pairs around every bucket's all-reduce
    bucket_order, bucket_offsets_ms = [], None
    if world > 1:
        from depthcore.ddp import bucket_timeline
        tr.buckets.timing = True
        _o, _l = tr.process_batch(inputs)
        tr.buckets.zero()
        _l["loss"].backward()
        tr.buckets.finish()
        tr.model_optimizer.step()
        tr.step += 1
        torch.cuda.synchronize()
        tl = bucket_timeline(tr.buckets)
        tr.buckets.timing = False
        bucket_order = [b for b, _, _, _ in tl]
        bucket_offsets_ms = [{"bucket": b, "bytes": nb, "starts_ms_after_backward_start": t0, "all_reduce_ms": d} for b, nb, t0, d in tl]
        del _o, _l
    PH = 4
    ph = [0.0, 0.0, 0.0, 0.0]
    for _ in range(PH):
        ts = [time.perf_counter()]

        def mark():
            torch.cuda.synchronize()
            ts.append(time.perf_counter())
        outputs, losses = tr.process_batch(inputs); mark()
        tr.buckets.zero()
        losses["loss"].backward(); mark()
        tr.buckets.finish(); mark()
        tr.model_optimizer.step(); mark()
        tr.step += 1
        for i in range(4):
            ph[i] += (ts[i + 1] - ts[i]) * 1e3 / PH
        del outputs, losses
    packed = tr.buckets.packed
    # what the step would pay if the data loader did NOT deliver the pixel-interleaved copies: three dc_pack_rgbx launches
    rgbx_pack_ms = None
    if ("color_packed", 0, 0) in inputs:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        frames = [inputs[("color", f, 0)] for f in (0, -1, 1)]
        for _ in range(3):
            keep = [ops.pack_rgbx(fr) for fr in frames]
        e0.record()
        for _ in range(20):
            keep = [ops.pack_rgbx(fr) for fr in frames]
        e1.record()
        torch.cuda.synchronize()
        rgbx_pack_ms = e0.elapsed_time(e1) / 20.0
        del keep

    # In the timed region the pose and the depth network run on two HIP streams, so a launch's event pair also spans
    # the time it shares the GPU with a kernel of the other stream.  The kernel's own duration (what rocprofv3, which
    # serialises dispatches, reports) is taken from SERIAL_STEPS single-stream steps run right after the timed region
    # with an event pair around every launch.
    SERIAL_STEPS = 3
    if tr.opt.overlap_streams:
        tr.opt.overlap_streams = False
        lanes, tr.wgrad_lanes = tr.wgrad_lanes, False
        ops.profile_enable(SERIAL_STEPS + 2)
        ops.conv_profile_enable((SERIAL_STEPS + 1) * 400, 1)
        for _ in range(SERIAL_STEPS):
            tr.train_step(inputs)
        torch.cuda.synchronize()
        prof = ops.profile_collect()
        fam = {k: ops.conv_profile_collect(k) for k in FAMILIES}
        ops.profile_enable(0)
        ops.conv_profile_enable(0, 1)
        tr.opt.overlap_streams = True
        tr.wgrad_lanes = lanes
        roof_src, fam_steps = "%d single-stream steps after the timed region, every launch" % SERIAL_STEPS, SERIAL_STEPS
    else:
        fam = fam_c
        roof_src, fam_steps = "timed region, every 7th launch", args.steps * args.windows / 7.0
    tmax = torch.tensor([dt], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        # replicas must stay in lock-step: after K identical Adam steps on averaged gradients ALL weights agree
        wsum = torch.stack([p.detach().double().sum() for p in tr.parameters_to_train])
        wmax, wmin = wsum.clone(), wsum.clone()
        dist.all_reduce(wmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(wmin, op=dist.ReduceOp.MIN)
        if float((wmax - wmin).abs().max()) > 1e-6 * float(wmax.abs().max() + 1):
            bad = int((wmax - wmin).abs().argmax())
            raise SystemExit("replicas diverged at parameter %d: %r vs %r" % (bad, float(wmax[bad]), float(wmin[bad])))
    dt = float(tmax.item())
    if not (loss_last == loss_last):
        raise SystemExit("loss is NaN")

    from depthcore import _lib as _dcl
    split_on = bool(_dcl.lib().dc_get_gemm_split()) and fam.get(7, {}).get("launches", 0) > 0
    if rank == 0:
        cfg = (args.num_layers, args.height, args.width, args.batch)
        cfg_key = {(18, 192, 640, 12): "c2", (50, 320, 1024, 8): "c3"}.get(cfg, "other") if args.front == "none" else "other"
        if args.front == "fusion" and cfg == (18, 192, 640, 12):
            cfg_key = "c5bf16" if args.nets_dtype == "bf16" else "c5"
        if args.front == "none" and args.nets_dtype != "f32":
            cfg_key = "other"
        label = {"c2": "BASELINE configs[1]: ", "c3": "BASELINE configs[2] (per rank): "}.get(cfg_key, "")
        if args.front == "gru":
            label = "BASELINE configs[3] (per rank): ConvGRU v5 temporal fusion, one sequence of %d frames, " % args.len_sequence
        elif args.front == "fusion":
            label = "BASELINE configs[4] (per rank, %s): Fusion_v3 attention fusion on frames {-2,-1,0}, " % (
                "fp32" if args.nets_dtype == "f32" else "networks' convolutions on the bf16 matrix cores, everything else fp32")
        N = args.batch * args.height * args.width
        bytes_fwd = sum(36.0 * N + 16.0 * (N >> (2 * s)) for s in range(4))
        bytes_bwd = sum(36.0 * N + 20.0 * (N >> (2 * s)) for s in range(4))
        nb, nf = max(prof["bwd_launches"], 1), max(prof["fwd_launches"], 1)
        bwd_ms, fwd_ms = prof["bwd_ms"] / nb, prof["fwd_ms"] / nf
        bwd_chain_ms, fwd_chain_ms = prof["bwd_chain_ms"] / nb, prof["fwd_chain_ms"] / nf
        ach = bytes_bwd / (bwd_chain_ms * 1e-3) / 1e9 if bwd_chain_ms > 0 else 0.0
        traffic, traffic_src = _traffic_for(cfg_key)

        def tr_bytes(k):
            v = traffic.get(k)
            return round(v["hbm_bytes_calibrated"], 0) if v and "hbm_bytes_calibrated" in v else None

        def valu(k):
            v = traffic.get(k)
            return v.get("sq_insts_valu") if v else None
        # every instrumented kernel family of the step; the roofline entry is the one that takes the most GPU time
        def family_entry(k):
            d, dc_ = fam[k], fam_c[k]
            if d["launches"] == 0 or d["ms"] <= 0:
                return None
            name, bound = FAMILIES[k]
            sec = d["ms"] * 1e-3
            tf, ex = d["flops"] / sec / 1e12, d["executed_flops"] / sec / 1e12
            e = {"kernel": name, "bound": bound, "family": k,
                 "ms_per_step": round(d["ms"] / fam_steps, 3), "launches_per_step": round(d["launches"] / fam_steps, 1),
                 "avg_kernel_ms": round(d["ms"] / d["launches"], 4), "launches_timed": d["launches"],
                 "algorithmic_flops_per_launch": round(d["flops"] / d["launches"], 0),
                 "algorithmic_bytes_per_launch": round(d["bytes"] / d["launches"], 0)}
            if bound == "mfma":
                # `achieved` / `frac` = what the matrix pipe does (<= 1): FLOPs actually issued to the MFMA units.  For the
                # Winograd kernels that is 16/36 of SURVEY 8d's algorithmic count (2 MAC of the direct convolution), which is
                # kept beside it as the direct-convolution equivalent; for a direct GEMM the two are the same number
                e.update({"achieved": round(ex, 2), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                          "frac": round(ex / MFMA_F32_PEAK_TFLOPS, 4),
                          "flops_definition": "FLOPs issued to the matrix cores (Winograd F(2x2,3x3): 16/36 of the direct "
                                              "convolution's 2 MAC; GEMMs: all of them), summed over the launches",
                          "issued_frac_of_peak": round(ex / MFMA_F32_PEAK_TFLOPS, 4),
                          "direct_conv_equivalent_tflops": round(tf, 2),
                          "direct_conv_equivalent_frac": round(tf / MFMA_F32_PEAK_TFLOPS, 4)})
            elif bound == "mfma_bf16":
                # split operands: `achieved` / `frac` = the bf16 matrix FLOPs actually issued (six products per fp32 multiply, padded
                # tiles) against the dense bf16 peak; the fp32 GEMM it stands for (SURVEY 8d's 2 MAC) beside it
                e.update({"bound": "mfma", "achieved": round(ex, 1), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                          "frac": round(ex / MFMA_BF16_PEAK_TFLOPS, 4),
                          "flops_definition": "bf16 matrix FLOPs issued: 6 partial products per fp32 multiply-add, padded tiles",
                          "fp32_equivalent_tflops": round(tf, 2), "fp32_equivalent_frac_of_fp32_matrix_peak": round(tf / MFMA_F32_PEAK_TFLOPS, 4)})
            else:
                gbs = d["bytes"] / sec / 1e9
                e.update({"achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                          "bytes_definition": "each fp32 operand tensor and the fp32 result once, per launch",
                          "matrix_core_tflops_bf16": round(ex, 1), "matrix_core_frac_of_bf16_peak": round(ex / MFMA_BF16_PEAK_TFLOPS, 4)})
            if dc_["launches"]:
                e["avg_kernel_ms_in_timed_region"] = round(dc_["ms"] / dc_["launches"], 4)
            return e
        fams = [e for e in (family_entry(k) for k in FAMILIES) if e]
        fams.sort(key=lambda e: -e["ms_per_step"])
        dom = dict(fams[0]) if fams else {"kernel": None, "bound": "mfma", "achieved": 0.0, "peak": MFMA_F32_PEAK_TFLOPS,
                                          "unit": "TFLOP/s", "frac": 0.0}
        dom_key = {0: "dc::wino_ps_kernel", 1: "dc::wino_wgrad_kernel", 2: "dc::c3b_conv_kernel", 3: "dc::c3b_wgrad_kernel", 4: "dc::g1_*",
                   5: "dc::cg_*", 6: "dc::stem_*", 7: "dc::g1x3_*"}.get(dom.get("family"))
        # ---- BASELINE metric 2: the fused warp + SSIM + smoothness kernels against HBM.  Round 4 moved the SSIM derivative, its
        # transposed 3x3 spread and the contraction with d(warped)/d(coords) into the TRAINING FORWARD; round 5 lets it go all the
        # way (dc::photo_fwdg_kernel<.., FULL>: Project3D / BackprojectDepth / disp_to_depth backward and the pose sums in the
        # same row march, ONE float per pixel and scale out), so the backward chain is dc::disp_grad_kernel alone (transposed
        # upsample + smoothness gradient + pose reduction, scaled by the step's upstream weights).  The honest figure is the
        # PAIR: SURVEY 8d's algorithmic bytes of forward + backward over the time of both launch chains.
        def chain(bytes_, ms):
            gbs = bytes_ / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
            return {"algorithmic_bytes_per_launch": bytes_, "avg_chain_ms": round(ms, 4), "achieved": round(gbs, 1),
                    "frac": round(gbs / HBM_PEAK_GBS, 4)}

        def kern(name, ms, bytes_, slots):
            v = valu(name)
            e = {"kernel": name, "avg_kernel_ms": round(ms, 4), "traffic": tr_bytes(name), "valu_wave_insts": v}
            if v and ms > 0:
                floor_s = v * slots * 64.0 / VALU_LANE_OPS_PEAK
                e.update({"valu_issue_floor_ms": round(floor_s * 1e3, 4), "valu_issue_frac": round(floor_s / (ms * 1e-3), 4),
                          "hbm_frac_at_valu_floor": round(bytes_ / floor_s / 1e9 / HBM_PEAK_GBS, 4)})
            return e
        pair_ms = fwd_chain_ms + bwd_chain_ms
        pair = chain(bytes_fwd + bytes_bwd, pair_ms)
        photometric = dict(pair, **{
            "kernel": "fused warp + SSIM + L1 + automask + smoothness, forward AND backward launch chains of a step (4 scales x 2 "
                      "frames): identity + smooth + dc::photo_fwdg_kernel (contracts to d loss / d upsampled disp + pose sums) + "
                      "finalize | dc::disp_grad_kernel (transposed upsample, smoothness gradient, pose-gradient reduction)",
            "bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "traffic": (tr_bytes("dc::identity_kernel") or 0) + (tr_bytes("dc::photo_fwdg_kernel") or 0) + (tr_bytes("dc::smooth_fwd_kernel") or 0)
                       + (tr_bytes("dc::photo_bwdg_kernel") or 0) + (tr_bytes("dc::disp_grad_kernel") or 0)
                       if tr_bytes("dc::photo_fwdg_kernel") else None,
            "traffic_source": traffic_src, "launches": prof["bwd_launches"],
            "bytes_definition": "SURVEY 8d: forward sum_s 36 N + 16 n_s, backward sum_s 36 N + 20 n_s (fp32, N = B H W)",
            # (only the PAIR is priced against SURVEY 8d's bytes: since round 4 the backward's window work runs in the forward, so a
            # per-chain fraction would credit the backward with bytes it no longer moves)
            "forward_chain": dict({"avg_chain_ms": round(fwd_chain_ms, 4)}, **kern("dc::photo_fwdg_kernel", fwd_ms, bytes_fwd + bytes_bwd, VALU_SLOTS_PER_INST["fwd"])),
            "backward_chain": dict({"avg_chain_ms": round(bwd_chain_ms, 4)}, **kern("dc::disp_grad_kernel", bwd_chain_ms, bytes_fwd + bytes_bwd, VALU_SLOTS_PER_INST["bwd"])),
            "valu_wave_insts_per_step": (sum(valu(k) or 0 for k in ("dc::identity_kernel", "dc::smooth_fwd_kernel", "dc::photo_fwdg_kernel",
                                                                     "dc::finalize_kernel", "dc::photo_bwdg_kernel", "dc::disp_grad_kernel"))
                                         if valu("dc::photo_fwdg_kernel") else None),
            "round3": {"forward_chain_ms": 0.17, "backward_chain_ms": 0.2487, "pair_frac": 0.148,
                       "note": "forward without gradient emission + the window backward (profiles/round3_c2_bench_n1.json)"},
            "limiter": "the training forward is VALU-issue bound (3 waves per SIMD at 167 VGPRs: loads issued behind stage B, the "
                       "projection backward and the 18 pose accumulators in stage C); the transposed upsample is latency / HBM bound",
            "valu_note": "SQ_INSTS_VALU per launch (cited PMC pass) x issue slots per instruction from the ISA mix "
                         "(profiles/round6_photo_isa_mix.txt); 2 cycles per slot per SIMD-32"})
        out = {
            "metric": "training images/sec at %dx%d bs%d (resnet%d depth+pose, 4-scale photometric+smoothness)"
                      % (args.height, args.width, args.batch, args.num_layers),
            "value": round(world * imgs_per_step * args.steps / dt, 3),
            "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": ("f32" if not split_on else "f32 (1x1 convolutions: bf16x3 split operands, fp32 accumulate)") if args.nets_dtype == "f32"
                     else "bf16-nets/f32-loss", "data": "synthetic",
            "windows_ms_per_step": [round(w, 3) for w in windows],
            "steady_state_ms_per_step": round(sorted(windows[1:])[len(windows[1:]) // 2], 3) if len(windows) > 1 else None,
            "config": {"workload": "%sresnet%d depth+pose, %dx%d, per-GPU batch %d, 4 scales, "
                                   "frames {0,-1,+1}, automasking, Adam lr 1e-4; random-init weights"
                                   % (label, args.num_layers, args.height, args.width, args.batch),
                       "global_batch": world * args.batch, "parallelism": "dp%d" % world,
                       "tiebreak_noise": "cpu-randn+h2d" if args.cpu_noise else "on-device counter RNG",
                       "inputs": "resident in HBM before the timed region; " + (
                           "as the device data step delivers them: planar (\"color\", f, s) / (\"color_aug\", f, s) plus the "
                           "pixel-interleaved RGBx copy (\"color_packed\", f, 0) of the three loss frames (dc_data_to_rgbx)"
                           if ("color_packed", 0, 0) in inputs else "planar tensors only (the loss repacks the three frames per step)"),
                       "rgbx_pack_ms_if_done_inside_the_step": round(rgbx_pack_ms, 4) if rgbx_pack_ms is not None else None,
                       "setup_steps": SETUP_STEPS,
                       "step_launch": "one hipGraph replay per step" if graphed else "eager (one launch per kernel)",
                       "streams": (("depth and pose branches on two HIP streams" + ("; the depth branch's is a high-priority stream (opt.step_priority)"
                                                                                     if tr._main_stream is not None and not graphed else ""))
                                   if tr.opt.overlap_streams else "one HIP stream")
                                  + ("; weight-gradient kernels on a companion stream of each (opt.wgrad_lanes)" if tr.wgrad_lanes else "")},
            "roofline": dict(dom, **{
                         # HBM bytes per launch from the PMC counters: CITED from the separate `rocprofv3 --pmc` passes kept under
                         # profiles/ (a counter pass serialises the step and cannot share a process with the timed region)
                         "traffic": tr_bytes(dom_key) if dom_key else None, "traffic_cited": bool(dom_key and tr_bytes(dom_key)),
                         "traffic_source": traffic_src if dom_key and tr_bytes(dom_key) else None,
                         "measured_in": roof_src,
                         "selection": "the instrumented kernel family with the largest GPU time per step; all of them under `families`",
                         "timed_region_note": "two-stream overlap: a launch shares the GPU with the other branch's kernels "
                                              "(sampled every 7th launch); rocprofv3 serialises dispatches and matches avg_kernel_ms",
                         "note": "`frac` is the fraction of the fp32 MFMA peak the pipe actually delivers (issued FLOPs); the Winograd "
                                 "kernels issue 16/36 of SURVEY 8d's algorithmic MACs, so their direct-convolution equivalent "
                                 "(`direct_conv_equivalent_frac`) is 2.25x that and may exceed 1",
                         "families": fams[1:],
                         "photometric": photometric}),
            "host_enqueue_ms_per_step": round(host_s / args.steps * 1e3, 3),
            "accumulate_grad_stream_warning_first_seen_in": accgrad_phase[0] if accgrad_phase else None,
            "phases_ms": {"forward": round(ph[0], 3), "backward_incl_overlapped_exchange": round(ph[1], 3),
                          "exposed_exchange_wait": round(ph[2], 3), "adam": round(ph[3], 3),
                          "note": "host-synchronised between phases (slower than the pipelined step); %d steps after the "
                                  "timed region" % PH},
            "loss_first": round(float(loss0), 6), "loss_last": round(loss_last, 6),
            "grad_bytes_allreduced_per_step": tr.buckets.nbytes if world > 1 else 0,
            "grad_buckets": len(tr.buckets.buckets) if world > 1 else 0,
            "grads_packed_by_copy_per_step": packed if world > 1 else 0,
            "dist_backend": (backend if world > 1 else None),
            # what the exchange actually ran on (checkable against the driver's launch): ranks the process group connected, the
            # device ordinal of every rank, the buckets in launch order with their bytes, and the communicator settings in effect
            "rccl_ranks": dist_info["ranks"] if world > 1 else None,
            "rank_devices": dist_info["devices"] if world > 1 else None,
            "grad_bucket_bytes_in_launch_order": [tr.buckets.flat[b].numel() * 4 for b in bucket_order] if world > 1 else None,
            "grad_bucket_launch_ms_after_backward_start": bucket_offsets_ms if world > 1 else None,
            "rccl_env": {k: os.environ[k] for k in sorted(os.environ) if k.startswith(("NCCL_", "RCCL_"))} if world > 1 else None,
        }
        if world > 1 and (backend != "nccl" or args.oversubscribe):
            out["rehearsal"] = "backend %s%s: not an RCCL/xGMI measurement" % (backend, ", ranks share GPUs" if args.oversubscribe else "")
        if world == 1 and not args.no_cpu_baseline and args.front == "none":
            out["cpu_baseline"] = cpu_baseline(opt, tr)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


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
    ap.add_argument("--cpu-probe", type=int, default=0, help="internal: child process of cpu_baseline (one B=1 oracle step at N threads)")
    ap.add_argument("--cpu-noise", action="store_true", help="reference-style CPU randn tie-break noise + H2D copy")
    ap.add_argument("--front", choices=["none", "gru", "fusion"], default="none",
                    help="sequence front-end: gru = ConvGRU v5 (configs[3], batch 1 x --len-sequence frames), fusion = Fusion_v3 (configs[4])")
    ap.add_argument("--len-sequence", type=int, default=3)
    ap.add_argument("--nets-dtype", choices=["f32", "bf16"], default="f32",
                    help="bf16: the reduced-precision-networks policy of BASELINE configs[4] (convolution operands rounded to bf16 "
                         "for the matrix cores, fp32 accumulate; tensors, master weights, BatchNorm and the loss stay fp32); "
                         "reported under its own dtype, never the headline")
    ap.add_argument("--windows", type=int, default=4, help="timed windows of --steps steps: the first is the reported value, the "
                                                            "others show the run-to-run spread (windows_ms_per_step)")
    ap.add_argument("--graph", action="store_true", help="capture the training step (with world > 1: incl. the RCCL exchange) in one hipGraph and replay it (opt.hip_graph)")
    ap.add_argument("--no-wino-cache", action="store_true", help="per-launch Winograd weight transforms (A/B of wino_weight_cache)")
    ap.add_argument("--no-packed-inputs", action="store_true",
                    help="inputs without the data step's pixel-interleaved RGBx copies of the three loss frames (a reference data "
                         "loader's batch): the photometric forward then repacks them at every step")
    ap.add_argument("--wgrad-lanes", type=int, default=2, help="weight-gradient kernels on companion streams (opt.wgrad_lanes: 0 off, 1 on, 2 auto)")
    ap.add_argument("--no-overlap", action="store_true", help="pose and depth networks on one stream (A/B of overlap_streams)")
    ap.add_argument("--step-priority", type=int, default=2, choices=[-1, 0, 2], help="the training loop (depth branch) on a high-priority stream: -1 on, 0 off, 2 auto (opt.step_priority; Trainer.on_step_stream)")
    ap.add_argument("--oversubscribe", action="store_true", help="rehearsal: let ranks share GPUs (use with DC_DIST_BACKEND=gloo)")
    ap.add_argument("--rehearse", action="store_true", help="rehearsal: CPU stand-in step over gloo (launcher / exchange plumbing only)")
    ap.add_argument("--rccl-algo", default=None, help="world > 1: NCCL_ALGO for the gradient exchange (Ring | Tree; RCCL's own choice "
                    "when absent).  xGMI is point-to-point (7 links per GPU): DESIGN 4 'exchange budget' prices ring vs direct")
    ap.add_argument("--rccl-proto", default=None, help="world > 1: NCCL_PROTO (Simple | LL | LL128)")
    ap.add_argument("--rccl-channels", type=int, default=0, help="world > 1: NCCL_MIN_NCHANNELS (0: RCCL's default)")
    ap.add_argument("--bucket-mb", type=float, default=0.0, help="world > 1: gradient bucket size in MB (0: opt.bucket_mb's default, 32)")
    args = ap.parse_args()
    # communicator settings are environment variables read when the process group connects: set them here, before any rank starts
    # (the launcher's children inherit them; under torchrun every rank parses the same flags before init_process_group)
    for flag, var in ((args.rccl_algo, "NCCL_ALGO"), (args.rccl_proto, "NCCL_PROTO"),
                      (str(args.rccl_channels) if args.rccl_channels > 0 else None, "NCCL_MIN_NCHANNELS")):
        if flag:
            os.environ[var] = flag
    if args.cpu_probe > 0:
        return cpu_probe(args.cpu_probe, args.height, args.width, args.num_layers)
    from depthcore import _lib as _dc_lib
    if _dc_lib.IS_VARIANT:          # tuning / ablation builds are for the sweep scripts under tools/ only
        raise SystemExit("bench.py measures the product library; unset DEPTHCORE_LIB (%s)" % _dc_lib.LIB_PATH)
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch(args, sys.argv[1:])
    return run_rank(args)


if __name__ == "__main__":
    sys.exit(main())
