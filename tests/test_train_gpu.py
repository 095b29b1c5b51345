"""The reference-shaped Trainer (process_batch -> backward -> Adam) on the GPU vs the CPU oracle's
full training step with the same weights, inputs and tie-break noise; fused vs layer-by-layer path."""
import numpy as np
import pytest
import torch

from oracle import ref_cpu as R
from oracle.train_step import CpuTrainer
from kink_tape import KinkTape
from helpers import close, close_frac, pass_rate_1e3, rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _setup(B, H, W, **kw):
    import trainer as T
    opt = T.default_options(batch_size=B, height=H, width=W, cpu_tiebreak_noise=True, **kw)
    tr = T.Trainer(opt, device=DEV, seed=3)
    tr.set_train()
    state = {k: {n: t.detach().cpu().clone() for n, t in m.state_dict().items()} for k, m in tr.models.items()}
    inputs = R.synthetic_inputs(B, H, W, seed=0)
    return tr, state, inputs


def test_process_batch_and_step_vs_oracle():
    B, H, W = 2, 64, 96
    tr, state, inputs = _setup(B, H, W)
    ct = CpuTrainer(state, R.Opt(height=H, width=W))
    oo, ol = ct.train_step(inputs, R.tiebreak_noise(B, H, W))
    dev_in = {k: v.to(DEV) for k, v in inputs.items()}
    torch.manual_seed(1234)          # the CPU randn stream of trainer.py:594-595
    go, gl = tr.train_step(dev_in)
    torch.cuda.synchronize()
    close(gl["loss"], ol["loss"], rtol=1e-3, atol=0)
    for s in range(4):
        close(gl["loss/%d" % s], ol["loss/%d" % s], rtol=1e-3, atol=0)
        close(go[("disp", s)], oo[("disp", s)], rtol=1e-3, atol=1e-5)
    for f in (-1, 1):
        close(go[("cam_T_cam", 0, f)], oo[("cam_T_cam", 0, f)], rtol=1e-3, atol=1e-6)


def _grads_by_model(named_grads):
    out = {}
    for (k, n), g in named_grads.items():
        out.setdefault(k, []).append(g.detach().double().cpu().flatten())
    return {k: torch.cat(v) for k, v in out.items()}


@pytest.mark.parametrize("fusion", [None, "v3", "gru"])
def test_all_parameter_gradients_vs_oracle(fusion):
    """Every parameter gradient of one full step (both encoders, decoders, [Fusion_v3]) against the CPU oracle's autograd,
    per network by relative L2 norm.  The photometric gradient is ill-conditioned in fp32 (see test_photo_gpu), so the
    bound is calibrated: the oracle is also run in fp64, and the HIP path may be at most 5x as far from fp64 as the
    oracle's own fp32 run (floor 2e-4; the ratio between two fp32 evaluations of such a sum is itself noisy) -- for EVERY
    network."""
    import trainer as T
    B, H, W = 2, 64, 96
    gru = fusion == "gru"                    # one sequence of B frames at batch size 1 (trainer_gru.py run_gru_v5)
    kw = dict(gru="v5", len_sequence=B) if gru else (dict(fusion="v3", frame_ids=[0, -2, -1, 1]) if fusion else {})
    opt = T.default_options(batch_size=1 if gru else B, height=H, width=W, cpu_tiebreak_noise=True, **kw)
    tr = T.Trainer(opt, device=DEV, seed=3)
    tr.set_train()
    if fusion == "v3":
        # Fusion_v3 has no output non-linearity: at random init its "disparities" have any sign and the geometry behind them
        # is chaotic.  Put the heads in the operating range of a trained model (disp ~ 0.5) so that gradients are comparable.
        with torch.no_grad():
            for k in range(1, 5):
                head = getattr(tr.models["fusion"], "fusion_block_%d" % k).conv3x3.conv
                head.weight.mul_(0.02)
                head.bias.fill_(0.5)
    if gru:                                  # h0 starts at zero: move it so that its path is exercised
        with torch.no_grad():
            for k in range(5):
                getattr(tr.models["gru"], "cgru_%d" % k).h0_layer1.normal_(0, 0.1, generator=torch.Generator(device=DEV).manual_seed(k))
    state = {k: {n: t.detach().cpu().clone() for n, t in m.state_dict().items()} for k, m in tr.models.items()}
    inputs = R.synthetic_inputs(B, H, W, seed=0, frame_ids=(0, -2, -1, 1) if fusion == "v3" else (0, -1, 1))
    noise = R.tiebreak_noise(B, H, W)

    # The HIP step runs first and records every ReLU / max-pool decision of the two encoders and the pose decoder
    # (tests/kink_tape.py: KinkTape); the oracle is then evaluated with those decisions imposed (oracle/kinks.py), so both sides
    # differentiate the same smooth function and a pre-activation within rounding of zero cannot re-route a gradient
    # (tests/test_encoder_gpu.py).  No network is exempted from the bound below.
    from depthcore import ops
    tapes = {}

    def taped(name, method="forward"):
        mod, orig = tr.models[name], getattr(tr.models[name], method)

        def fwd(*a, **k):
            with KinkTape() as t:
                out = orig(*a, **k)
            tapes[name] = t.entries
            return out
        setattr(mod, method, fwd)
    for name, method in (("encoder", "forward"), ("pose_encoder", "forward_pairs"), ("pose", "forward_poses")):
        taped(name, method)
    torch.manual_seed(1234)
    tr.buckets.zero()
    dev_in = {k: v.to(DEV) for k, v in inputs.items()}
    if gru:                                  # the sequence schema: one dict entry per frame of the sequence
        dev_in = {k + (j,): v[j:j + 1].contiguous() for k, v in dev_in.items() if k[0] != "color_aug" for j in range(B)}
    _, gl = tr.process_batch(dev_in)
    gl["loss"].backward()
    gh = _grads_by_model({(k, n): p.grad for k, m in tr.models.items() for n, p in m.named_parameters() if p.grad is not None})
    assert set(tapes) == {"encoder", "pose_encoder", "pose"}

    def oracle(dtype):
        st = {k: {n: (t.to(dtype) if t.is_floating_point() else t) for n, t in sd.items()} for k, sd in state.items()}
        ct = CpuTrainer(st, R.Opt(height=H, width=W))
        inp = {k: v.to(dtype) for k, v in inputs.items()}
        _, ol = ct.process_batch(inp, [n.to(dtype) for n in noise], kinks=tapes)
        ol["loss"].backward()
        return float(ol["loss"].detach()), _grads_by_model({(k, n): t.grad for k, sd in ct.state.items() for n, t in sd.items()
                                                            if t.requires_grad and t.grad is not None}), ct
    l64, g64, ct64 = oracle(torch.float64)
    l32, g32, ct32 = oracle(torch.float32)
    rep64 = ct64.kink_report
    # the imposed decisions differ from the fp64 oracle's own only on near-ties: |pre-activation| within 4x the fp32 oracle's
    # own rounding error on that tensor (tests/test_encoder_gpu.py)
    from oracle.kinks import uncalibrated_disagreements
    far = [(na,) + d for (na, a), (nb, b) in zip(ct64.kink_objs, ct32.kink_objs) for d in uncalibrated_disagreements(a, b)]
    assert not far, far
    assert abs(float(gl["loss"].detach()) - l64) <= (1e-3 if fusion == "v3" else 1e-4) * abs(l64)
    assert set(gh) == set(g64)
    report = {}
    for k in g64:
        assert gh[k].shape == g64[k].shape, k          # the same parameters received a gradient
        e_hip, e_32 = rel_l2(gh[k], g64[k]), rel_l2(g32[k], g64[k])
        report[k] = (e_hip, e_32)
    # how much of the checker the device decides: the fraction of the imposed ReLU / max-pool decisions that differ from the
    # fp64 oracle's own (each of them shown above to be a near-tie); everything else the oracle decides exactly as it would alone
    imposed = sum(t.numel() for entries in tapes.values() for _, t in entries)
    flipped = sum(d[3] for d in rep64)
    print("per-network gradient error (hip vs f64, f32 oracle vs f64):", report,
          "; decisions differing from fp64: %d of %d imposed (%.2e)" % (flipped, imposed, flipped / max(imposed, 1)),
          "; 1e-3 pointwise pass-rate per network:", {k: round(pass_rate_1e3(gh[k], g64[k]), 4) for k in g64})
    assert flipped <= 2e-4 * imposed, (flipped, imposed)
    # Tightened in round 6 from 5x to 2x the fp32 oracle's own distance from fp64: measured (profiles/round6_parity_passrates.txt, the
    # three front-ends) the HIP path sits at 0.09-0.95 of it -- encoder 1.3e-3 vs 3.0e-3, depth 5.7e-4 vs 1.1e-3, pose encoder 5.0e-4 vs
    # 5.2e-4 (the worst ratio), fusion 2.1e-4 vs 4.6e-4, gru 6.4e-5 vs 3.0e-4 -- i.e. never further from fp64 than the oracle itself
    for k, (e_hip, e_32) in report.items():
        assert e_hip <= 2.0 * e_32 + 2e-4, (k, report)


@pytest.mark.parametrize("B,H,W", [(2, 64, 96), (12, 192, 640)])
def test_fused_equals_layerwise_path(B, H, W):
    """The fused photometric op against the layer-by-layer sequence of independent kernels, on the same networks; the
    second case is the full BASELINE configs[1] size (a property check in place of a CPU run of that size)."""
    tr, state, inputs = _setup(B, H, W)
    dev_in = {k: v.to(DEV) for k, v in inputs.items()}
    res = []
    for fused in (True, False):
        tr.opt.fused_loss = fused
        torch.manual_seed(1234)
        tr.buckets.zero()
        outputs, losses = tr.process_batch(dict(dev_in))
        losses["loss"].backward()
        g = torch.cat([p.grad.flatten() for n, p in tr.models["depth"].named_parameters()]).clone()
        gp = torch.cat([p.grad.flatten() for n, p in tr.models["pose"].named_parameters()]).clone()
        res.append((float(losses["loss"].detach()), g, gp))
    assert abs(res[0][0] - res[1][0]) / abs(res[1][0]) < 1e-4
    assert rel_l2(res[0][1], res[1][1]) < 2e-2
    assert rel_l2(res[0][2], res[1][2]) < 2e-2


def test_materialized_outputs_schema():
    B, H, W = 2, 64, 96
    tr, state, inputs = _setup(B, H, W, materialize_logs=True)
    dev_in = {k: v.to(DEV) for k, v in inputs.items()}
    torch.manual_seed(1234)
    outputs, losses = tr.process_batch(dev_in)
    for s in range(4):
        assert outputs[("depth", 0, s)].shape == (B, 1, H, W)
        assert outputs["identity_selection/%d" % s].shape == (B, H, W)
        for f in (-1, 1):
            assert outputs[("sample", f, s)].shape == (B, H, W, 2)
            assert outputs[("color", f, s)].shape == (B, 3, H, W)
            assert outputs[("color_identity", f, s)] is dev_in[("color", f, 0)]
    assert set(losses) == {"loss/0", "loss/1", "loss/2", "loss/3", "loss"}


def test_loss_decreases_over_steps():
    import trainer as T
    from depthcore.synthetic import synthetic_batch
    B, H, W = 4, 64, 96
    tr = T.Trainer(T.default_options(batch_size=B, height=H, width=W), device=DEV, seed=0)
    tr.set_train()
    inputs = synthetic_batch(B, H, W, torch.device(DEV))
    ls = []
    for _ in range(30):
        _, l = tr.train_step(inputs)
        ls.append(float(l["loss"]))
    assert all(np.isfinite(ls)) and min(ls[-5:]) < ls[0]


def test_loss_curve_matches_cpu_oracle():
    """Loss curve vs the CPU oracle: Adam steps on the same batch with the same weights and the same tie-break
    noise.  The first steps must agree to ~1e-5; afterwards the two fp32 trajectories separate because the
    early Adam update is lr*g/|g| = +-lr per parameter (v ~ g^2), so a gradient whose sign is decided by
    rounding noise moves that parameter in opposite directions on the two machines.  Measured separation
    stays below 1e-2 over the horizon tested; 1e-3 (north_star) holds for the first ~6 steps."""
    B, H, W = 2, 64, 96
    tr, state, inputs = _setup(B, H, W)
    ct = CpuTrainer(state, R.Opt(height=H, width=W))
    dev_in = {k: v.to(DEV) for k, v in inputs.items()}
    gpu, cpu = [], []
    for step in range(12):
        g = torch.Generator().manual_seed(5000 + step)
        noise = [torch.randn(B, 2, H, W, generator=g) for _ in range(4)]
        _, ol = ct.train_step(inputs, noise)
        cpu.append(float(ol["loss"].detach()))
        tr._noise = lambda b, n, _nz=noise: [t.to(DEV) for t in _nz]
        _, gl = tr.train_step(dev_in)
        gpu.append(float(gl["loss"].detach()))
    gpu, cpu = np.array(gpu), np.array(cpu)
    rel = np.abs(gpu - cpu) / np.abs(cpu)
    assert rel[:5].max() < 1e-4, rel[:5]
    assert rel[:7].max() < 1e-3, rel[:7]
    assert rel.max() < 2e-2, (rel.max(), gpu[-3:], cpu[-3:])


def test_teacher_forced_loss_curve_100_steps():
    """100 Adam steps of the CPU oracle on changing batches; before every step the GPU trainer is given the oracle's
    current weights and BatchNorm buffers and evaluates the same batch with the same tie-break noise.  Unlike the free-
    running comparison above this does not compound the +-lr sign flips of early Adam, so the whole curve has to agree:
    every loss within 1e-3 (north_star), the typical one within a few 1e-6."""
    B, H, W = 2, 64, 96
    tr, state, _ = _setup(B, H, W)
    ct = CpuTrainer(state, R.Opt(height=H, width=W))
    rel = []
    for step in range(100):
        inputs = R.synthetic_inputs(B, H, W, seed=100 + step)
        g = torch.Generator().manual_seed(9000 + step)
        noise = [torch.randn(B, 2, H, W, generator=g) for _ in range(4)]
        for k, m in tr.models.items():
            m.load_state_dict({n: t.detach() for n, t in ct.state[k].items()})
        tr._noise = lambda b, n, _nz=noise: [t.to(DEV) for t in _nz]
        with torch.no_grad():
            go, gl = tr.process_batch({k: v.to(DEV) for k, v in inputs.items()})
        oo, ol = ct.train_step(inputs, noise)
        rel.append(abs(float(gl["loss"]) - float(ol["loss"].detach())) / abs(float(ol["loss"].detach())))
        if step % 25 == 0:
            for s_ in range(4):
                close(go[("disp", s_)], oo[("disp", s_)], rtol=1e-3, atol=1e-5)
    rel = np.array(rel)
    assert rel.max() < 1e-3, (rel.max(), int(rel.argmax()))
    assert np.median(rel) < 2e-5, np.median(rel)


def test_free_running_loss_curve_100_steps_calibrated():
    """north_star / SURVEY 8d: "loss curve over >= 100 steps within 1e-3 of the CPU run".  Free-running (nobody is handed the
    other's weights), 100 Adam steps on changing batches with the same tie-break noise, three trajectories from the same
    initial state: the HIP trainer (fp32), the CPU oracle in fp32 and the CPU oracle in fp64.

    A free-running comparison of two fp32 machines cannot hold 1e-3 for 100 steps on ANY implementation -- early Adam moves
    every parameter by +-lr whatever the gradient's size, so a gradient whose sign is decided by rounding sends that parameter
    opposite ways -- and this test MEASURES that instead of asserting it: S(t) = |L_cpu32(t) - L_cpu64(t)| / |L_cpu64(t)| is the
    separation of the oracle from itself at another precision.  The separation grows in jumps (one flipped parameter at a
    time), at steps that differ between machines, so the gate compares running maxima with slack in time: by step t the HIP
    trajectory is no further from the fp64 one than K = 4 times what the oracle's own fp32 trajectory reaches by step 2t + 5
    (+ 1e-5), it agrees to 1e-4 while the oracle agrees with itself to 2.5e-5, and all three curves end in the same place
    (mean of the last 10 losses within 2 %).  Measured (MI355X, this seed): HIP 2e-6 / 1e-4 / 4e-4 / 9e-3 / 1.4e-2 at steps
    1 / 5 / 10 / 25 / 100, the fp32 oracle 1e-7 / 8e-5 / 2e-4 / 7e-3 / 1.5e-2.
    The printed S(t) is the evidence for the sign-flip explanation: it grows to the 1e-3..1e-2 level by itself."""
    B, H, W, N = 2, 64, 96, 100
    tr, state, _ = _setup(B, H, W)
    c32 = CpuTrainer(state, R.Opt(height=H, width=W))
    c64 = CpuTrainer({k: {n: (t.double() if t.is_floating_point() else t) for n, t in sd.items()} for k, sd in state.items()},
                     R.Opt(height=H, width=W))
    L = np.zeros((3, N))
    for step in range(N):
        inputs = R.synthetic_inputs(B, H, W, seed=300 + step)
        g = torch.Generator().manual_seed(7000 + step)
        noise = [torch.randn(B, 2, H, W, generator=g) for _ in range(4)]
        tr._noise = lambda b, n, _nz=noise: [t.to(DEV) for t in _nz]
        _, gl = tr.train_step({k: v.to(DEV) for k, v in inputs.items()})
        L[0, step] = float(gl["loss"].detach())
        L[1, step] = float(c32.train_step(inputs, noise)[1]["loss"].detach())
        L[2, step] = float(c64.train_step({k: v.double() for k, v in inputs.items()}, [t.double() for t in noise])[1]["loss"].detach())
    D = np.abs(L[0] - L[2]) / np.abs(L[2])          # HIP vs fp64 oracle
    S = np.abs(L[1] - L[2]) / np.abs(L[2])          # fp32 oracle vs fp64 oracle: the yardstick
    Dm, Sm = np.maximum.accumulate(D), np.maximum.accumulate(S)
    print("free-running separation from the fp64 oracle, running max at steps 1/5/10/25/50/100: HIP %s | fp32 oracle %s"
          % (["%.1e" % Dm[i] for i in (0, 4, 9, 24, 49, 99)], ["%.1e" % Sm[i] for i in (0, 4, 9, 24, 49, 99)]))
    ahead = Sm[np.minimum(N - 1, 2 * np.arange(N) + 5)]
    assert np.all(Dm <= 4.0 * ahead + 1e-5), (int(np.argmax(Dm - 4.0 * ahead)), Dm.max(), Sm.max())
    calm = Sm <= 2.5e-5                              # the stretch in which the oracle still agrees with itself
    assert calm[0] and np.all(D[calm] <= 1e-4), (D[calm].max() if calm.any() else None)
    # Where the curves end (mean of the last 10 losses).  The fp32 oracle must end within 2 % of the fp64 one; the HIP run within
    # the calibrated bound -- 4x the oracle's own largest separation (floor 2 %).  A fixed 2 % for the HIP run was NOT a property
    # of the kernels: of six equally valid roundings of the same arithmetic (DC_WINO_FORCE = a tile variant / reduction split
    # forced on every Winograd launch: 1,2,1 / 2,2,1 / 1,2,4 / 1,2,2 / 2,2,2, and the cost model's own picks) three end this
    # 100-step run within 2 % of the fp64 oracle and three end 2.8 %, 2.8 % and 3.6 % below it (gpurun_out/r5c, round 4).
    tail = L[:, -10:].mean(axis=1)
    print("mean of the last 10 losses: HIP %.5f | fp32 oracle %.5f | fp64 oracle %.5f" % tuple(tail))
    assert abs(tail[1] - tail[2]) <= 0.02 * abs(tail[2]), tail
    assert abs(tail[0] - tail[2]) <= max(0.02, 4.0 * Sm[-1]) * abs(tail[2]), (tail, Sm[-1])
    assert L[0, -10:].mean() < L[0, :10].mean()      # and it trains


def test_checkpoint_roundtrip(tmp_path):
    import trainer as T
    from depthcore.synthetic import synthetic_batch
    B, H, W = 2, 64, 96
    a = T.Trainer(T.default_options(batch_size=B, height=H, width=W), device=DEV, seed=1)
    a.set_train()
    inputs = synthetic_batch(B, H, W, torch.device(DEV))
    a.train_step(inputs)
    a.save_model(str(tmp_path))
    sd = torch.load(str(tmp_path / "encoder.pth"))
    assert sd["height"] == H and sd["width"] == W and "encoder.conv1.weight" in sd       # trainer.py:721-725
    b = T.Trainer(T.default_options(batch_size=B, height=H, width=W), device=DEV, seed=2)
    b.load_model(str(tmp_path))
    b.set_train()
    _, la = a.train_step(inputs)
    _, lb = b.train_step(inputs)
    assert abs(float(la["loss"].detach()) - float(lb["loss"].detach())) < 1e-6


@pytest.mark.parametrize("tag", ["v1ms", "pmask"])
def test_trainer_ablations_golden(golden, tag):
    """a15 ablations on the layer-by-layer HIP path against the reference's outputs (tests/golden/trainer_ablations.npz):
    `v1_multiscale` (warps and losses at each scale's own resolution) and `predictive_mask` (+ disable_automasking)."""
    import trainer as T
    import make_golden as MG
    import make_golden_r2 as MG2
    from layers import transformation_from_parameters
    g = golden["trainer_ablations"]
    B, H, W = MG.B, MG.H, MG.W
    kw = dict(v1_multiscale=True) if tag == "v1ms" else dict(disable_automasking=True, predictive_mask=True)
    tr = T.Trainer(T.default_options(batch_size=B, height=H, width=W, cpu_tiebreak_noise=True, **kw), device=DEV, seed=0)
    inputs = {k: v.to(DEV) for k, v in R.synthetic_inputs(B, H, W, seed=0).items()}
    disp, aa, tr_, mask = MG2.ablation_inputs(tag)
    leaves = [disp[s].to(DEV).requires_grad_() for s in range(4)]
    pose = [aa[-1].to(DEV).requires_grad_(), aa[1].to(DEV).requires_grad_(), tr_[-1].to(DEV).requires_grad_(),
            tr_[1].to(DEV).requires_grad_()]
    outputs = {("disp", s): leaves[s] for s in range(4)}
    masks = []
    if mask is not None:
        masks = [mask[s].to(DEV).requires_grad_() for s in range(4)]
        outputs["predictive_mask"] = {("disp", s): masks[s] for s in range(4)}
    for j, f in enumerate((-1, 1)):
        outputs[("cam_T_cam", 0, f)] = transformation_from_parameters(pose[j], pose[2 + j], invert=(f < 0))
        close(outputs[("cam_T_cam", 0, f)], g[tag + "_T_%d" % f], rtol=1e-4, atol=1e-6)
    tr.generate_images_pred(inputs, outputs)
    torch.manual_seed(1234)
    losses = tr.compute_losses(inputs, outputs)
    close(losses["loss"], g[tag + "_loss"], rtol=1e-3, atol=0)
    grads = torch.autograd.grad(losses["loss"], leaves + pose + masks)

    # Gradient bound calibrated like the default path's (test_photo_gpu): the oracle, pinned to the reference's outputs for
    # these ablations by tests/test_oracle_golden.py, is evaluated on the same leaves in fp64 and in fp32; the HIP gradients
    # may be at most 3x as far from fp64 as the oracle's own fp32 evaluation is (floor 2e-4).
    def oracle(dtype):
        inp = {k: v.to(dtype) for k, v in R.synthetic_inputs(B, H, W, seed=0).items()}
        d_, aa_, tr__, m_ = MG2.ablation_inputs(tag)
        lv = [d_[s].to(dtype).requires_grad_() for s in range(4)]
        ps = [aa_[-1].to(dtype).requires_grad_(), aa_[1].to(dtype).requires_grad_(), tr__[-1].to(dtype).requires_grad_(),
              tr__[1].to(dtype).requires_grad_()]
        out = {("disp", s): lv[s] for s in range(4)}
        ms = []
        if m_ is not None:
            ms = [m_[s].to(dtype).requires_grad_() for s in range(4)]
            out["predictive_mask"] = {("disp", s): ms[s] for s in range(4)}
        for j, f in enumerate((-1, 1)):
            out[("cam_T_cam", 0, f)] = R.transformation_from_parameters(ps[j], ps[2 + j], invert=(f < 0))
        opt = R.Opt(height=H, width=W, **kw)
        R.generate_images_pred(inp, out, opt)
        torch.manual_seed(1234)
        nz = None if tag == "pmask" else [torch.randn(B, 2, H >> s, W >> s).to(dtype) for s in range(4)]
        return torch.autograd.grad(R.compute_losses(inp, out, opt, nz)["loss"], lv + ps + ms)
    g64, g32 = oracle(torch.float64), oracle(torch.float32)
    report = []
    for i in range(len(grads)):
        e_hip, e_32 = rel_l2(grads[i], g64[i]), rel_l2(g32[i], g64[i])
        report.append((i, e_hip, e_32))
        assert e_hip <= 3.0 * e_32 + 2e-4, report
    print("ablation %s gradients (leaf, |hip-f64|/|f64|, |f32-f64|/|f64|):" % tag, report)
    # ... and against the reference's own fp32 numbers.  Those carry the same conditioning error (the reference's fp32
    # disparity gradient at scale 1 of `pmask` is itself 4 % away from fp64), so each bound is twice the fixture's own
    # distance from the fp64 oracle (+ 1e-3): a check that fixture, oracle and kernels solve the same problem.
    def vs_fixture(i, name):
        want = g[tag + name]
        allowed = 2.0 * rel_l2(want, g64[i]) + 1e-3
        assert rel_l2(grads[i], want) <= allowed, (name, rel_l2(grads[i], want), allowed)
    for s in range(4):
        close(losses["loss/%d" % s], g[tag + "_loss%d" % s], rtol=1e-3, atol=0)
        vs_fixture(s, "_gdisp%d" % s)
        if masks:
            vs_fixture(8 + s, "_gmask%d" % s)
    for j, f in enumerate((-1, 1)):
        vs_fixture(4 + j, "_gaa_%d" % f)
        vs_fixture(6 + j, "_gtr_%d" % f)


def test_v1_multiscale_on_the_fused_kernels_golden(golden):
    """`v1_multiscale` on the fused kernels (Trainer.fused_losses_v1: one single-scale fused call per scale at that scale's own
    resolution) against the reference's outputs (tests/golden/trainer_ablations.npz, tag v1ms) and against the layer-by-layer
    HIP path on the same leaves: losses 1e-3 / 1e-4, disparity gradients 2e-3, pose gradients within the conditioning budget."""
    import trainer as T
    import make_golden as MG
    import make_golden_r2 as MG2
    from layers import transformation_from_parameters
    g = golden["trainer_ablations"]
    B, H, W = MG.B, MG.H, MG.W
    tr = T.Trainer(T.default_options(batch_size=B, height=H, width=W, cpu_tiebreak_noise=True, v1_multiscale=True), device=DEV, seed=0)
    inputs = {k: v.to(DEV) for k, v in R.synthetic_inputs(B, H, W, seed=0).items()}
    disp, aa, tr_, _ = MG2.ablation_inputs("v1ms")

    def run(fused):
        leaves = [disp[s].to(DEV).requires_grad_() for s in range(4)]
        pose = [aa[-1].to(DEV).requires_grad_(), aa[1].to(DEV).requires_grad_(), tr_[-1].to(DEV).requires_grad_(),
                tr_[1].to(DEV).requires_grad_()]
        outputs = {("disp", s): leaves[s] for s in range(4)}
        for j, f in enumerate((-1, 1)):
            outputs[("cam_T_cam", 0, f)] = transformation_from_parameters(pose[j], pose[2 + j], invert=(f < 0))
        torch.manual_seed(1234)
        if fused:
            losses = tr.fused_losses_v1(inputs, outputs)
        else:
            tr.generate_images_pred(inputs, outputs)
            losses = tr.compute_losses(inputs, outputs)
        return losses, torch.autograd.grad(losses["loss"], leaves + pose), outputs
    lf, gf, of = run(True)
    ll, gl, ol = run(False)
    close(lf["loss"], g["v1ms_loss"], rtol=1e-3, atol=0)
    close(lf["loss"], ll["loss"], rtol=1e-4, atol=0)
    for s in range(4):
        close(lf["loss/%d" % s], g["v1ms_loss%d" % s], rtol=1e-3, atol=0)
        close(lf["loss/%d" % s], ll["loss/%d" % s], rtol=1e-4, atol=0)
        assert rel_l2(gf[s], gl[s]) <= 2e-3, (s, rel_l2(gf[s], gl[s]))
        assert of[("argmin", s)].shape == (B, H >> s, W >> s)
        sel = (of[("argmin", s)] > 1).float()
        assert float((sel != ol["identity_selection/%d" % s]).float().mean()) <= 1e-3
    for i in range(4, 8):
        assert rel_l2(gf[i], gl[i]) <= 3e-2, (i, rel_l2(gf[i], gl[i]))
    # and a whole training step takes this path by default now
    _, losses = tr.train_step({k: v.clone() for k, v in inputs.items()})
    assert torch.isfinite(losses["loss"]) and ("argmin", 3) in _


@pytest.mark.parametrize("variant", ["pmask", "pmask+avg", "pmask+v1"])
def test_predictive_mask_on_the_fused_kernels(golden, variant):
    """`predictive_mask` (+ disable_automasking, trainer.py:571-590) on the fused kernels: the masks multiply the reprojection
    losses inside photo_fwdg_kernel (they ride in the load slots of the unused tie-break noise), scale the SSIM / L1 derivative,
    and get their own gradient (d to_optimise / d mask_f = the unmasked loss of the frame the min() took); the BCE term stays a
    torch expression.  Against the reference's outputs for the plain variant (tests/golden/trainer_ablations.npz, tag pmask)
    and against the layer-by-layer HIP path on the same leaves for all three (with avg_reprojection; at each scale's own size)."""
    import trainer as T
    import make_golden as MG
    import make_golden_r2 as MG2
    from layers import transformation_from_parameters
    g = golden["trainer_ablations"]
    B, H, W = MG.B, MG.H, MG.W
    kw = dict(disable_automasking=True, predictive_mask=True)
    if variant == "pmask+avg":
        kw["avg_reprojection"] = True
    if variant == "pmask+v1":
        kw["v1_multiscale"] = True
    tr = T.Trainer(T.default_options(batch_size=B, height=H, width=W, cpu_tiebreak_noise=True, **kw), device=DEV, seed=0)
    inputs = {k: v.to(DEV) for k, v in R.synthetic_inputs(B, H, W, seed=0).items()}
    disp, aa, tr_, mask = MG2.ablation_inputs("pmask")

    def run(fused):
        leaves = [disp[s].to(DEV).requires_grad_() for s in range(4)]
        pose = [aa[-1].to(DEV).requires_grad_(), aa[1].to(DEV).requires_grad_(), tr_[-1].to(DEV).requires_grad_(),
                tr_[1].to(DEV).requires_grad_()]
        masks = [mask[s].to(DEV).requires_grad_() for s in range(4)]
        outputs = {("disp", s): leaves[s] for s in range(4)}
        outputs["predictive_mask"] = {("disp", s): masks[s] for s in range(4)}
        for j, f in enumerate((-1, 1)):
            outputs[("cam_T_cam", 0, f)] = transformation_from_parameters(pose[j], pose[2 + j], invert=(f < 0))
        torch.manual_seed(1234)
        if fused:
            losses = tr.fused_losses_v1(inputs, outputs) if "v1" in variant else tr.fused_losses(inputs, outputs)
        else:
            tr.generate_images_pred(inputs, outputs)
            losses = tr.compute_losses(inputs, outputs)
        return losses, torch.autograd.grad(losses["loss"], leaves + pose + masks)
    lf, gf = run(True)
    ll, gl = run(False)
    if variant == "pmask":
        close(lf["loss"], g["pmask_loss"], rtol=1e-3, atol=0)
        for s in range(4):
            close(lf["loss/%d" % s], g["pmask_loss%d" % s], rtol=1e-3, atol=0)
    close(lf["loss"], ll["loss"], rtol=1e-4, atol=0)
    for s in range(4):
        close(lf["loss/%d" % s], ll["loss/%d" % s], rtol=1e-4, atol=0)

    # Gradients against the layer-by-layer path on the same leaves.  Maps (disparities, masks): pointwise with a small
    # outlier budget -- min() and the SSIM clamp make the loss piecewise smooth, and ONE routing decision taken the other way on
    # a rounding-level tie moves a whole window of gradients: the reference's own fp32 disparity gradient at scale 1 of this
    # fixture is 4 % (L2) away from fp64 for exactly that reason (test_trainer_ablations_golden), and so is the fused path's,
    # while 99.5 % of the elements agree to 2e-3.  Pose gradients (sums over all pixels): normwise, the conditioning budget.
    for s in range(4):
        scale_g = float(gl[s].abs().mean())
        bad = max(5e-3, 4.0 / gl[s].numel())        # (a scale-3 map has 192 elements: one flipped window touches 2-4 of them)
        close_frac(gf[s], gl[s], rtol=2e-3, atol=2e-3 * scale_g, bad=bad, msg="d disp %d" % s)
        scale_m = float(gl[8 + s].abs().mean())
        close_frac(gf[8 + s], gl[8 + s], rtol=2e-3, atol=2e-3 * scale_m, bad=max(5e-3, 8.0 / gl[8 + s].numel()), msg="d mask %d" % s)
    for i in range(4, 8):
        assert rel_l2(gf[i], gl[i]) <= 3e-2, ("pose", i, rel_l2(gf[i], gl[i]))
    print("%s: |fused - layer| / |layer| of the disparity gradients: %s" % (variant, ["%.1e" % rel_l2(gf[s], gl[s]) for s in range(4)]))
    # a whole training step (mask decoder included) runs on the fused path, and evaluation (no gradient) gives the same loss
    batch = {k: v.clone() for k, v in inputs.items()}
    torch.manual_seed(7)
    with torch.no_grad():
        _, le = tr.process_batch(dict(batch))
    torch.manual_seed(7)
    _, lt = tr.train_step(dict(batch))
    assert torch.isfinite(lt["loss"])
    close(le["loss"], lt["loss"], rtol=2e-5, atol=0)


def test_fusion_v3_training_steps():
    """BASELINE configs[4] wiring at a small size: frames [-2,-1,0] through encoder + decoder + Fusion_v3, loss finite and
    decreasing over Adam steps, every fusion parameter that the reference trains receives a gradient."""
    import trainer as T
    from depthcore.synthetic import synthetic_batch
    B, H, W = 2, 64, 96
    opt = T.default_options(batch_size=B, height=H, width=W, fusion="v3", frame_ids=[0, -2, -1, 1])
    tr = T.Trainer(opt, device=DEV, seed=0)
    tr.set_train()
    inputs = synthetic_batch(B, H, W, torch.device(DEV), frame_ids=(0, -2, -1, 1))
    ls = []
    for _ in range(12):
        outputs, l = tr.train_step(inputs)
        ls.append(float(l["loss"]))
    assert outputs[("disp", 0)].shape == (B, 1, H, W)
    assert all(np.isfinite(ls)) and min(ls[-4:]) < ls[0], ls
    for n, p in tr.models["fusion"].named_parameters():
        assert (p.grad is None) == n.startswith("fusion_block_4.upscale"), n


@pytest.mark.parametrize("gru,dtype", [(None, "f32"), ("v5", "f32"), (None, "bf16")])
def test_wino_weight_cache_changes_nothing(gru, dtype):
    """dc_wino_cache_*: one batched weight transform per step instead of one launch per convolution.  Same device function,
    so the first step's loss agrees to the last bits with and without the cache and the following steps to the rounding of
    the few order-dependent reductions; the set of cached variants is complete after the first step.  `bf16`: the prepared
    weights of the bf16 direct kernels (c3b_wprep_item) ride in the same table."""
    import trainer as T
    from depthcore.synthetic import synthetic_batch, synthetic_sequence_batch

    def run(cache):
        kw = dict(gru="v5", len_sequence=3, batch_size=1) if gru else dict(batch_size=2)
        opt = T.default_options(height=64, width=96, wino_weight_cache=cache, nets_dtype=dtype, **kw)
        tr = T.Trainer(opt, device=DEV, seed=5)
        tr.set_train()
        inputs = synthetic_sequence_batch(3, 64, 96, torch.device(DEV), seed=2) if gru else synthetic_batch(2, 64, 96, torch.device(DEV), seed=2)
        losses, variants = [], []
        for _ in range(3):
            _, l = tr.train_step(dict(inputs))
            losses.append(float(l["loss"].detach()))
            variants.append(tr.wino_cache.variants())
        w = tr.models["encoder"].encoder.layer1[0].conv1.weight.detach().clone()
        tr.wino_cache.close()
        return losses, variants, w

    la, va, wa = run(True)
    lb, vb, wb = run(False)
    assert va[0] > 20 and va[0] == va[1] == va[2], va          # every variant is met in the first step
    assert vb == [0, 0, 0]
    assert abs(la[0] - lb[0]) <= 1e-6 * abs(lb[0]), (la, lb)      # same forward arithmetic, same fixed-order reductions
    assert np.allclose(la, lb, rtol=2e-4, atol=0), (la, lb)      # (Adam turns last-bit gradient differences into +-lr updates)
    assert float((wa - wb).abs().max()) <= 7e-4                   # two trajectories, 3 steps, each update within +-lr = 1e-4


@pytest.mark.parametrize("gru", [None, "v5"])
def test_hip_graph_replays_train_like_eager_steps(gru):
    """opt.hip_graph: after three eager steps the whole step (two streams, backward, capturable fused Adam, the device-side
    noise-seed increment) is captured in one hipGraph and replayed.  Same kernels, same seeds: the loss sequence follows the
    eager trainer's; replays really train (losses move) and the static output tensors are rewritten in place.
    (64 x 128: every convolution is a depthcore launch -- the library convolution that tiny or odd maps fall back to is not
    capturable.)"""
    import trainer as T
    from depthcore.synthetic import synthetic_batch, synthetic_sequence_batch

    def run(graph, n=7):
        kw = dict(gru="v5", len_sequence=3, batch_size=1) if gru else dict(batch_size=2)
        tr = T.Trainer(T.default_options(height=64, width=128, hip_graph=graph, **kw), device=DEV, seed=5)
        tr.set_train()
        batches = [synthetic_sequence_batch(3, 64, 128, torch.device(DEV), seed=s) if gru else synthetic_batch(2, 64, 128, torch.device(DEV), seed=s)
                   for s in (2, 3)]
        losses = []
        for i in range(n):
            _, l = tr.train_step(dict(batches[i % 2]))
            losses.append(float(l["loss"].detach()))
        assert tr.step == n
        captured = tr._graph is not None
        tr.wino_cache.close()
        return losses, captured

    le, ce = run(False)
    lg, cg = run(True)
    assert cg and not ce
    assert np.allclose(le[:3], lg[:3], rtol=1e-5), (le, lg)          # eager warm-up steps of both trainers
    assert np.allclose(le, lg, rtol=5e-4), (le, lg)                  # captured step + replays (capturable Adam: bias correction in fp32 on the device)
    assert len(set(round(v, 7) for v in lg[3:])) == len(lg[3:])      # every replay is a new step


@pytest.mark.parametrize("mode", ["eager", "gru"])
def test_wgrad_lanes_change_nothing(mode):
    """opt.wgrad_lanes: the convolutions' weight-gradient kernels run on companion streams of the backward's two streams and
    are joined before Adam.  Same kernels on the same operands, only their streams differ: losses AND every parameter after
    five steps are bitwise those of the single-lane trainer -- also with the ConvGRU front-end, whose cell weights are used
    three times per graph and therefore must stay on the backward's own stream (autograd sums their gradients there)."""
    import trainer as T
    from depthcore import ops
    from depthcore.synthetic import synthetic_batch, synthetic_sequence_batch
    gru = mode == "gru"

    def run(lanes, n=5):
        kw = dict(gru="v5", len_sequence=3, batch_size=1) if gru else dict(batch_size=2)
        tr = T.Trainer(T.default_options(height=64, width=128, wgrad_lanes=lanes, **kw), device=DEV, seed=5)
        assert tr.wgrad_lanes == bool(lanes)
        tr.set_train()
        batches = [synthetic_sequence_batch(3, 64, 128, torch.device(DEV), seed=s) if gru else synthetic_batch(2, 64, 128, torch.device(DEV), seed=s)
                   for s in (2, 3)]
        losses = [float(tr.train_step(dict(batches[i % 2]))[1]["loss"].detach()) for i in range(n)]
        torch.cuda.synchronize()
        params = {"%s.%s" % (m, k): v.detach().clone() for m, net in tr.models.items() for k, v in net.named_parameters()}
        used = len(ops.WgradLanes._lanes)
        tr.close()
        return losses, params, used

    l0, p0, _ = run(0)
    l1, p1, used = run(1)
    assert used >= 2                                   # a lane for each of the two branch streams exists
    assert l0 == l1, (l0, l1)
    bad = [k for k in p0 if not torch.equal(p0[k], p1[k])]
    assert not bad, bad[:5]


def test_step_stream_priority_changes_nothing():
    """Trainer.on_step_stream(): the training loop on the trainer's high-priority stream (opt.step_priority; the pose branch's
    side stream and the weight-gradient lanes stay at normal priority).  Same kernels on the same operands in the same order per
    stream, only the priority of one stream differs: losses AND every parameter after five steps are bitwise those of the loop
    on the caller's stream; the context is a no-op with step_priority = 0, hands the caller's stream back, and "auto" (2) picks
    the stream only for the configurations it was measured on (Trainer.__init__)."""
    import trainer as T
    from depthcore.synthetic import synthetic_batch

    def run(prio, n=5):
        tr = T.Trainer(T.default_options(height=64, width=128, batch_size=2, wgrad_lanes=1, step_priority=prio), device=DEV, seed=5)
        tr.set_train()
        cur = torch.cuda.current_stream(DEV)
        with tr.on_step_stream():
            inside = torch.cuda.current_stream(DEV)
            batches = [synthetic_batch(2, 64, 128, torch.device(DEV), seed=s) for s in (2, 3)]
            losses = [tr.train_step(dict(batches[i % 2]))[1]["loss"].detach() for i in range(n)]
        assert torch.cuda.current_stream(DEV) == cur
        losses = [float(l) for l in losses]              # (read on the caller's stream, which waits for the loop's)
        torch.cuda.synchronize()
        params = {"%s.%s" % (m, k): v.detach().clone() for m, net in tr.models.items() for k, v in net.named_parameters()}
        tr.close()
        return losses, params, inside != cur, inside.priority

    l0, p0, moved0, _ = run(0)
    l1, p1, moved1, prio1 = run(-1)
    assert not moved0 and moved1 and prio1 == -1
    assert l0 == l1, (l0, l1)
    bad = [k for k in p0 if not torch.equal(p0[k], p1[k])]
    assert not bad, bad[:5]
    auto = {kw: T.Trainer(T.default_options(step_priority=2, **dict(kw)), device="cpu").step_priority
            for kw in ((("batch_size", 12),), (("batch_size", 1),), (("batch_size", 8), ("num_layers", 50), ("height", 320), ("width", 1024)))}
    assert list(auto.values()) == [-1, 0, 0], auto


def test_graph_capture_after_an_unclosed_collected_trainer():
    """The round-3 abort (gpurun_out/r3s): Trainers of FAILED tests were never close()d and sat in reference cycles (exception
    <-> frame); torch.cuda.graph() runs gc.collect() + empty_cache() when it begins a capture, so their weights were freed --
    and returned to the driver -- inside the capture, where the weight cache may not rebuild its descriptor table; the
    captured refresh then named freed memory and the replay page-faulted, which the HSA runtime turns into abort().
    Now every cache owner has its own table (a refresh only ever reads its owner's live weights) and the Trainer collects
    garbage BEFORE it starts a capture.  Here: an eager trainer B takes a step, is tied into a cycle and dropped without
    close(); graph-mode trainer A must capture, replay and follow an eager trainer that never saw a B."""
    import gc
    import trainer as T
    from depthcore.synthetic import synthetic_batch
    dev = torch.device(DEV)
    batches = [synthetic_batch(2, 64, 128, dev, seed=s) for s in (2, 3)]

    def opts(graph):
        return T.default_options(height=64, width=128, batch_size=2, hip_graph=graph)

    ref = T.Trainer(opts(False), device=DEV, seed=5)
    ref.set_train()
    want = [float(ref.train_step(dict(batches[i % 2]))[1]["loss"].detach()) for i in range(7)]
    ref.close()

    gc.collect()
    gc.disable()                                  # the collection must happen where torch.cuda.graph / the Trainer do it
    try:
        b = T.Trainer(opts(False), device=DEV, seed=9)
        b.set_train()
        b.train_step(dict(batches[0]))
        owner_b = b.wino_cache._owner
        b.cycle = b                               # unreachable but not freed by reference counting, like a failed test's frame
        del b
        a = T.Trainer(opts(True), device=DEV, seed=5)
        a.set_train()
        got = [float(a.train_step(dict(batches[i % 2]))[1]["loss"].detach()) for i in range(7)]
        torch.cuda.synchronize()
    finally:
        gc.enable()
    assert a._graph is not None and getattr(a, "_graph_failed", 0) == 0
    assert a.wino_cache._owner != owner_b
    assert np.allclose(want, got, rtol=5e-4), (want, got)
    a.close()


def test_failed_capture_falls_back_to_eager_instead_of_killing_the_process():
    """A launch that is REFUSED inside the capture (an entry point returning DC_E*, here injected) must surface as a warning +
    an eager step -- host-side step counter rolled back, the losses of an eager trainer from then on -- never as an abort."""
    import trainer as T
    from depthcore._lib import DepthcoreError
    from depthcore.synthetic import synthetic_batch
    dev = torch.device(DEV)
    batches = [synthetic_batch(2, 64, 128, dev, seed=s) for s in (2, 3)]
    ref = T.Trainer(T.default_options(height=64, width=128, batch_size=2), device=DEV, seed=5)
    ref.set_train()
    want = [float(ref.train_step(dict(batches[i % 2]))[1]["loss"].detach()) for i in range(6)]
    ref.close()
    a = T.Trainer(T.default_options(height=64, width=128, batch_size=2, hip_graph=True), device=DEV, seed=5)
    a.set_train()
    orig = a._train_step_eager

    def sabotaged(inputs):
        if torch.cuda.is_current_stream_capturing():
            a.wino_cache.refresh()                # part of the step is already recorded when the launch is refused
            raise DepthcoreError("dc_conv3x3_fwd failed: DC_ELAUNCH (injected)")
        return orig(inputs)
    a._train_step_eager = sabotaged
    with pytest.warns(UserWarning, match="capture of the training step failed"):
        got = [float(a.train_step(dict(batches[i % 2]))[1]["loss"].detach()) for i in range(6)]
    assert a.step == 6 and a._graph is None and a._graph_failed == 1
    assert np.allclose(want, got, rtol=5e-4), (want, got)
    a.close()


def test_illegal_call_inside_a_capture_raises_and_does_not_abort():
    """An ILLEGAL call inside the capture (a device synchronisation by foreign code) invalidates it.  The Trainer ends the
    capture, clears the error state and tries the step eagerly; if the runtime's capture state cannot be recovered it raises a
    RuntimeError that says so.  Either way: a Python exception or a correct step -- never an aborted interpreter.  (In a child
    process, tests/capture_illegal_child.py: a poisoned runtime must not take the other tests with it.)"""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, os.path.join(here, "capture_illegal_child.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.returncode, r.stderr[-3000:])
    assert "outcome=raised" in r.stdout or "outcome=recovered" in r.stdout, r.stdout[-2000:]
    print(r.stdout.strip().splitlines()[-1])


def test_second_trainer_does_not_break_a_captured_graph():
    """The Winograd weight cache is process-wide and a captured hipGraph bakes its buffers into kernel arguments: building
    (and closing) another Trainer must neither free nor rewrite anything the first trainer's graph reads or writes.  Trainer
    A captures; trainer B is built, trains eagerly with its own weights and is closed; A's replays must still follow an
    eager trainer that never saw a B."""
    import trainer as T
    from depthcore.synthetic import synthetic_batch
    dev = torch.device(DEV)
    batches = [synthetic_batch(2, 64, 128, dev, seed=s) for s in (2, 3)]

    def opts(graph):
        return T.default_options(height=64, width=128, batch_size=2, hip_graph=graph)

    ref = T.Trainer(opts(False), device=DEV, seed=5)
    ref.set_train()
    want = [float(ref.train_step(dict(batches[i % 2]))[1]["loss"].detach()) for i in range(9)]
    ref.close()

    a = T.Trainer(opts(True), device=DEV, seed=5)
    a.set_train()
    got = [float(a.train_step(dict(batches[i % 2]))[1]["loss"].detach()) for i in range(5)]
    assert a._graph is not None
    b = T.Trainer(opts(False), device=DEV, seed=77)          # registers ITS weights in the shared registry
    b.set_train()
    for i in range(2):
        b.train_step(dict(batches[i % 2]))                   # new variants -> the descriptor table is rebuilt
        got.append(float(a.train_step(dict(batches[(5 + i) % 2]))[1]["loss"].detach()))
    b.close()
    del b
    torch.cuda.empty_cache()
    junk = [torch.full((1 << 20,), float("nan"), device=dev) for _ in range(64)]      # whatever was freed gets reused
    got += [float(a.train_step(dict(batches[(7 + i) % 2]))[1]["loss"].detach()) for i in range(2)]
    del junk
    a.close()
    assert np.allclose(want, got, rtol=5e-4), (want, got)


def test_hip_graph_with_two_batch_shapes():
    """A new input shape inside graph mode: its first steps run eagerly (new kernel variants, workspace sizes and weight-cache
    variants may allocate and synchronise -- illegal inside a capture), then it gets its own graph; going back to the first
    shape replays that shape's graph.  Losses follow an eager trainer fed the same sequence of batches."""
    import trainer as T
    from depthcore.synthetic import synthetic_batch
    dev = torch.device(DEV)
    small = [synthetic_batch(2, 64, 128, dev, seed=s) for s in (2, 3)]
    wide = [synthetic_batch(1, 64, 256, dev, seed=s) for s in (4, 5)]
    seq = [small[0], small[1], small[0], small[1], small[0], wide[0], wide[1], wide[0], wide[1], wide[0], small[1], wide[1],
           small[0]]

    def run(graph):
        tr = T.Trainer(T.default_options(height=64, width=128, batch_size=2, hip_graph=graph), device=DEV, seed=9)
        tr.set_train()
        # the geometry modules are built per (batch, height, width): give the second shape its own
        out = []
        for b in seq:
            B, _, H, W = b[("color", 0, 0)].shape
            tr.opt.batch_size, tr.opt.height, tr.opt.width = B, H, W
            tr.loss_batch = B
            out.append(float(tr.train_step(dict(b))[1]["loss"].detach()))
        n = len(tr._graphs)
        tr.close()
        return out, n

    le, ne = run(False)
    lg, ng = run(True)
    assert ne == 0 and ng == 2
    assert np.allclose(le, lg, rtol=1e-3), (le, lg)


def test_gru_hidden_state_freeze_schedule():
    """trainer_gru.py:295-307: when epoch + 1 == h_s_epoch the learned initial states stop receiving gradients (and stop
    moving: Adam skips parameters without a gradient); the rest of the model keeps training."""
    import trainer as T
    from depthcore.synthetic import synthetic_sequence_batch
    tr = T.Trainer(T.default_options(height=64, width=128, batch_size=1, gru="v5", len_sequence=3, h_s_epoch=2), device=DEV, seed=4)
    tr.set_train()
    batch = synthetic_sequence_batch(3, 64, 128, torch.device(DEV), seed=1)
    h0 = [c.h0_layer1 for c in tr.models["gru"].cells()]
    tr.start_epoch(0)
    tr.train_step(dict(batch))
    assert all(p.requires_grad and p.grad is not None and float(p.grad.abs().sum()) > 0 for p in h0)
    before = [p.detach().clone() for p in h0]
    w_before = tr.models["gru"].cgru_0.cgru_1.conv_gates.weight.detach().clone()
    tr.start_epoch(1)                                     # (1 + 1) == h_s_epoch
    assert not any(p.requires_grad for p in h0)
    tr.train_step(dict(batch))
    assert all(p.grad is None for p in h0)
    assert all(torch.equal(a, b) for a, b in zip(before, h0))
    assert not torch.equal(w_before, tr.models["gru"].cgru_0.cgru_1.conv_gates.weight)
    tr.close()


def test_c3_full_train_step_properties(monkeypatch):
    """BASELINE configs[2] per rank (resnet50, 320x1024, B=8): one whole training step -- finite, no framework convolution
    entered, and bitwise reproducible from the same state (all reductions are fixed-order)."""
    import torch.nn.functional as F
    import trainer as T
    from depthcore.synthetic import synthetic_batch
    calls = []
    orig = F.conv2d
    monkeypatch.setattr(torch.nn.functional, "conv2d", lambda *a, **k: (calls.append(1), orig(*a, **k))[1])
    real = torch.nn.Conv2d.forward
    monkeypatch.setattr(torch.nn.Conv2d, "forward", lambda self, x: (calls.append(self), real(self, x))[1])
    batch = synthetic_batch(8, 320, 1024, torch.device(DEV), seed=3)
    res = []
    for _ in range(2):
        tr = T.Trainer(T.default_options(height=320, width=1024, batch_size=8, num_layers=50), device=DEV, seed=6)
        tr.set_train()
        _, losses = tr.train_step(dict(batch))
        g = torch.cat([p.grad.flatten() for p in tr.parameters_to_train if p.grad is not None])
        res.append((float(losses["loss"].detach()), g.clone(), tr.models["encoder"].encoder.layer3[5].conv3.weight.detach().clone()))
        tr.close()
        del tr
        torch.cuda.empty_cache()
    assert calls == [], "library convolution entered for %r" % calls[:3]
    assert np.isfinite(res[0][0]) and bool(torch.isfinite(res[0][1]).all())
    assert res[0][0] == res[1][0] and torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][2], res[1][2])


@pytest.mark.parametrize("cfg", ["c2", "c4", "c5"])
def test_full_size_train_step_properties(cfg, monkeypatch):
    """The other BASELINE configurations at their full sizes, like test_c3_full_train_step_properties: configs[1] (resnet18,
    192x640, B=12), configs[3] (ConvGRU v5, one sequence of 3 frames at 192x640 -- trainer_gru.py:595-644, batch size 1) and
    configs[4] in fp32 (Fusion_v3 on frames {-2,-1,0}, 192x640, B=12 -- trainer_fusion_v3.py:311-330): one whole training step
    is finite, enters NO framework convolution, and is bitwise reproducible from the same state (fixed-order reductions)."""
    import torch.nn.functional as F
    import trainer as T
    from depthcore.synthetic import synthetic_batch, synthetic_sequence_batch
    calls = []
    orig = F.conv2d
    monkeypatch.setattr(torch.nn.functional, "conv2d", lambda *a, **k: (calls.append(1), orig(*a, **k))[1])
    real = torch.nn.Conv2d.forward
    monkeypatch.setattr(torch.nn.Conv2d, "forward", lambda self, x: (calls.append(self), real(self, x))[1])
    dev = torch.device(DEV)
    if cfg == "c4":
        kw = dict(gru="v5", len_sequence=3, batch_size=1)
        batch = synthetic_sequence_batch(3, 192, 640, dev, seed=3)
    elif cfg == "c5":
        kw = dict(fusion="v3", frame_ids=[0, -2, -1, 1], batch_size=12)
        batch = synthetic_batch(12, 192, 640, dev, seed=3, frame_ids=(0, -2, -1, 1))
    else:
        kw = dict(batch_size=12)
        batch = synthetic_batch(12, 192, 640, dev, seed=3)
    res = []
    for _ in range(2):
        tr = T.Trainer(T.default_options(height=192, width=640, **kw), device=DEV, seed=6)
        tr.set_train()
        out, losses = tr.train_step(dict(batch))
        g = torch.cat([p.grad.flatten() for p in tr.parameters_to_train if p.grad is not None])
        res.append((float(losses["loss"].detach()), g.clone(), out[("disp", 0)].detach().clone()))
        tr.close()
        del tr
        torch.cuda.empty_cache()
    assert calls == [], "library convolution entered for %r" % calls[:3]
    assert np.isfinite(res[0][0]) and bool(torch.isfinite(res[0][1]).all()) and float(res[0][1].abs().max()) > 0
    assert res[0][2].shape[-2:] == (192, 640)
    assert res[0][0] == res[1][0] and torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][2], res[1][2])


def test_gradient_sums_in_kernel_epilogues_change_nothing():
    """The sums autograd would form for tensors with several consumers -- an encoder feature map (trunk + decoder skip), a stage
    input (3x3 / 2 conv1 + 1x1 / 2 downsample), the decoder's x (dispconv + next upconv) -- ride in the store epilogues of the
    data-gradient kernels (ops.SkipSum, pair ops.GradFork): every parameter gradient of a training step equals the
    autograd-summed one to rounding (three-operand sums associate differently), and no gradient is left uncollected."""
    from networks import resnet_encoder as RE, depth_decoder as DD
    from depthcore import ops
    for layers in (18, 50):
        res = {}
        for on in (True, False):
            RE.SKIP_SUMS, DD.X_FORK, gf = on, on, RE.GRAD_FORK
            RE.GRAD_FORK = on
            try:
                tr, state, inputs = _setup(2, 64, 128, num_layers=layers)
                dev_in = {k: v.to(DEV) for k, v in inputs.items()}
                torch.manual_seed(1234)
                tr.buckets.zero()
                _, losses = tr.process_batch(dict(dev_in))
                losses["loss"].backward()
                ops.assert_no_dangling_sums()
                res[on] = {(k, n): p.grad.detach().clone() for k, m in tr.models.items() for n, p in m.named_parameters() if p.grad is not None}
                res[on]["loss"] = losses["loss"].detach().clone()
            finally:
                RE.SKIP_SUMS, DD.X_FORK, RE.GRAD_FORK = True, True, gf
        assert torch.equal(res[True]["loss"], res[False]["loss"])
        assert set(res[True]) == set(res[False])
        worst = max((rel_l2(res[True][k], res[False][k]), k) for k in res[True])
        assert worst[0] < 1e-5, worst
