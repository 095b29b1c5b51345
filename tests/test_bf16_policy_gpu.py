"""BASELINE configs[4]'s policy -- "networks in reduced precision, loss in fp32" -- at the level of a training step.

The reference has no autocast, so there is no reference number for it; parity is defined HERE, not in prose:

  * yardstick = torch's own statement of the policy: the CPU oracle with its networks under `torch.autocast("cpu",
    torch.bfloat16)` and the photometric loss in fp32 (oracle/train_step.py `nets_autocast`), measured against the fp64
    oracle.  The HIP trainer with `nets_dtype="bf16"` (convolution operands rounded to bf16 for the matrix cores, fp32
    accumulate, everything else fp32) must be NO FURTHER from fp64 than 1.5x that yardstick (+ a small floor), for the
    loss, every disparity map, the poses, and every network's parameter gradient (relative L2);
  * kernel-level exactness of the rounding itself is tests/test_conv_bf16_gpu.py (against an oracle with operands rounded
    the same way, rel. 1e-6);
  * at the full configs[4] size: finite, bitwise reproducible, every convolution that can take the bf16 kernels does, and
    the loss within 2e-2 of the fp32 path's."""
import numpy as np
import pytest
import torch

from oracle import ref_cpu as R
from oracle.train_step import CpuTrainer
from helpers import rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _grads_by_model(named):
    out = {}
    for (k, n), g in named.items():
        out.setdefault(k, []).append(g.detach().double().cpu().flatten())
    return {k: torch.cat(v) for k, v in out.items()}


@pytest.mark.parametrize("fusion", [None, "v3"])
def test_bf16_policy_step_vs_autocast_yardstick(fusion):
    import trainer as T
    B, H, W = 2, 64, 128
    kw = dict(fusion="v3", frame_ids=[0, -2, -1, 1]) if fusion else {}
    opt = T.default_options(batch_size=B, height=H, width=W, cpu_tiebreak_noise=True, nets_dtype="bf16", **kw)
    tr = T.Trainer(opt, device=DEV, seed=3)
    tr.set_train()
    if fusion:      # heads in the operating range of a trained model (see tests/test_train_gpu.py)
        with torch.no_grad():
            for k in range(1, 5):
                head = getattr(tr.models["fusion"], "fusion_block_%d" % k).conv3x3.conv
                head.weight.mul_(0.02)
                head.bias.fill_(0.5)
    state = {k: {n: t.detach().cpu().clone() for n, t in m.state_dict().items()} for k, m in tr.models.items()}
    inputs = R.synthetic_inputs(B, H, W, seed=0, frame_ids=(0, -2, -1, 1) if fusion else (0, -1, 1))
    noise = R.tiebreak_noise(B, H, W)

    def oracle(dtype, autocast=None):
        st = {k: {n: (t.to(dtype) if t.is_floating_point() else t) for n, t in sd.items()} for k, sd in state.items()}
        ct = CpuTrainer(st, R.Opt(height=H, width=W))
        inp = {k: v.to(dtype) for k, v in inputs.items()}
        oo, ol = ct.process_batch(inp, [n.to(dtype) for n in noise], nets_autocast=autocast)
        ol["loss"].backward()
        g = _grads_by_model({(k, n): t.grad for k, sd in ct.state.items() for n, t in sd.items() if t.requires_grad and t.grad is not None})
        return float(ol["loss"].detach()), {k: v.detach() for k, v in oo.items() if torch.is_tensor(v)}, g
    l64, o64, g64 = oracle(torch.float64)
    lac, oac, gac = oracle(torch.float32, torch.bfloat16)

    torch.manual_seed(1234)
    tr.buckets.zero()
    ho, hl = tr.process_batch({k: v.to(DEV) for k, v in inputs.items()})
    hl["loss"].backward()
    gh = _grads_by_model({(k, n): p.grad for k, m in tr.models.items() for n, p in m.named_parameters() if p.grad is not None})

    report = {}

    def check(name, e_hip, e_ac, floor):
        report[name] = (e_hip, e_ac)
        assert e_hip <= 1.5 * e_ac + floor, (name, report)
    check("loss", abs(float(hl["loss"].detach()) - l64) / abs(l64), abs(lac - l64) / abs(l64), 2e-4)
    for s in range(4):
        check("disp%d" % s, rel_l2(ho[("disp", s)], o64[("disp", s)]), rel_l2(oac[("disp", s)], o64[("disp", s)]), 1e-4)
    for f in (-1, 1):
        check("T%d" % f, rel_l2(ho[("cam_T_cam", 0, f)], o64[("cam_T_cam", 0, f)]),
              rel_l2(oac[("cam_T_cam", 0, f)], o64[("cam_T_cam", 0, f)]), 1e-5)
    assert set(gh) == set(g64)
    for k in g64:
        check("grad " + k, rel_l2(gh[k], g64[k]), rel_l2(gac[k], g64[k]), 1e-3)
    print("bf16 policy (quantity: |hip - f64| / |f64|, |torch autocast - f64| / |f64|):", report)


def test_bf16_policy_full_size_properties(monkeypatch):
    """BASELINE configs[4] per rank: Fusion_v3 on frames {-2,-1,0}, B = 12, 192 x 640.  Two trainers from the same seed:
    bitwise identical loss and gradients; the fp32 trainer's loss within 2e-2; the bf16 kernels really run (the launch
    counters of the instrumented families: no Winograd launch is left for a 3x3 convolution of the step)."""
    import trainer as T
    from depthcore import ops
    from depthcore.synthetic import synthetic_batch
    batch = synthetic_batch(12, 192, 640, torch.device(DEV), seed=3, frame_ids=(0, -2, -1, 1))
    res = {}
    for name, dt in (("a", "bf16"), ("b", "bf16"), ("f32", "f32")):
        opt = T.default_options(batch_size=12, height=192, width=640, fusion="v3", frame_ids=[0, -2, -1, 1], nets_dtype=dt)
        tr = T.Trainer(opt, device=DEV, seed=6)
        tr.set_train()
        if name == "a":
            ops.conv_profile_enable(600, 1)
        _, losses = tr.train_step(dict(batch))
        torch.cuda.synchronize()
        if name == "a":
            counts = {k: ops.conv_profile_collect(k)["launches"] for k in range(4)}
            ops.conv_profile_enable(0, 1)
        g = torch.cat([p.grad.flatten() for p in tr.parameters_to_train if p.grad is not None])
        res[name] = (float(losses["loss"].detach()), g.clone())
        tr.close()
        del tr
        torch.cuda.empty_cache()
    assert np.isfinite(res["a"][0]) and bool(torch.isfinite(res["a"][1]).all())
    assert res["a"][0] == res["b"][0] and torch.equal(res["a"][1], res["b"][1])
    assert abs(res["a"][0] - res["f32"][0]) <= 2e-2 * abs(res["f32"][0]), (res["a"][0], res["f32"][0])
    assert counts[0] == 0 and counts[1] == 0 and counts[2] > 80 and counts[3] > 20, counts
