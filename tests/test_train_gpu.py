"""The reference-shaped Trainer (process_batch -> backward -> Adam) on the GPU vs the CPU oracle's
full training step with the same weights, inputs and tie-break noise; fused vs layer-by-layer path."""
import numpy as np
import pytest
import torch

from oracle import ref_cpu as R
from oracle.train_step import CpuTrainer
from helpers import close, rel_l2

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
    # weights after one Adam step (Adam normalises the step to ~lr, so compare the update direction)
    upd_h, upd_c = [], []
    for k, m in tr.models.items():
        for n, p in m.named_parameters():
            if ".fc." in n:
                continue
            upd_h.append((p.detach().cpu() - state[k][n]).flatten())
            upd_c.append((ct.state[k][n].detach() - state[k][n]).flatten())
    uh, uc = torch.cat(upd_h), torch.cat(upd_c)
    agree = float((torch.sign(uh) == torch.sign(uc)).float().mean())
    assert agree > 0.97, agree


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
