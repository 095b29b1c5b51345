"""f2: the ConvGRU v5 temporal fusion on the HIP path (fused conv blocks + dc_gru_* gate kernels) against the reference's
fixtures (tests/golden/convgru.npz) and the CPU oracle."""
import numpy as np
import pytest
import torch

from oracle import gru_ref as G
import make_golden as MG
import make_golden_r2 as MG2
from helpers import T, close, rel_l2
from test_gru_oracle import cell_state, v5_state

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_cell_vs_reference_fixture(golden):
    import networks
    g = golden["convgru"]
    c = MG2.GRU_CELL
    m = networks.ConvGRUModel_v1({"input_dim": c["C"], "hidden_dim_1": c["C"], "height": c["H"], "width": c["W"]}, (3, 3), True, "cpu")
    m.load_state_dict(cell_state())
    m = m.to(DEV)
    x, h, _ = MG2.gru_cell_inputs()
    x, h = x.to(DEV).requires_grad_(), h.to(DEV).requires_grad_()
    y = m(x, h)
    close(y, g["cell_y"], rtol=1e-4, atol=1e-5)
    ps = {k: p for k, p in m.named_parameters() if k != "h0_layer1"}
    gr = torch.autograd.grad((y * T(g["cell_cot"]).to(DEV)).sum(), [x, h] + list(ps.values()))
    assert rel_l2(gr[0], g["cell_gx"]) < 2e-4 and rel_l2(gr[1], g["cell_gh"]) < 2e-4
    for j, k in enumerate(ps):
        got = gr[2 + j].cpu() if gr[2 + j].numel() <= 4096 else torch.from_numpy(MG.summ(gr[2 + j]))
        close(got, g["cell_g_" + k], rtol=2e-3, atol=2e-4, msg=k)


def test_gate_kernels_vs_torch():
    from depthcore import ops
    g = torch.Generator().manual_seed(3)
    B, C, H, W = 2, 5, 7, 9
    gates = torch.rand(B, 2 * C, H, W, generator=g)
    h, cnm = torch.randn(B, C, H, W, generator=g), torch.randn(B, C, H, W, generator=g)
    cot1, cot2 = torch.randn(B, C, H, W, generator=g), torch.randn(B, C, H, W, generator=g)
    lo = [t.clone().requires_grad_() for t in (gates, h, cnm)]
    ro = lo[0][:, :C] * lo[1]
    bo = (1 - lo[0][:, C:]) * lo[1] + lo[0][:, C:] * lo[2]
    go = torch.autograd.grad((ro * cot1).sum() + (bo * cot2).sum(), lo)
    lh = [t.to(DEV).requires_grad_() for t in (gates, h, cnm)]
    rh = ops.gru_reset_times_state(lh[0], lh[1])
    bh = ops.gru_blend(lh[0], lh[1], lh[2])
    gh = torch.autograd.grad((rh * cot1.to(DEV)).sum() + (bh * cot2.to(DEV)).sum(), lh)
    close(rh, ro, rtol=1e-6, atol=1e-7); close(bh, bo, rtol=1e-6, atol=1e-7)
    for a, b in zip(gh, go):
        close(a, b, rtol=1e-5, atol=1e-6)
    f, Hs = torch.randn(3, 4, 5, 6, generator=g), torch.randn(4, 4, 5, 6, generator=g)
    cot = torch.randn(3, 4, 5, 6, generator=g)
    fo, Ho = f.clone().requires_grad_(), Hs.clone().requires_grad_()
    oo = fo + (Ho[1:] + Ho[:-1]) / 2
    go = torch.autograd.grad((oo * cot).sum(), [fo, Ho])
    fh, Hh = f.to(DEV).requires_grad_(), Hs.to(DEV).requires_grad_()
    oh = ops.gru_sequence_residual(fh, Hh)
    gh = torch.autograd.grad((oh * cot.to(DEV)).sum(), [fh, Hh])
    close(oh, oo, rtol=1e-6, atol=1e-7)
    for a, b in zip(gh, go):
        close(a, b, rtol=1e-6, atol=1e-7)


def test_blocks_v5_sequence_vs_reference_fixture(golden):
    """ConvGRUBlocks_v5 over a 2-frame sequence at the reference's hard-coded 192x640 feature sizes, with the hidden-state
    aggregation of trainer_gru.py:607-639: outputs and every gradient (features, weights, learned initial states)."""
    import networks
    g = golden["convgru"]
    blk = networks.ConvGRUBlocks_v5(kernel_size=(3, 3), bias=True, device="cpu")
    assert list(blk.state_dict().keys()) == list(g["v5_keys"])
    blk.load_state_dict(v5_state())
    blk = blk.to(DEV)
    feats, gen = MG2.gru_v5_features()
    cots = None
    feats = [f.to(DEV).requires_grad_() for f in feats]
    fused = blk.run_sequence(feats)
    cots = [torch.randn(f.shape, generator=gen) / f[0].numel() ** 0.5 for f in fused]
    tot = sum((f * c.to(DEV)).sum() for f, c in zip(fused, cots))
    ps = dict(blk.named_parameters())
    gr = torch.autograd.grad(tot, feats + list(ps.values()))
    for k in range(5):
        close(MG.summ(fused[k]), g["v5_out%d" % k], rtol=1e-3, atol=1e-3, msg="out%d" % k)
        close(MG.summ(gr[k]), g["v5_gfeat%d" % k], rtol=2e-3, atol=2e-5, msg="gfeat%d" % k)
    for j, k in enumerate(ps):
        close(MG.summ(gr[5 + j]), g["v5_g_" + k], rtol=5e-3, atol=5e-5, msg=k)


def test_stack_frames_is_torch_cat():
    """dc_gather_copy (ops.stack_frames): the sequence trainer's per-step concatenations in one launch -- bitwise torch.cat,
    incl. 4x4 intrinsics (64-byte segments), odd element counts and an unaligned view."""
    from depthcore import ops
    g = torch.Generator().manual_seed(9)
    groups = [[torch.rand(1, 3, 24, 40, generator=g).to(DEV) for _ in range(3)], [torch.rand(1, 4, 4, generator=g).to(DEV) for _ in range(3)],
              [torch.rand(2, 5, 7, generator=g).to(DEV), torch.rand(1, 5, 7, generator=g).to(DEV)],
              [torch.rand(1, 3, 9, generator=g).to(DEV)[:, :, 1:], torch.rand(1, 3, 8, generator=g).to(DEV)]]
    outs = ops.stack_frames(groups)
    for o, grp in zip(outs, groups):
        assert torch.equal(o, torch.cat(grp, 0))
    many = [[torch.rand(1, 17, generator=g).to(DEV) for _ in range(3)] for _ in range(40)]       # 120 segments: two launches
    for o, grp in zip(ops.stack_frames(many), many):
        assert torch.equal(o, torch.cat(grp, 0))


def test_level_nodes_vs_per_op_graph():
    """ops._GruLevel (one autograd node per level, frames walked inside it) against the per-op graph of the same module on a
    4-frame sequence: same kernels for the convolutions and the gate products, so outputs and every gradient agree to rounding
    of the differently ordered sums; with a frozen initial state (trainer_gru.py h_s_epoch) no gradient is reported for it."""
    import networks
    from networks import convgru as CG
    torch.manual_seed(5)
    blk = networks.ConvGRUBlocks_v5(kernel_size=(3, 3), bias=True, device="cpu", height=64, width=96).to(DEV)
    for p_ in blk.parameters():
        p_.data.normal_(0, 0.05)
    n = 4
    feats = [torch.randn(n, c, 64 >> (k + 1), 96 >> (k + 1), device=DEV).requires_grad_() for k, c in enumerate((64, 64, 128, 256, 512))]
    cots = [torch.randn_like(f) for f in feats]
    ps = list(blk.parameters())
    res = {}
    for mode in (True, False):
        CG.LEVEL_NODES = mode
        try:
            outs = blk.run_sequence(feats)
            gr = torch.autograd.grad(sum((o * c).sum() for o, c in zip(outs, cots)), feats + ps)
        finally:
            CG.LEVEL_NODES = True
        res[mode] = ([o.detach() for o in outs], gr)
    for a, b in zip(res[True][0], res[False][0]):
        assert torch.equal(a, b)
    for j, (a, b) in enumerate(zip(res[True][1], res[False][1])):
        assert rel_l2(a, b) < 1e-5, (j, rel_l2(a, b))
    blk.cgru_2.h0_layer1.requires_grad_(False)
    outs = blk.run_sequence(feats)
    sum((o * c).sum() for o, c in zip(outs, cots)).backward()
    assert blk.cgru_2.h0_layer1.grad is None and blk.cgru_1.h0_layer1.grad is not None


def test_gru_training_steps():
    """BASELINE configs[3] wiring: one sequence of 3 frames (batch size 1), encoder -> ConvGRU v5 -> decoder, pose on the
    stacked pairs, loss on the stacked sequence; finite losses over Adam steps, the learned h0 states receive gradients."""
    import trainer as T_
    from depthcore.synthetic import synthetic_sequence_batch
    H, W, n = 64, 96, 3
    opt = T_.default_options(batch_size=1, height=H, width=W, gru="v5", len_sequence=n)
    tr = T_.Trainer(opt, device=DEV, seed=0)
    tr.set_train()
    inputs = synthetic_sequence_batch(n, H, W, torch.device(DEV))
    ls = []
    for _ in range(6):
        outputs, l = tr.train_step(inputs)
        ls.append(float(l["loss"]))
    assert outputs[("disp", 0)].shape == (n, 1, H, W) and all(np.isfinite(ls))
    for k in range(5):
        h0 = getattr(tr.models["gru"], "cgru_%d" % k).h0_layer1
        assert h0.grad is not None and float(h0.grad.abs().sum()) > 0
        assert float(h0.detach().abs().sum()) > 0          # moved away from its zero initialisation
