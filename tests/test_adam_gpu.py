"""depthcore.optim.Adam (dc_adam_step) against torch.optim.Adam -- the reference's optimiser (trainer.py:110-113) -- on the
same parameters and gradients: the updates, the skipped-tensor semantics, and the state_dict interchange both ways."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
SHAPES = [(1,), (3,), (5,), (1, 1, 1, 3, 1), (64,), (4096,), (4097,), (16, 16, 3, 3), (10001,), (256, 128, 3, 3), (1000, 512)]


def _params(seed):
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(s, generator=g) * 0.1 for s in SHAPES]


def _grads(seed, step, skip=()):
    g = torch.Generator().manual_seed(1000 * seed + step)
    return [None if i in skip else torch.randn(s, generator=g) * (0.01 if i % 2 else 1.0) for i, s in enumerate(SHAPES)]


def _rel(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-300))


def test_matches_torch_adam_over_steps_with_skipped_tensors():
    from depthcore.optim import Adam
    init = _params(1)
    ours = [torch.nn.Parameter(p.clone().to(DEV)) for p in init]
    ref = [torch.nn.Parameter(p.clone().double()) for p in init]                 # fp64 reference of the same recurrence
    ref32 = [torch.nn.Parameter(p.clone()) for p in init]                        # torch's own fp32 (the yardstick for the bound)
    o1, o2, o3 = Adam(ours, 1e-2, (0.9, 0.999), 1e-8), torch.optim.Adam(ref, 1e-2), torch.optim.Adam(ref32, 1e-2, foreach=False)
    for step in range(6):
        skip = (2, 7) if step in (1, 2) else ()        # tensors 2 and 7 get no gradient on two steps: skipped, their step count stays
        gs = _grads(1, step, skip)
        for plist, cast in ((ours, lambda t: t.to(DEV)), (ref, lambda t: t.double()), (ref32, lambda t: t.clone())):
            for p, g in zip(plist, gs):
                p.grad = None if g is None else cast(g)
        o1.step(); o2.step(); o3.step()
    torch.cuda.synchronize()
    for i, (a, b, c) in enumerate(zip(ours, ref, ref32)):
        ea, ec = _rel(a.detach().cpu(), b.detach()), _rel(c.detach(), b.detach())
        assert ea <= max(4 * ec, 3e-7), (i, SHAPES[i], ea, ec)
        st = o1.state[a]
        assert float(st["step"]) == (4.0 if i in (2, 7) else 6.0)
        assert _rel(st["exp_avg"].cpu(), o2.state[b]["exp_avg"]) <= 1e-6
        assert _rel(st["exp_avg_sq"].cpu(), o2.state[b]["exp_avg_sq"]) <= 1e-6


def test_state_dict_interchanges_with_torch_adam_both_ways():
    from depthcore.optim import Adam
    init = _params(2)
    a = [torch.nn.Parameter(p.clone().to(DEV)) for p in init]
    b = [torch.nn.Parameter(p.clone().to(DEV)) for p in init]
    oa, ob = Adam(a, 3e-3), torch.optim.Adam(b, 3e-3)
    sched = torch.optim.lr_scheduler.StepLR(oa, 2, 0.1)                       # the reference's scheduler drives it unchanged
    for step in range(3):
        for plist in (a, b):
            for p, g in zip(plist, _grads(2, step)):
                p.grad = g.to(DEV)
        oa.step(); ob.step(); sched.step()
    assert oa.param_groups[0]["lr"] == pytest.approx(3e-4)
    ob.param_groups[0]["lr"] = 3e-4
    # ours -> torch and torch -> ours, then one more step on each side from the other's state
    sa, sb = copy.deepcopy(oa.state_dict()), copy.deepcopy(ob.state_dict())
    assert set(sa["param_groups"][0]) - {"initial_lr"} == set(sb["param_groups"][0])          # (initial_lr: the scheduler's)
    a2 = [torch.nn.Parameter(p.detach().clone()) for p in b]
    b2 = [torch.nn.Parameter(p.detach().clone()) for p in a]
    oa2, ob2 = Adam(a2, 1.0), torch.optim.Adam(b2, 1.0)
    sa_file = copy.deepcopy(sa)          # (torch's load_state_dict adopts CPU `step` tensors without copying them)
    oa2.load_state_dict(sb)
    ob2.load_state_dict(sa)
    assert oa2.param_groups[0]["lr"] == pytest.approx(3e-4) and ob2.param_groups[0]["lr"] == pytest.approx(3e-4)
    gs = _grads(2, 9)
    for plist in (a, b, a2, b2):
        for p, g in zip(plist, gs):
            p.grad = g.to(DEV)
    for o in (oa, ob, oa2, ob2):
        o.step()
    torch.cuda.synchronize()
    for i in range(len(SHAPES)):
        assert _rel(a2[i].detach(), b[i].detach()) <= 2e-6, i       # ours continued from torch's file == torch continued
        assert _rel(b2[i].detach(), a[i].detach()) <= 2e-6, i       # torch continued from our file == ours continued
        assert float(oa2.state[a2[i]]["step"]) == 4.0
    # the file carries the flags torch.optim.Adam(params, lr) would have written, not this class's internals: a reference-side
    # optimiser on CPU parameters loads it and steps (capturable=True would make torch refuse CPU parameters)
    assert sa_file["param_groups"][0]["capturable"] is False and sa_file["param_groups"][0]["fused"] is None
    assert all(st["step"].device.type == "cpu" for st in sa_file["state"].values())
    c = [torch.nn.Parameter(p.detach().cpu().clone()) for p in b2]
    oc = torch.optim.Adam(c, 1.0)
    oc.load_state_dict(sa_file)
    for p, g in zip(c, gs):
        p.grad = g.clone()
    oc.step()
    assert all(float(oc.state[p]["step"]) == 4.0 for p in c)


def test_trainer_steps_with_depthcore_adam_match_torch_adam():
    """Two trainers from the same state, one per optimiser.  The first update starts from identical weights and bit-identical
    gradients (deterministic kernels), so the weights after it may differ by Adam's own rounding only.  From then on the
    comparison is loose by nature: min-reprojection, auto-masking, ReLU and max-pool are discrete choices, a 2e-9 weight
    difference moves gradients by ~1e-7 (measured: tools/dbg_adam.py), and Adam normalises every element's update to ~lr --
    so later steps are held to the same loss and to the distance the updates can cover."""
    import trainer as T
    from depthcore.synthetic import synthetic_batch
    B, H, W = 2, 64, 128
    t1 = T.Trainer(T.default_options(batch_size=B, height=H, width=W), device=DEV, seed=3)
    t2 = T.Trainer(T.default_options(batch_size=B, height=H, width=W, torch_adam=1), device=DEV, seed=3)
    assert type(t1.model_optimizer).__module__ == "depthcore.optim" and isinstance(t2.model_optimizer, torch.optim.Adam)
    for k in t1.models:
        t2.models[k].load_state_dict(t1.models[k].state_dict())
    t1.set_train(); t2.set_train()
    inputs = synthetic_batch(B, H, W, torch.device(DEV), seed=4)
    lr = t1.opt.learning_rate
    for step in range(3):
        l1, l2 = t1.train_step(dict(inputs))[1], t2.train_step(dict(inputs))[1]
        torch.cuda.synchronize()
        v1, v2 = float(l1["loss"].detach()), float(l2["loss"].detach())
        assert abs(v1 - v2) <= (0.0 if step == 0 else 1e-5) * abs(v2), (step, v1, v2)       # step 0: the same forward to the bit
        for k in t1.models:
            for (n, p1), (_, p2) in zip(t1.models[k].named_parameters(), t2.models[k].named_parameters()):
                d = float((p1.detach() - p2.detach()).abs().max())
                if step == 0:
                    assert p1.grad is None or torch.equal(p1.grad, p2.grad), (k, n)
                    assert d <= 1e-3 * lr, (k, n, d)            # one update of ~lr per element, equal to ~1e-5 of itself
                assert d <= 2 * (step + 1) * lr, (k, n, d)      # nothing can be further apart
