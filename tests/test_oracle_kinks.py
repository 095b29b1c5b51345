"""oracle/kinks.py on the CPU: imposing a network's OWN recorded ReLU / max-pool decisions changes nothing, the tape
bookkeeping (order, sub-batch slices, exhaustion) is checked, and decisions recorded in fp32 differ from the fp64 ones only
on near-ties.  (The GPU tests impose the HIP path's decisions the same way: tests/test_encoder_gpu.py.)"""
import pytest
import torch
import torch.nn.functional as F

from oracle.kinks import ForcedKinks, Kinks
from oracle.resnet_ref import resnet_encoder_forward


class Recorder(Kinks):
    """What tests/kink_tape.py records on the GPU, restated on the CPU: post-ReLU outputs and dc_maxpool codes."""

    def __init__(self):
        self.entries = []

    def relu(self, x):
        y = F.relu(x)
        self.entries.append(("relu", y.detach()))
        return y

    def max_pool(self, x):
        y, idx = F.max_pool2d(x, 3, 2, 1, return_indices=True)
        W = x.shape[3]
        Ho, Wo = y.shape[2], y.shape[3]
        iy, ix = idx // W, idx % W
        oy = torch.arange(Ho).view(1, 1, Ho, 1)
        ox = torch.arange(Wo).view(1, 1, 1, Wo)
        code = (iy - (oy * 2 - 1)) * 3 + (ix - (ox * 2 - 1))
        self.entries.append(("maxpool", code.to(torch.uint8)))
        return y


def _state(num_layers, nimg=1, seed=0):
    import networks
    torch.manual_seed(seed)
    enc = networks.ResnetEncoder(num_layers, False, num_input_images=nimg)
    return {k: v.detach().clone() for k, v in enc.state_dict().items()}


@pytest.mark.parametrize("num_layers", [18, 50])
def test_imposing_own_decisions_is_the_identity(num_layers):
    st = _state(num_layers)
    x = torch.rand(2, 3, 32, 64, generator=torch.Generator().manual_seed(1))
    rec = Recorder()
    plain = resnet_encoder_forward(st, x, num_layers, kinks=rec)
    assert [k for k, _ in rec.entries].count("relu") == {18: 17, 50: 49}[num_layers]
    forced = ForcedKinks(rec.entries)
    again = resnet_encoder_forward(st, x, num_layers, kinks=forced)
    forced.done()
    assert not forced.disagree
    for a, b in zip(plain, again):
        assert torch.equal(a, b)
    # gradients through the imposed gather / mask equal the plain ones
    w = st["encoder.conv1.weight"].clone().requires_grad_()
    st2 = dict(st, **{"encoder.conv1.weight": w})
    g_plain, = torch.autograd.grad(sum(f.sum() for f in resnet_encoder_forward(st2, x, num_layers)), w)
    g_forced, = torch.autograd.grad(sum(f.sum() for f in resnet_encoder_forward(st2, x, num_layers, kinks=ForcedKinks(rec.entries))), w)
    assert torch.allclose(g_plain, g_forced, rtol=1e-5, atol=1e-7)


def test_sub_batch_slices_and_bookkeeping():
    st = _state(18)
    x = torch.rand(4, 3, 32, 64, generator=torch.Generator().manual_seed(2))
    # two independent sub-batches stacked along the batch, as the HIP pose encoder records them
    recs = [Recorder(), Recorder()]
    parts = [resnet_encoder_forward(st, x[2 * g:2 * g + 2], 18, kinks=recs[g]) for g in range(2)]
    tape = [(k, torch.cat([a, b], 0)) for (k, a), (_, b) in zip(recs[0].entries, recs[1].entries)]
    for g in range(2):
        fk = ForcedKinks(tape, slice(2 * g, 2 * g + 2))
        out = resnet_encoder_forward(st, x[2 * g:2 * g + 2], 18, kinks=fk)
        fk.done()
        assert not fk.disagree and all(torch.equal(a, b) for a, b in zip(out, parts[g]))
    short = ForcedKinks(tape[:-1], slice(0, 2))
    with pytest.raises(AssertionError, match="exhausted"):
        resnet_encoder_forward(st, x[:2], 18, kinks=short)
    longer = ForcedKinks(tape + [tape[-1]], slice(0, 2))
    resnet_encoder_forward(st, x[:2], 18, kinks=longer)
    with pytest.raises(AssertionError, match="unused"):
        longer.done()


def test_fp32_decisions_differ_from_fp64_only_on_near_ties():
    st = _state(18)
    x = torch.rand(2, 3, 64, 128, generator=torch.Generator().manual_seed(3))
    rec = Recorder()
    resnet_encoder_forward(st, x, 18, kinks=rec)
    st64 = {k: (v.double() if v.is_floating_point() else v) for k, v in st.items()}
    from oracle.kinks import uncalibrated_disagreements
    fk = ForcedKinks(rec.entries, keep_pre=True)
    resnet_encoder_forward(st64, x.double(), 18, kinks=fk)
    fk.done()
    fk32 = ForcedKinks(rec.entries, keep_pre=True)
    resnet_encoder_forward(st, x, 18, kinks=fk32)
    assert not uncalibrated_disagreements(fk, fk32), fk.disagree
    # a decision flipped away from a tie is caught
    big = rec.entries[5][1].clone()
    pos = (big > big.mean() + big.std()).nonzero()[0]
    big[tuple(pos)] = 0.0
    tampered = list(rec.entries)
    tampered[5] = ("relu", big)
    fk = ForcedKinks(tampered, keep_pre=True)
    resnet_encoder_forward(st64, x.double(), 18, kinks=fk)
    fk32 = ForcedKinks(tampered, keep_pre=True)
    resnet_encoder_forward(st, x, 18, kinks=fk32)
    assert uncalibrated_disagreements(fk, fk32)
