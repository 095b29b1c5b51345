"""Imposed kinks for comparing piecewise-smooth networks across precisions (TEST INFRASTRUCTURE, see oracle/__init__.py).

A ReLU / max-pool network is only piecewise smooth: a pre-activation within rounding of zero (or two window entries within
rounding of each other) is routed differently by an fp32 and an fp64 evaluation, and on small maps one such element moves
every upstream gradient by a fraction of a percent -- which says nothing about the kernels under test.  The HIP path
records the decisions it took (tests/kink_tape.py: the output of every fused ReLU, the argmax code of the max-pool);
`ForcedKinks` replays those decisions inside the oracle, so both sides evaluate the SAME smooth function and every
feature map and gradient can be held to the rounding-level bound with no exception list.  Where the oracle's own decision
differs from the imposed one, the size of the pre-activation (or of the gap between the two window entries) is recorded:
the tests assert that it is at rounding level, i.e. that the recorded decisions are legitimate ones.

Reference semantics restated: `F.relu` (networks/resnet_encoder.py:91 via torchvision, networks/pose_decoder.py:45-52),
`nn.MaxPool2d(3, 2, 1)` (torchvision ResNet.maxpool, reached from networks/resnet_encoder.py:93).
"""
import torch
import torch.nn.functional as F


class Kinks:
    """The plain functions (no imposition)."""

    def relu(self, x):
        return F.relu(x)

    def max_pool(self, x):
        return F.max_pool2d(x, 3, 2, 1)


class ForcedKinks(Kinks):
    """entries: the tape of one module call, in call order -- ("relu", y) with y the HIP op's post-ReLU output, or
    ("maxpool", code) with code[n,c,oy,ox] = ky * 3 + kx of the selected window entry (dc_maxpool3x3s2_fwd).
    `batch`: slice of the recorded batch this evaluation covers (the HIP path stacks independent sub-batches)."""

    def __init__(self, entries, batch=None, keep_pre=False):
        self.entries = list(entries)
        self.batch = batch if batch is not None else slice(None)
        self.cursor = 0
        self.disagree = []       # (kind, index, number of differing decisions, max |margin| / rms of the tensor)
        self.pre = {} if keep_pre else None      # entry index -> the tensor the decision was taken on (detached)

    def _next(self, kind):
        assert self.cursor < len(self.entries), "kink tape exhausted: the oracle makes more %s decisions than were recorded" % kind
        k, t = self.entries[self.cursor]
        assert k == kind, "kink tape out of step: recorded %s, oracle asks for %s (entry %d)" % (k, kind, self.cursor)
        self.cursor += 1
        return t.detach().cpu()[self.batch]

    def done(self):
        assert self.cursor == len(self.entries), "kink tape has %d unused entries" % (len(self.entries) - self.cursor)

    def relu(self, x):
        y = self._next("relu")
        assert y.shape == x.shape, (tuple(y.shape), tuple(x.shape))
        mask = y > 0
        xd = x.detach()
        if self.pre is not None:
            self.pre[self.cursor - 1] = xd
        diff = (xd > 0) != mask
        n = int(diff.sum())
        if n:
            rms = float(xd.double().pow(2).mean().sqrt())
            self.disagree.append(("relu", self.cursor - 1, n, float(xd[diff].abs().max()) / max(rms, 1e-30)))
        return x * mask.to(x.dtype)

    def max_pool(self, x):
        code = self._next("maxpool").long()
        N, C, H, W = x.shape
        Ho, Wo = code.shape[2], code.shape[3]
        if self.pre is not None:
            self.pre[self.cursor - 1] = x.detach()
        oy = torch.arange(Ho).view(1, 1, Ho, 1)
        ox = torch.arange(Wo).view(1, 1, 1, Wo)
        iy, ix = oy * 2 - 1 + code // 3, ox * 2 - 1 + code % 3
        assert int(iy.min()) >= 0 and int(iy.max()) < H and int(ix.min()) >= 0 and int(ix.max()) < W
        idx = (iy * W + ix).reshape(N, C, Ho * Wo)
        out = x.reshape(N, C, H * W).gather(2, idx).reshape(N, C, Ho, Wo)
        ref, ridx = F.max_pool2d(x.detach(), 3, 2, 1, return_indices=True)
        diff = ridx.reshape(N, C, Ho * Wo) != idx
        n = int(diff.sum())
        if n:
            rms = float(x.detach().double().pow(2).mean().sqrt())
            gap = (ref.reshape(N, C, Ho * Wo) - out.detach().reshape(N, C, Ho * Wo))[diff].abs().max()
            self.disagree.append(("maxpool", self.cursor - 1, n, float(gap) / max(rms, 1e-30)))
        return out


def uncalibrated_disagreements(kn64, kn32, factor=4.0, floor=1e-6):
    """The disagreements of `kn64` (ForcedKinks of an fp64 evaluation, keep_pre=True) that are NOT explained by fp32
    rounding: a recorded fp32 decision may differ from the fp64 one at an element only if the fp64 value there is within the
    recorded path's own rounding error of zero.  That error is calibrated per tensor by the oracle itself: `kn32` is the same
    evaluation in fp32, and the margin allowed is `factor` x its largest deviation from fp64 on that tensor (+ `floor`),
    both relative to the tensor's rms."""
    bad = []
    for kind, idx, n, margin in kn64.disagree:
        a, b = kn64.pre[idx].double(), kn32.pre[idx].double()
        rms = float(a.pow(2).mean().sqrt())
        e32 = float((a - b).abs().max()) / max(rms, 1e-30)
        if margin > factor * e32 + floor:
            bad.append((kind, idx, n, margin, e32))
    return bad
