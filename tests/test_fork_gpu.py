"""GradFork (depthcore.ops): the gradient of a residual block's input -- conv1's data gradient + the skip's gradient -- summed in
the store epilogue of conv1's data-gradient kernel (dc_wino3x3_dgrad_add / dc_conv1x1_dgrad_add) instead of by an autograd
add (torchvision BasicBlock / Bottleneck behind reference networks/resnet_encoder.py:74-98).  One fp32 addition either way,
so the results must be IDENTICAL to the un-forked path's, for the input and for every parameter."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _run(block_cls, inplanes, planes, shape, fork, seed=0, ds_stride=0):
    import torch.nn as nn
    from networks import resnet_encoder as RE
    torch.manual_seed(seed)
    ds = None
    if ds_stride:       # a downsample branch as ResNetTrunk._make_layer builds it (v1.5: the block's stride sits on conv2)
        ds = nn.Sequential(nn.Conv2d(inplanes, planes * block_cls.expansion, 1, ds_stride, bias=False), nn.BatchNorm2d(planes * block_cls.expansion))
    blk = (block_cls(inplanes, planes, ds_stride, ds) if ds_stride else block_cls(inplanes, planes)).to(DEV).train()
    x = torch.randn(*shape, device=DEV).requires_grad_()
    so = ds_stride or 1
    cot = torch.randn(shape[0], planes * block_cls.expansion, shape[2] // so, shape[3] // so, device=DEV)
    old = RE.GRAD_FORK
    RE.GRAD_FORK = fork
    try:
        made = []
        orig = RE._ops.GradFork
        RE._ops.GradFork = lambda **kw: made.append(orig(**kw)) or made[-1]
        y = blk(x * 1.0)                       # (x * 1.0: the block input is a non-leaf, as inside the trunk)
        (y * cot).sum().backward()
    finally:
        RE.GRAD_FORK = old
        RE._ops.GradFork = orig
    torch.cuda.synchronize()
    return x.grad.clone(), {n: p.grad.clone() for n, p in blk.named_parameters()}, y.detach().clone(), made


@pytest.mark.parametrize("kind,shape", [("basic", (4, 64, 24, 80)), ("basic", (2, 128, 12, 40)), ("basic", (12, 512, 6, 20)),
                                        ("bottleneck", (2, 256, 16, 64)), ("bottleneck", (4, 512, 8, 32))])
def test_forked_block_gradients_equal_autograd_sum(kind, shape):
    from networks import resnet_encoder as RE
    cls = RE.BasicBlock if kind == "basic" else RE.Bottleneck
    planes = shape[1] // cls.expansion
    gx1, gp1, y1, made1 = _run(cls, shape[1], planes, shape, True)
    gx0, gp0, y0, made0 = _run(cls, shape[1], planes, shape, False)
    assert len(made1) == 1 and not made0                                     # the fork was really used / really off
    assert made1[0].addend is None and not made1[0].armed                     # parked and collected exactly once
    assert torch.equal(y1, y0)
    assert torch.equal(gx1, gx0), float((gx1 - gx0).abs().max())
    for n in gp0:
        assert torch.equal(gp1[n], gp0[n]), n


def test_block_with_downsample_or_eval_gets_no_fork_and_second_backward_raises():
    from depthcore._lib import DepthcoreError
    from networks import resnet_encoder as RE
    import torch.nn as nn
    x = torch.randn(2, 64, 16, 32, device=DEV).requires_grad_()
    ds = nn.Sequential(nn.Conv2d(64, 128, 1, 2, bias=False), nn.BatchNorm2d(128))
    assert RE._fork_for(RE.BasicBlock(64, 128, 2, ds).to(DEV).train(), x) is None      # downsample branch: two convolutions share x
    assert RE._fork_for(RE.BasicBlock(64, 64).to(DEV).eval(), x) is None
    assert RE._fork_for(RE.BasicBlock(64, 64).to(DEV).train(), x.detach()) is None     # nothing upstream needs the gradient
    blk = RE.BasicBlock(64, 64).to(DEV).train()
    y = blk(x * 1.0)
    y.sum().backward(retain_graph=True)
    with pytest.raises(DepthcoreError):
        y.sum().backward()                                                             # one backward per forward


@pytest.mark.parametrize("shape,planes,ds_stride", [((2, 64, 16, 64), 64, 1), ((2, 256, 16, 64), 128, 2), ((4, 512, 8, 32), 256, 2)])
def test_pair_fork_of_bottleneck_with_downsample_equals_autograd_sum(shape, planes, ds_stride):
    """x feeds conv1 AND the 1x1 `downsample` (layer1.0: stride 1; layer2-4.0: stride 2): whichever data gradient is computed
    second adds the first (in its store epilogue for stride 1, by the library's in-place pass for stride 2)."""
    from networks import resnet_encoder as RE
    gx1, gp1, y1, made1 = _run(RE.Bottleneck, shape[1], planes, shape, True, ds_stride=ds_stride)
    gx0, gp0, y0, made0 = _run(RE.Bottleneck, shape[1], planes, shape, False, ds_stride=ds_stride)
    assert len(made1) == 1 and made1[0].pair and made1[0].arrived == 2 and made1[0].addend is None and not made0
    assert torch.equal(y1, y0)
    assert torch.equal(gx1, gx0), float((gx1 - gx0).abs().max())
    for n in gp0:
        assert torch.equal(gp1[n], gp0[n]), n


def test_decoder_output_left_out_of_the_loss_is_reported_not_lost():
    """the depth decoder's pair fork between dispconv(i) and upconv(i-1, 0) assumes both backwards run.  A loss
    that uses only the finest scale leaves a parked gradient nobody collects: `assert_no_dangling_sums()` (the
    Trainer calls it after every backward) must raise instead of letting the levels below train on an incomplete gradient; a loss
    over every scale passes."""
    import numpy as np
    import networks
    from depthcore import ops
    from depthcore._lib import DepthcoreError
    torch.manual_seed(0)
    nce = np.array([64, 64, 128, 256, 512])
    dec = networks.DepthDecoder(nce).to(DEV).train()
    feats = [torch.randn(2, c, 64 >> (i + 1), 128 >> (i + 1), device=DEV).requires_grad_() for i, c in enumerate(nce)]
    out = dec([f * 1.0 for f in feats])
    sum(out[("disp", s)].mean() for s in range(4)).backward()
    ops.assert_no_dangling_sums()                                  # every reader ran: nothing parked
    out = dec([f * 1.0 for f in feats])
    out[("disp", 0)].mean().backward()                             # scales 1..3 left out: dispconv(1..3) never run backward
    with pytest.raises(DepthcoreError, match="GradFork"):
        ops.assert_no_dangling_sums()
    ops.assert_no_dangling_sums()                                  # (reported once, then cleared)
