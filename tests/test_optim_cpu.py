"""Host logic of depthcore.optim.Adam that needs no GPU: hyper-parameter validation, the torch.optim.Adam param_group keys,
and the loud failure off the GPU (there is no CPU update path)."""
import pytest
import torch


def test_param_group_keys_match_torch_adam_and_bad_hyperparameters_raise():
    from depthcore.optim import Adam
    p = [torch.nn.Parameter(torch.zeros(3))]
    ours, ref = Adam(p, 1e-4), torch.optim.Adam([torch.nn.Parameter(torch.zeros(3))], 1e-4)
    assert set(ours.param_groups[0]) == set(ref.param_groups[0])
    assert ours.param_groups[0]["betas"] == (0.9, 0.999) and ours.param_groups[0]["eps"] == 1e-8       # torch's defaults (reference uses them)
    for bad in (dict(lr=-1.0), dict(betas=(1.0, 0.999)), dict(betas=(0.9, -0.1)), dict(eps=-1e-8)):
        with pytest.raises(ValueError):
            Adam(p, **bad)


def test_cpu_parameters_fail_loudly():
    from depthcore.optim import Adam
    from depthcore._lib import DepthcoreError
    p = torch.nn.Parameter(torch.zeros(8))
    p.grad = torch.ones(8)
    with pytest.raises(DepthcoreError):
        Adam([p], 1e-3).step()
    assert torch.equal(p.detach(), torch.zeros(8))
