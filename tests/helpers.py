"""Shared helpers for the parity tests (oracle = checker, HIP path = thing under test)."""
import numpy as np
import torch

from oracle import ref_cpu as R


def T(a):
    return torch.from_numpy(np.asarray(a))


def close(a, b, rtol=1e-5, atol=1e-6, msg=""):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol, err_msg=msg)


def close_frac(a, b, rtol, atol, bad=5e-3, msg="", atol_rel=0.0):
    """Pointwise tolerance with a small outlier budget: floor / argmin / clamp make the path
    piecewise-smooth, so 1-ulp differences legitimately flip isolated pixels."""
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    # atol_rel: absolute slack as a fraction of the tensor's mean magnitude (sums with cancellation)
    viol = np.abs(a - b) > atol + atol_rel * np.abs(b).mean() + rtol * np.abs(b)
    assert viol.mean() <= bad, "%s: %.4f%% of elements out of tolerance (max abs diff %.3e)" % (
        msg, 100 * viol.mean(), np.abs(a - b).max())


def rel_l2(a, b):
    a = a.detach().cpu().double().numpy() if isinstance(a, torch.Tensor) else np.asarray(a, np.float64)
    b = b.detach().cpu().double().numpy() if isinstance(b, torch.Tensor) else np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def oracle_photo(inputs, disps, Ts, opt, noise, dtype=torch.float32):
    """Run the CPU oracle's generate_images_pred + compute_losses; return losses, outputs, grads."""
    cast = (lambda t: t.to(dtype)) if dtype != torch.float32 else (lambda t: t)
    inputs = {k: cast(v) for k, v in inputs.items()}
    disps = [cast(d.detach().cpu()).requires_grad_() for d in disps]
    Ts = [cast(t.detach().cpu()).requires_grad_() for t in Ts]
    outputs = {("disp", s): disps[s] for s in range(len(disps))}
    outputs[("cam_T_cam", 0, -1)] = Ts[0]
    outputs[("cam_T_cam", 0, 1)] = Ts[1]
    R.generate_images_pred(inputs, outputs, opt)
    noise = None if noise is None else [cast(n) for n in noise]
    losses = R.compute_losses(inputs, outputs, opt, noise)
    grads = torch.autograd.grad(losses["loss"], disps + Ts)
    return losses, outputs, grads[:len(disps)], grads[len(disps):]


def random_poses(B, seed, scale=0.01):
    g = torch.Generator().manual_seed(seed)
    out = []
    for f in (-1, 1):
        aa = scale * torch.randn(B, 1, 3, generator=g)
        tr = scale * torch.randn(B, 1, 3, generator=g)
        out.append(R.transformation_from_parameters(aa, tr, invert=(f < 0)))
    return out
