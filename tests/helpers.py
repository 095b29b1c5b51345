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


# ---- data step (row f4): cases shared by tests/golden/make_golden_data.py and the tests -------------------------------
# name: (native H, native W, height, width, num_scales, flip, jitter order or None, factors (b, c, s, hue), seed)
DATA_CASES = {
    "kitti_small": (94, 311, 48, 160, 4, True, (2, 0, 3, 1), (1.13, 0.86, 1.19, -0.07), 1),
    "down_noflip": (75, 250, 32, 96, 3, False, (3, 1, 0, 2), (0.81, 1.2, 0.8, 0.1), 2),
    "upscale": (20, 30, 32, 64, 2, True, None, None, 3),
    "same_width": (60, 64, 32, 64, 2, True, (1, 2, 3, 0), (1.0, 0.95, 1.05, 0.0), 4),
    "interp_only": (40, 100, 16, 48, 2, False, (0, 1, 2, 3), (0.9, 0.85, 0.8, -0.1), 5),
}


def data_case_image(h, w, seed):
    """uint8 (h, w, 3): smooth ramps + blocks of saturated colours + noise, so clips, grey pixels and every hue sextant occur."""
    rng = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    base = np.stack([255.0 * xx / max(w - 1, 1), 255.0 * yy / max(h - 1, 1), 127.5 + 127.5 * np.sin(xx / 7.0 + yy / 5.0)], -1)
    img = base + rng.normal(0, 25, (h, w, 3))
    blocks = rng.randint(0, 2, (h // 8 + 1, w // 8 + 1, 3)) * 255
    mask = rng.rand(h // 8 + 1, w // 8 + 1) < 0.25
    big = np.repeat(np.repeat(blocks, 8, 0), 8, 1)[:h, :w]
    bigm = np.repeat(np.repeat(mask, 8, 0), 8, 1)[:h, :w]
    img = np.where(bigm[..., None], big, img)
    grey = rng.rand(h, w) < 0.05
    img = np.where(grey[..., None], img[..., :1], img)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def pass_rate_1e3(a, b, rel=1e-3):
    """Fraction of elements that meet the PLAIN pointwise tolerance of BASELINE.json's north_star -- |a - b| <= 1e-3 |b|, with
    1e-3 of the tensor's rms as the absolute floor for elements near zero.  Reported next to the calibrated / normwise gates so
    that the distance to that tolerance is a number."""
    if isinstance(a, torch.Tensor):
        a = a.detach().cpu().numpy()
    if isinstance(b, torch.Tensor):
        b = b.detach().cpu().numpy()
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    rms = float(np.sqrt((b * b).mean())) if b.size else 0.0
    ok = np.abs(a - b) <= rel * (np.abs(b) + rel * rms + 1e-30)
    return float(ok.mean()) if ok.size else 1.0
