"""Row f4 (data step) on the GPU: libdepthcore's dc_data_* kernels through depthcore.data.GpuPreprocessor, BIT-EXACT against
Pillow's committed outputs (tests/golden/data_pillow.npz) and the numpy restatement (oracle/data_ref.py)."""
import numpy as np
import pytest
import torch

from helpers import DATA_CASES, data_case_image
from oracle import data_ref as D

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device("cuda:0")


@pytest.mark.parametrize("case", sorted(DATA_CASES))
def test_golden_cases_bit_exact(golden, case):
    from depthcore.data import GpuPreprocessor
    hn, wn, h, w, scales, flip, order, factors, seed = DATA_CASES[case]
    native = torch.from_numpy(data_case_image(hn, wn, seed)).to(_dev()).view(1, 1, hn, wn, 3).contiguous()
    pre = GpuPreprocessor(h, w, num_scales=scales, frame_idxs=(0,), device=_dev())
    out = pre(native, flips=[flip], jitters=[(order, factors) if order is not None else None])
    g = golden["data_pillow"]
    for s in range(scales):
        for key, name in ((("color", 0, s), "color"), (("color_aug", 0, s), "aug")):
            want = D.to_tensor(g["%s/%s%d" % (case, name, s)])
            got = out[key][0].cpu().numpy()
            assert got.dtype == np.float32 and got.shape == want.shape
            assert np.array_equal(got, want), (case, key, int((got != want).sum()))


@pytest.mark.parametrize("hw", [(192, 640), (320, 1024)])
def test_kitti_batch_vs_oracle(hw):
    """B = 3 items x 3 frames of native 375 x 1242, mixed flips / jitters: every byte of every scale equals the CPU pipeline."""
    from depthcore.data import GpuPreprocessor
    h, w = hw
    B, frames = 3, (0, -1, 1)
    rng = np.random.RandomState(0)
    native = np.stack([np.stack([data_case_image(375, 1242, 100 + 10 * f + b) for b in range(B)]) for f in range(3)])
    flips = [True, False, True]
    jitters = [((3, 0, 1, 2), (0.83, 1.17, 0.91, 0.093)), None, ((1, 2, 0, 3), (1.2, 0.8, 1.2, -0.1))]
    pre = GpuPreprocessor(h, w, frame_idxs=frames, device=_dev())
    out = pre(torch.from_numpy(native).to(_dev()), flips, jitters)
    assert len(out) == 2 * 3 * 4 + 3                          # + ("color_packed", f, 0) of the three frames
    del rng
    # the pixel-interleaved RGBx copy the photometric kernels gather from: the same values as ("color", f, 0), x = 0
    from depthcore import ops
    for f in frames:
        pk = out[("color_packed", f, 0)]
        assert pk.shape == (B, h, w, 4) and pk.is_contiguous() and pk.data_ptr() % 16 == 0
        assert torch.equal(pk, ops.pack_rgbx(out[("color", f, 0)]))
        assert torch.equal(pk[..., :3].permute(0, 3, 1, 2), out[("color", f, 0)]) and not pk[..., 3].any()
    for b in range(B):
        for i, f in enumerate(frames):
            want = D.preprocess_item(native[i, b], h, w, 4, flips[b], jitters[b])
            for s in range(4):
                for n in ("color", "color_aug"):
                    got = out[(n, f, s)][b].cpu().numpy()
                    assert np.array_equal(got, want[(n, s)]), (b, f, s, n, int((got != want[(n, s)]).sum()))
    assert out[("color", 0, 0)].shape == (B, 3, h, w) and out[("color", 0, 0)].is_contiguous()


def test_hue_over_the_rgb_cube():
    """Every RGB triple through rgb -> hsv -> shift -> rgb on the device, against the restatement (itself pinned on the whole
    cube against Pillow in tests/test_data_cpu.py)."""
    from depthcore import _lib
    L = _lib.lib()
    g = np.arange(256, dtype=np.uint8)
    cube = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(1, -1, 3)
    npix = cube.shape[1]
    dev = _dev()
    src = torch.from_numpy(cube).to(dev)
    sums = torch.zeros(1, dtype=torch.int64, device=dev)
    for shift in (0, 37, 231):
        img = src.clone()
        steps = torch.tensor([[3, -1, -1, -1]], dtype=torch.int32, device=dev)
        params = torch.tensor([[float(shift), 0, 0, 0]], dtype=torch.float32, device=dev)
        _lib.check(L.dc_data_jitter(img.data_ptr(), 1, npix, steps.data_ptr(), params.data_ptr(), sums.data_ptr(), _lib.stream(img)), "jit")
        hsv = D.rgb_to_hsv(cube[0].reshape(4096, 4096, 3))
        hsv[..., 0] = (hsv[..., 0].astype(np.int32) + shift) & 0xFF
        want = D.hsv_to_rgb(hsv).reshape(-1, 3)
        got = img.cpu().numpy()[0]
        assert np.array_equal(got, want), (shift, int((got != want).any(-1).sum()))


def test_blend_ops_all_factors_and_contrast_mean():
    """brightness / contrast / saturation over a fine grid of factors on both sides of 1 (truncating and clipping blends),
    two images per launch with different ops so the per-image dispatch and the exact mean are exercised."""
    from depthcore import _lib
    L = _lib.lib()
    dev = _dev()
    a, b = data_case_image(96, 320, 21), data_case_image(96, 320, 22)
    for op, fn in ((0, D.adjust_brightness), (1, D.adjust_contrast), (2, D.adjust_saturation)):
        for f in np.linspace(0.8, 1.2, 9):
            f = float(f)
            other = (op + 1) % 3
            img = torch.from_numpy(np.stack([a, b])).to(dev)
            steps = torch.tensor([[-1, op, -1, -1], [other, -1, -1, -1]], dtype=torch.int32, device=dev)
            params = torch.tensor([[0, f, 0, 0], [f, 0, 0, 0]], dtype=torch.float32, device=dev)
            sums = torch.empty(2, dtype=torch.int64, device=dev)
            _lib.check(L.dc_data_jitter(img.data_ptr(), 2, 96 * 320, steps.data_ptr(), params.data_ptr(), sums.data_ptr(), _lib.stream(img)),
                       "jit")
            got = img.cpu().numpy()
            assert np.array_equal(got[0], fn(a, f)), (op, f)
            assert np.array_equal(got[1], (D.adjust_brightness, D.adjust_contrast, D.adjust_saturation)[other](b, f)), (other, f)


def test_bad_arguments_are_refused():
    from depthcore import _lib
    from depthcore.data import GpuPreprocessor
    L = _lib.lib()
    pre = GpuPreprocessor(32, 64, device=_dev())
    with pytest.raises(_lib.DepthcoreError):
        pre(torch.zeros(3, 1, 40, 80, 3, dtype=torch.float32, device=_dev()))
    with pytest.raises(_lib.DepthcoreError):
        pre(torch.zeros(2, 1, 40, 80, 3, dtype=torch.uint8, device=_dev()))
    with pytest.raises(_lib.DepthcoreError):
        GpuPreprocessor(32, 64, device="cpu")
    x = torch.zeros(1, 8, 8, 3, dtype=torch.uint8, device=_dev())
    t = torch.zeros(64, dtype=torch.int32, device=_dev())
    assert L.dc_data_resize_axis(x.data_ptr(), x.data_ptr(), 1, 8, 8, 4, 1, t.data_ptr(), t.data_ptr(), 3, None, None) == -1   # wrong ksize
    assert L.dc_data_resize_axis(x.data_ptr(), x.data_ptr(), 1, 8, 8, 4, 2, t.data_ptr(), t.data_ptr(), 13, None, None) == -1  # axis
    assert L.dc_data_to_tensor(None, x.data_ptr(), 1, 64, None) == -1
