"""Row f4 (data step), CPU side: the numpy restatement (oracle/data_ref.py) against Pillow's own outputs -- committed
vectors (tests/golden/data_pillow.npz, made by tests/golden/make_golden_data.py) and, when Pillow is importable, live on
the KITTI size chain and the whole RGB cube -- and libdepthcore's HOST coefficient table against the restatement."""
import numpy as np
import pytest

from helpers import DATA_CASES, data_case_image
from oracle import data_ref as D


def _oracle_case(case):
    hn, wn, h, w, scales, flip, order, factors, seed = DATA_CASES[case]
    img = data_case_image(hn, wn, seed)
    if flip:
        img = img[:, ::-1]
    out = []
    for s in range(scales):
        img = D.resize_lanczos(img, h >> s, w >> s)
        aug = D.color_jitter(img, order, factors) if order is not None else img
        out.append((img, aug))
    return out


@pytest.mark.parametrize("case", sorted(DATA_CASES))
def test_oracle_matches_pillow_vectors(golden, case):
    g = golden["data_pillow"]
    for s, (img, aug) in enumerate(_oracle_case(case)):
        assert np.array_equal(img, g["%s/color%d" % (case, s)]), (case, s, "resize")
        assert np.array_equal(aug, g["%s/aug%d" % (case, s)]), (case, s, "jitter")


def test_colour_conversions_match_pillow_vectors(golden):
    g = golden["data_pillow"]
    grid = np.arange(0, 256, 5, dtype=np.uint8)
    cube = np.stack(np.meshgrid(grid, grid, grid, indexing="ij"), -1).reshape(-1, len(grid), 3)
    assert np.array_equal(D.rgb_to_hsv(cube), g["cube/hsv"])
    assert np.array_equal(D.hsv_to_rgb(cube), g["cube/rgb_from_hsv"])
    assert np.array_equal(D.rgb_to_l(cube), g["cube/l"])


def test_preprocess_item_schema_and_values():
    """preprocess_item = flip, pyramid where scale s is resized from scale s-1, ToTensor (true /255), same jitter at every scale."""
    native = data_case_image(47, 155, 9)
    jit = ((1, 3, 0, 2), (1.1, 0.9, 1.15, 0.04))
    out = D.preprocess_item(native, 24, 80, num_scales=3, flip=True, jitter=jit)
    img = native[:, ::-1]
    for s in range(3):
        img = D.resize_lanczos(img, 24 >> s, 80 >> s)
        c = out[("color", s)]
        assert c.shape == (3, 24 >> s, 80 >> s) and c.dtype == np.float32
        assert np.array_equal(c, (img.transpose(2, 0, 1).astype(np.float32) / np.float32(255)))
        assert np.array_equal(out[("color_aug", s)], D.to_tensor(D.color_jitter(img, *jit)))
    plain = D.preprocess_item(native, 24, 80, num_scales=2)
    assert plain[("color_aug", 1)] is plain[("color", 1)]


# ------------------------------------------------------------------------------------------------------- live against Pillow
def test_oracle_vs_installed_pillow_kitti_chain():
    Image = pytest.importorskip("PIL.Image")
    from PIL import ImageEnhance
    lanczos = getattr(Image, "Resampling", Image).LANCZOS
    native = data_case_image(375, 1242, 11)
    pil = Image.fromarray(native)
    img = native
    for s in range(4):
        pil = pil.resize((640 >> s, 192 >> s), lanczos)
        img = D.resize_lanczos(img, 192 >> s, 640 >> s)
        assert np.array_equal(img, np.asarray(pil)), s
    pil0 = Image.fromarray(D.resize_lanczos(native, 192, 640))
    a = np.asarray(pil0)
    for f in (0.8, 0.93, 1.0, 1.07, 1.2):
        assert np.array_equal(D.adjust_brightness(a, f), np.asarray(ImageEnhance.Brightness(pil0).enhance(f))), f
        assert np.array_equal(D.adjust_contrast(a, f), np.asarray(ImageEnhance.Contrast(pil0).enhance(f))), f
        assert np.array_equal(D.adjust_saturation(a, f), np.asarray(ImageEnhance.Color(pil0).enhance(f))), f


def test_oracle_vs_installed_pillow_rgb_cube():
    Image = pytest.importorskip("PIL.Image")
    g = np.arange(256, dtype=np.uint8)
    for r0 in range(0, 256, 64):          # the whole cube, in four slabs to bound memory
        cube = np.stack(np.meshgrid(g[r0:r0 + 64], g, g, indexing="ij"), -1).reshape(-1, 256, 3)
        assert np.array_equal(D.rgb_to_hsv(cube), np.asarray(Image.fromarray(cube).convert("HSV"))), r0
        assert np.array_equal(D.hsv_to_rgb(cube), np.asarray(Image.fromarray(cube, "HSV").convert("RGB"))), r0


# --------------------------------------------------------------------------------------------- host side of the C ABI (no GPU)
@pytest.mark.parametrize("sizes", [(1242, 640), (375, 192), (640, 320), (192, 96), (80, 40), (24, 12), (30, 64), (1024, 1024),
                                   (1241, 1024), (376, 320), (7, 3), (3, 7), (1, 1)])
def test_host_resample_table_matches_oracle(sizes):
    from depthcore.data import resample_table
    bounds, kk = resample_table(*sizes)
    ob, ok = D.resample_coeffs(*sizes)
    assert kk.shape == ok.shape
    assert np.array_equal(bounds, ob)
    assert np.array_equal(kk, ok)
    assert np.all(kk.sum(1) > 0)


def test_sample_item_draw_order():
    """mono_dataset.py:139-140, :186-188: color-aug coin, flip coin, then four uniform factors and the shuffled order."""
    import random

    from depthcore.data import hue_shift, sample_item
    rng = random.Random(5)
    got = [sample_item(True, rng) for _ in range(50)]
    rng = random.Random(5)
    for flip, jit in got:
        aug = rng.random() > 0.5
        assert flip == (rng.random() > 0.5)
        if aug:
            f = [rng.uniform(0.8, 1.2), rng.uniform(0.8, 1.2), rng.uniform(0.8, 1.2), rng.uniform(-0.1, 0.1)]
            order = [0, 1, 2, 3]
            rng.shuffle(order)
            assert jit == (tuple(order), tuple(f))
        else:
            assert jit is None
    assert sample_item(False) == (False, None)
    assert hue_shift(-0.1) == 231 and hue_shift(0.05) == 12 and hue_shift(0.0) == 0
