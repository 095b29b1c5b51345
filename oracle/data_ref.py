"""CPU restatement of the reference's per-item data step: datasets/mono_dataset.py:92-211
(`preprocess` + the image part of `__getitem__`) -- horizontal flip, the Lanczos ("ANTIALIAS") resize pyramid, ToTensor
and torchvision's ColorJitter on PIL images.

TEST INFRASTRUCTURE (see oracle/__init__.py).  The arithmetic lives in two third-party dependencies that are NOT under
/root/reference: Pillow (unpinned; its `Image.resize`, `Image.blend`, `convert("L" / "HSV" / "RGB")`) and torchvision
(absent here; `transforms.ColorJitter` / `functional_pil.adjust_*`, version unpinned, pre-0.13 era API).  Their published
integer algorithms are restated here in numpy:
  * resize: Pillow's two-pass 8-bit resampling (libImaging/Resample.c): double-precision Lanczos coefficients normalised
    per output pixel, quantised to 22 fractional bits, accumulated in int32 with rounding, clipped to 8 bits after EACH
    pass (horizontal first);
  * ColorJitter: torchvision applies brightness / contrast / saturation (PIL `ImageEnhance` = `Image.blend` of a degenerate
    image with the input) and hue (shift of the H channel of PIL's HSV conversion) in a random order with one parameter
    set per item.
The restatement is pinned against the installed Pillow itself (tests/test_data_cpu.py: bit-exact on random images, the
KITTI size chain 375x1242 -> 192x640 -> ... and the whole RGB cube for the HSV round trip); the torchvision glue (order
of operations, parameter ranges, `int(mean + 0.5)` of the contrast image) follows its published source by reading.
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def _lanczos(x):
    def sinc(v):
        if v == 0.0:
            return 1.0
        v = v * math.pi
        return math.sin(v) / v
    return sinc(x) * sinc(x / 3.0) if -3.0 <= x < 3.0 else 0.0


def resample_coeffs(in_size, out_size):
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc for the Lanczos filter (support 3) over the whole axis.
    -> bounds (out_size, 2) int32 [xmin, count], kk (out_size, ksize) int32."""
    scale = float(in_size) / out_size
    filterscale = max(scale, 1.0)
    support = 3.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [_lanczos((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = sum(w)                       # (left-to-right double sum, as the C loop)
        for x in range(xmax):
            v = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _clip8(acc):
    return np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)


def resample_axis(img, out_size, axis):
    """One 8-bit pass along `axis` of an (H, W, C) uint8 image."""
    in_size = img.shape[axis]
    bounds, kk = resample_coeffs(in_size, out_size)
    src = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.empty((out_size,) + src.shape[1:], np.uint8)
    for xx in range(out_size):
        x0, n = int(bounds[xx, 0]), int(bounds[xx, 1])
        acc = (1 << (PRECISION_BITS - 1)) + np.tensordot(kk[xx, :n].astype(np.int64), src[x0:x0 + n], axes=(0, 0))
        out[xx] = _clip8(acc)
    return np.moveaxis(out, 0, axis)


def resize_lanczos(img, out_h, out_w):
    """PIL `img.resize((out_w, out_h), Image.LANCZOS)` (= the removed `Image.ANTIALIAS`, datasets/mono_dataset.py:57) of an
    (H, W, 3) uint8 array: horizontal pass, then vertical pass (a pass is skipped when that size is unchanged)."""
    out = img
    if out.shape[1] != out_w:
        out = resample_axis(out, out_w, 1)
    if out.shape[0] != out_h:
        out = resample_axis(out, out_h, 0)
    return out


def to_tensor(img):
    """torchvision ToTensor on a uint8 HWC image: CHW float32, true division by 255."""
    return (img.astype(np.float32).transpose(2, 0, 1) / np.float32(255.0)).astype(np.float32)


# ---------------------------------------------------------------------------------------------------------------- ColorJitter
def _blend(degenerate, img, factor):
    """PIL Image.blend(degenerate, img, factor) for uint8 (libImaging/Blend.c): interpolation truncates, extrapolation
    (factor outside [0, 1]) clips then truncates; the arithmetic is single precision."""
    a = np.float32(factor)
    d, x = degenerate.astype(np.int32), img.astype(np.int32)
    t = (d.astype(np.float32) + a * (x - d).astype(np.float32)).astype(np.float32)
    if 0.0 <= factor <= 1.0:
        return t.astype(np.uint8)                                     # (UINT8) cast: truncation
    return np.where(t <= 0.0, 0, np.where(t >= 255.0, 255, t.astype(np.int32))).astype(np.uint8)


def rgb_to_l(img):
    """PIL convert("L"): ITU-R 601-2 luma in 16.16 fixed point with rounding."""
    r, g, b = (img[..., i].astype(np.int64) for i in range(3))
    return ((r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16).astype(np.uint8)


def adjust_brightness(img, f):
    return _blend(np.zeros_like(img), img, f)


def adjust_contrast(img, f):
    mean = int(rgb_to_l(img).astype(np.float64).mean() + 0.5)        # ImageEnhance.Contrast: int(ImageStat mean + 0.5)
    return _blend(np.full_like(img, mean), img, f)


def adjust_saturation(img, f):
    return _blend(np.repeat(rgb_to_l(img)[..., None], 3, axis=-1), img, f)


def rgb_to_hsv(img):
    """PIL convert("HSV") (libImaging/Convert.c rgb2hsv_row, after colorsys).  The C code keeps h, s, rc, gc, bc in `float`
    but its literals are `double`: every expression is evaluated in double and ROUNDED TO FLOAT when stored, and the final
    `(int)(h * 255.0)` / `(int)(s * 255.0)` are double products of those floats, truncated."""
    r, g, b = (img[..., i].astype(np.int32) for i in range(3))
    maxc, minc = np.maximum(np.maximum(r, g), b), np.minimum(np.minimum(r, g), b)
    cr = (maxc - minc).astype(np.float32)
    safe = np.where(cr == 0, np.float32(1), cr)
    s = (cr / np.where(maxc == 0, 1, maxc).astype(np.float32)).astype(np.float32)
    rc, gc, bc = (((maxc - c).astype(np.float32) / safe).astype(np.float32) for c in (r, g, b))
    rc64, gc64, bc64 = rc.astype(np.float64), gc.astype(np.float64), bc.astype(np.float64)
    h = np.where(r == maxc, (bc - gc).astype(np.float64), np.where(g == maxc, 2.0 + rc64 - bc64, 4.0 + gc64 - rc64)).astype(np.float32)
    h = np.fmod(h.astype(np.float64) / 6.0 + 1.0, 1.0).astype(np.float32)
    uh = np.clip((h.astype(np.float64) * 255.0).astype(np.int32), 0, 255)
    us = np.clip((s.astype(np.float64) * 255.0).astype(np.int32), 0, 255)
    grey = maxc == minc
    return np.stack([np.where(grey, 0, uh), np.where(grey, 0, us), maxc], -1).astype(np.uint8)


def hsv_to_rgb(hsv):
    """PIL HSV -> RGB (Convert.c hsv2rgb): sextant i = floor(h * 6 / 255) in double, f and fs stored as float, p / q / t
    = round() (half away from zero) of double products."""
    h, s, v = (hsv[..., i].astype(np.int32) for i in range(3))
    fh = h.astype(np.float64) * 6.0 / 255.0
    i = np.floor(fh).astype(np.int32)
    f = (fh - i).astype(np.float32).astype(np.float64)
    fs = (s.astype(np.float64) / 255.0).astype(np.float32).astype(np.float64)
    vf = v.astype(np.float64)

    def rnd(x):
        return np.clip(np.floor(x + 0.5).astype(np.int32), 0, 255)     # round(): half away from zero, x >= 0 here
    p = rnd(vf * (1.0 - fs))
    q = rnd(vf * (1.0 - fs * f))
    t = rnd(vf * (1.0 - fs * (1.0 - f)))
    i6 = i % 6
    r = np.choose(i6, [v, q, p, p, t, v])
    g = np.choose(i6, [t, v, v, q, p, p])
    b = np.choose(i6, [p, p, t, v, v, q])
    grey = s == 0
    return np.stack([np.where(grey, v, r), np.where(grey, v, g), np.where(grey, v, b)], -1).astype(np.uint8)


def adjust_hue(img, hue_factor):
    """torchvision functional_pil.adjust_hue: H (uint8) += uint8(hue_factor * 255) with wrap-around."""
    hsv = rgb_to_hsv(img)
    hsv[..., 0] = (hsv[..., 0].astype(np.int32) + int(np.uint8(int(hue_factor * 255) & 0xFF))) & 0xFF
    return hsv_to_rgb(hsv)


JITTER_OPS = (adjust_brightness, adjust_contrast, adjust_saturation, adjust_hue)


def color_jitter(img, order, factors):
    """`order`: permutation of (0 brightness, 1 contrast, 2 saturation, 3 hue); `factors`: the four parameters
    (brightness / contrast / saturation in [0.8, 1.2], hue in [-0.1, 0.1], datasets/mono_dataset.py:73-76)."""
    for k in order:
        img = JITTER_OPS[k](img, factors[k])
    return img


def preprocess_item(native, height, width, num_scales=4, flip=False, jitter=None):
    """datasets/mono_dataset.py:92-118 + the flip of get_color for ONE frame: native (Hn, Wn, 3) uint8 ->
    {("color", s): CHW float32, ("color_aug", s): CHW float32} for s in 0..num_scales-1.  `jitter` = (order, factors) or None."""
    img = native[:, ::-1] if flip else native
    out = {}
    for s in range(num_scales):
        img = resize_lanczos(img, height >> s, width >> s)            # scale s is resized from scale s-1 (-1 = native)
        out[("color", s)] = to_tensor(img)
        out[("color_aug", s)] = to_tensor(color_jitter(img, *jitter)) if jitter is not None else out[("color", s)]
    return out
