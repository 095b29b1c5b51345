"""Golden vectors of the per-item data step, produced by the INSTALLED Pillow (the library the
reference's datasets/mono_dataset.py calls; torchvision is not installed, so its PIL glue -- ImageEnhance for brightness /
contrast / saturation, the H-channel shift for hue -- is spelled out with Pillow calls below).

    python tests/golden/make_golden_data.py        -> tests/golden/data_pillow.npz

Inputs are regenerated from seeds by tests/helpers (`data_case_image`); the file holds only expected outputs.
`np.uint8(hue_factor * 255)` wraps modulo 256 in the NumPy 1.x the reference ran on (NumPy 2 raises for negatives);
the wrap is written explicitly here.
"""
import os
import sys

import numpy as np
from PIL import Image, ImageEnhance

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from helpers import DATA_CASES, data_case_image  # noqa: E402

LANCZOS = getattr(Image, "Resampling", Image).LANCZOS


def pil_hue(img, hue_factor):
    h, s, v = img.convert("HSV").split()
    np_h = (np.array(h, dtype=np.int32) + (int(hue_factor * 255) & 0xFF)) & 0xFF
    return Image.merge("HSV", (Image.fromarray(np_h.astype(np.uint8), "L"), s, v)).convert("RGB")


PIL_OPS = (lambda im, f: ImageEnhance.Brightness(im).enhance(f), lambda im, f: ImageEnhance.Contrast(im).enhance(f),
           lambda im, f: ImageEnhance.Color(im).enhance(f), pil_hue)


def main():
    out = {}
    for name, (hn, wn, h, w, scales, flip, order, factors, seed) in DATA_CASES.items():
        img = Image.fromarray(data_case_image(hn, wn, seed))
        if flip:
            img = img.transpose(Image.FLIP_LEFT_RIGHT)
        for s in range(scales):
            img = img.resize((w >> s, h >> s), LANCZOS)
            out["%s/color%d" % (name, s)] = np.asarray(img).copy()
            aug = img
            if order is not None:
                for k in order:
                    aug = PIL_OPS[k](aug, factors[k])
            out["%s/aug%d" % (name, s)] = np.asarray(aug).copy()
    # colour conversions on a strided sample of the RGB cube (the full cube is checked live when Pillow is importable)
    g = np.arange(0, 256, 5, dtype=np.uint8)
    cube = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, len(g), 3)
    out["cube/hsv"] = np.asarray(Image.fromarray(cube).convert("HSV")).copy()
    out["cube/rgb_from_hsv"] = np.asarray(Image.fromarray(cube, "HSV").convert("RGB")).copy()
    out["cube/l"] = np.asarray(Image.fromarray(cube).convert("L")).copy()
    np.savez_compressed(os.path.join(HERE, "data_pillow.npz"), **out)
    print("wrote", len(out), "arrays,", os.path.getsize(os.path.join(HERE, "data_pillow.npz")), "bytes")


if __name__ == "__main__":
    main()
