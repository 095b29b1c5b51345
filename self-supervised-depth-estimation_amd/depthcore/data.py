"""Per-item data step on the device: datasets/mono_dataset.py:92-118 `preprocess` and the image part of
:139-211 `__getitem__`, for a whole batch of decoded frames at once.

The reference runs this in `num_workers` CPU processes per item with Pillow / torchvision; here the decoded frames are
uploaded once as uint8 HWC and everything downstream (flip, Lanczos pyramid, ColorJitter, ToTensor) happens in HBM through
libdepthcore's `dc_data_*` entry points, byte-exact with the CPU pipeline (tests/test_data_gpu.py, oracle/data_ref.py).
Random draws stay on the host (`sample_item`), in the reference's order.  There is no CPU path.
"""
import random

import numpy as np
import torch

from . import _lib
from ._lib import check, stream

BRIGHTNESS, CONTRAST, SATURATION, HUE = 0, 1, 2, 3
NO_OP = -1


def resample_table(in_size, out_size):
    """Pillow's 8-bit Lanczos coefficient table of one axis (host, exact): bounds (out, 2) int32, kk (out, ksize) int32."""
    L = _lib.lib()
    ksize = L.dc_resample_ksize(int(in_size), int(out_size))
    if ksize <= 0:
        raise _lib.DepthcoreError("bad resample sizes %r -> %r" % (in_size, out_size))
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    check(L.dc_resample_table(int(in_size), int(out_size), bounds.ctypes.data, kk.ctypes.data), "dc_resample_table")
    return bounds, kk


def sample_item(is_train=True, rng=random, brightness=(0.8, 1.2), contrast=(0.8, 1.2), saturation=(0.8, 1.2), hue=(-0.1, 0.1)):
    """The random draws of one item in the reference's order (datasets/mono_dataset.py:139-140, :186-188):
    do_color_aug, do_flip, then ColorJitter.get_params (the four factors, then the order of the four transforms).
    -> (do_flip, jitter) with jitter = None or (order, factors) indexed by BRIGHTNESS..HUE."""
    do_color_aug = is_train and rng.random() > 0.5
    do_flip = is_train and rng.random() > 0.5
    jitter = None
    if do_color_aug:
        factors = [rng.uniform(*brightness), rng.uniform(*contrast), rng.uniform(*saturation), rng.uniform(*hue)]
        order = [0, 1, 2, 3]
        rng.shuffle(order)
        jitter = (tuple(order), tuple(factors))
    return bool(do_flip), jitter


def hue_shift(hue_factor):
    """torchvision functional_pil.adjust_hue: `np.uint8(hue_factor * 255)` (truncation toward zero, wrap to 8 bits)."""
    return int(hue_factor * 255) & 0xFF


class GpuPreprocessor:
    """inputs-dict builder: native frames (uint8 HWC on the device) -> ("color" / "color_aug", frame, scale) float tensors."""

    def __init__(self, height, width, num_scales=4, frame_idxs=(0, -1, 1), device="cuda"):
        self.height, self.width, self.num_scales = int(height), int(width), int(num_scales)
        self.frame_idxs = tuple(frame_idxs)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.DepthcoreError("GpuPreprocessor needs a GPU device; there is no CPU path")
        self._tables = {}

    def _table(self, in_size, out_size):
        key = (in_size, out_size)
        if key not in self._tables:
            bounds, kk = resample_table(in_size, out_size)
            self._tables[key] = (torch.from_numpy(bounds).to(self.device), torch.from_numpy(kk).to(self.device), kk.shape[1])
        return self._tables[key]

    def _resize(self, img, out_h, out_w, flip):
        """(n, H, W, 3) uint8 -> (n, out_h, out_w, 3): width pass (carrying the mirror), then height pass."""
        L = _lib.lib()
        n, H, W, _ = img.shape
        st = stream(img)
        fp = flip.data_ptr() if flip is not None else None
        if W != out_w:
            bounds, kk, ks = self._table(W, out_w)
            dst = torch.empty((n, H, out_w, 3), dtype=torch.uint8, device=img.device)
            check(L.dc_data_resize_axis(img.data_ptr(), dst.data_ptr(), n, H, W, out_w, 1, bounds.data_ptr(), kk.data_ptr(), ks, fp, st),
                  "dc_data_resize_axis")
            img = dst
        elif flip is not None:
            dst = torch.empty_like(img)
            check(L.dc_data_flip(img.data_ptr(), dst.data_ptr(), n, H, W, fp, st), "dc_data_flip")
            img = dst
        if H != out_h:
            bounds, kk, ks = self._table(H, out_h)
            dst = torch.empty((n, out_h, out_w, 3), dtype=torch.uint8, device=img.device)
            check(L.dc_data_resize_axis(img.data_ptr(), dst.data_ptr(), n, H, out_w, out_h, 0, bounds.data_ptr(), kk.data_ptr(), ks, None, st),
                  "dc_data_resize_axis")
            img = dst
        return img

    def _to_tensor(self, img):
        L = _lib.lib()
        n, H, W, _ = img.shape
        out = torch.empty((n, 3, H, W), dtype=torch.float32, device=img.device)
        check(L.dc_data_to_tensor(img.data_ptr(), out.data_ptr(), n, H * W, stream(img)), "dc_data_to_tensor")
        return out

    def __call__(self, native, flips=None, jitters=None):
        """native: (F, B, Hn, Wn, 3) uint8 device tensor, frame-major in the order of `frame_idxs` (the decoded
        ("color", f, -1) images of B items); flips: B booleans; jitters: B entries, None or (order, factors).
        One item's flip / jitter applies to all of its frames and scales (mono_dataset.py:94-97).
        -> {("color", f, s), ("color_aug", f, s)}: (B, 3, h, w) float32."""
        if native.dtype != torch.uint8 or native.dim() != 5 or native.shape[-1] != 3 or not native.is_cuda or not native.is_contiguous():
            raise _lib.DepthcoreError("native frames must be a contiguous (F, B, H, W, 3) uint8 device tensor")
        Fn, B, Hn, Wn, _ = native.shape
        if Fn != len(self.frame_idxs):
            raise _lib.DepthcoreError("%d frame slabs for frame_idxs %r" % (Fn, self.frame_idxs))
        L = _lib.lib()
        n = Fn * B
        flip = None
        if flips is not None and any(flips):
            flip = torch.tensor([1 if flips[b] else 0 for _ in range(Fn) for b in range(B)], dtype=torch.uint8).to(native.device)
        steps = params = sums = None
        if jitters is not None and any(j is not None for j in jitters):
            st_h = np.full((n, 4), NO_OP, np.int32)
            pr_h = np.zeros((n, 4), np.float32)
            for b, j in enumerate(jitters):
                if j is None:
                    continue
                order, factors = j
                for k, op in enumerate(order):
                    val = float(hue_shift(factors[op])) if op == HUE else factors[op]
                    for f in range(Fn):
                        st_h[f * B + b, k] = op
                        pr_h[f * B + b, k] = val
            steps = torch.from_numpy(st_h).to(native.device)
            params = torch.from_numpy(pr_h).to(native.device)
            sums = torch.empty(n, dtype=torch.int64, device=native.device)
        out = {}
        img = native.view(n, Hn, Wn, 3)
        for s in range(self.num_scales):
            h, w = self.height // (2 ** s), self.width // (2 ** s)
            img = self._resize(img, h, w, flip if s == 0 else None)
            color = torch.empty((n, 3, h, w), dtype=torch.float32, device=img.device)
            aug = torch.empty_like(color) if steps is not None else None
            check(L.dc_data_jitter_to_tensor(img.data_ptr(), color.data_ptr(), aug.data_ptr() if aug is not None else None, n, h * w,
                                             steps.data_ptr() if aug is not None else None, params.data_ptr() if aug is not None else None,
                                             sums.data_ptr() if aug is not None else None, stream(img)), "dc_data_jitter_to_tensor")
            color = color.view(Fn, B, 3, h, w)
            color_aug = aug.view(Fn, B, 3, h, w) if aug is not None else color
            for i, f in enumerate(self.frame_idxs):
                out[("color", f, s)] = color[i]
                out[("color_aug", f, s)] = color_aug[i]
            if s == 0:
                # the photometric kernels gather from pixel-interleaved RGBx: written here, once per item, from the same uint8
                # level (same division by 255 -> the same values as ("color", f, 0)) instead of repacked in every training step
                packed = torch.empty((n, h, w, 4), dtype=torch.float32, device=img.device)
                check(L.dc_data_to_rgbx(img.data_ptr(), packed.data_ptr(), n, h * w, stream(img)), "dc_data_to_rgbx")
                packed = packed.view(Fn, B, h, w, 4)
                for i, f in enumerate(self.frame_idxs):
                    out[("color_packed", f, 0)] = packed[i]
        return out

    def intrinsics(self, K, batch):
        """("K", s), ("inv_K", s) for a normalised 4x4 intrinsics matrix (mono_dataset.py:170-180), repeated over the batch."""
        out = {}
        for s in range(self.num_scales):
            Ks = np.array(K, dtype=np.float32).copy()
            Ks[0, :] *= self.width // (2 ** s)
            Ks[1, :] *= self.height // (2 ** s)
            inv = np.linalg.pinv(Ks)
            out[("K", s)] = torch.from_numpy(Ks).to(self.device).unsqueeze(0).repeat(batch, 1, 1).contiguous()
            out[("inv_K", s)] = torch.from_numpy(inv).to(self.device).unsqueeze(0).repeat(batch, 1, 1).contiguous()
        return out
