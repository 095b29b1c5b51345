"""Synthetic KITTI-shaped batches with the reference's dict schema
(datasets/mono_dataset.py:122-183, datasets/kitti_dataset.py:25-28), generated on the device."""
import numpy as np
import torch
import torch.nn.functional as F

KITTI_K = np.array([[0.58, 0, 0.5, 0], [0, 1.92, 0.5, 0], [0, 0, 1, 0], [0, 0, 0, 1]], dtype=np.float32)


def synthetic_batch(batch, height, width, device, num_scales=4, frame_ids=(0, -1, 1), seed=0, smooth=True, packed=False):
    """U(0,1) images (optionally 5x5 box-smoothed so SSIM is not saturated), pyramid by 2x2 means,
    KITTI intrinsics scaled per pyramid level, inv_K = pinv(K).  `packed`: also what the device data step emits next to
    ("color", f, 0) -- ("color_packed", f, 0), the pixel-interleaved RGBx copy the photometric kernels gather from
    (depthcore.data.GpuPreprocessor, dc_data_to_rgbx) -- for the three frames of the loss."""
    g = torch.Generator(device=device).manual_seed(seed)
    inputs = {}
    for f in frame_ids:
        base = torch.rand(batch, 3, height, width, device=device, generator=g)
        if smooth:
            base = F.avg_pool2d(F.pad(base, (2, 2, 2, 2), mode="reflect"), 5, 1)
        for s in range(num_scales):
            img = base if s == 0 else F.avg_pool2d(base, 2 ** s)
            inputs[("color", f, s)] = img.contiguous()
            inputs[("color_aug", f, s)] = inputs[("color", f, s)]
        if packed and f in (0, -1, 1):
            from . import ops
            inputs[("color_packed", f, 0)] = ops.pack_rgbx(inputs[("color", f, 0)])
    for s in range(num_scales):
        K = KITTI_K.copy()
        K[0, :] *= width // (2 ** s)
        K[1, :] *= height // (2 ** s)
        inv_K = np.linalg.pinv(K)
        inputs[("K", s)] = torch.from_numpy(K).to(device).unsqueeze(0).repeat(batch, 1, 1).contiguous()
        inputs[("inv_K", s)] = torch.from_numpy(inv_K).to(device).unsqueeze(0).repeat(batch, 1, 1).contiguous()
    return inputs


def synthetic_sequence_batch(len_sequence, height, width, device, num_scales=4, frame_ids=(0, -1, 1), seed=0):
    """One sequence of `len_sequence` frames in the schema of datasets/kitti_dataset_seq.py:110-140 (batch size 1):
    ("color", f, s, j), ("K", s, j), ("inv_K", s, j)."""
    flat = synthetic_batch(len_sequence, height, width, device, num_scales, frame_ids, seed)
    out = {}
    for k, v in flat.items():
        if k[0] == "color_aug":
            continue
        for j in range(len_sequence):
            out[k + (j,)] = v[j:j + 1].contiguous()
    return out
