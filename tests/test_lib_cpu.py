"""No-GPU checks of the C-ABI boundary: the library builds/loads, exports every symbol that
include/depthcore.h declares, the ctypes struct mirrors the C struct, and the ops refuse CPU tensors."""
import ctypes
import os
import re
import subprocess

import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(REPO, "include", "depthcore.h")


def declared_symbols():
    src = open(HDR).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dc_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from depthcore import _lib
    L = _lib.lib()
    names = declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(L, n), "libdepthcore.so does not export %s" % n
        assert n in _lib.EXPORTS, "ctypes binding table lacks %s" % n
    assert _lib.MISSING == []
    assert set(_lib.EXPORTS) == set(names)
    assert L.dc_arch() == b"gfx950"


def test_photo_desc_layout_matches_c(tmp_path):
    """sizeof / a few offsetof of dc_photo_desc as seen by gcc vs the ctypes mirror."""
    from depthcore import _lib
    c = tmp_path / "sz.c"
    c.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "depthcore.h"\nint main(){printf("%zu %zu %zu %zu %zu\\n",'
                 'sizeof(dc_photo_desc), offsetof(dc_photo_desc,target), offsetof(dc_photo_desc,rng_seed),'
                 'offsetof(dc_photo_desc,sample), offsetof(dc_photo_desc,workspace_bytes));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(REPO, "include"), str(c), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    D = _lib.PhotoDesc
    want = [ctypes.sizeof(D), D.target.offset, D.rng_seed.offset, D.sample.offset, D.workspace_bytes.offset]
    assert got == want


def test_workspace_and_bytes_queries_without_gpu():
    from depthcore import _lib
    L = _lib.lib()
    d = _lib.PhotoDesc()
    d.B, d.H, d.W, d.num_scales = 12, 192, 640, 4
    N = 12 * 192 * 640
    prev = L.dc_set_photo_full(0)
    try:
        ws = L.dc_photo_workspace(ctypes.byref(d))
        # the round-4 split: idl 2N + three RGBx copies 12N + d(upsampled disp) 4N + the forward's d(loss)/d(source coords), 4 floats x 4 scales
        assert ws > (17 + 16) * N * 4 and ws < (20 + 16) * N * 4
        L.dc_set_photo_full(1)
        ws_full = L.dc_photo_workspace(ctypes.byref(d))
        # the forward that goes all the way (default): no d(loss)/d(source coords) buffers
        assert ws_full > 17 * N * 4 and ws_full < 20 * N * 4
    finally:
        L.dc_set_photo_full(prev)
    d.flags = _lib.OPT_NO_GRAD                        # evaluation: no gradient emission, the smaller workspace
    ws_eval = L.dc_photo_workspace(ctypes.byref(d))
    assert ws_eval > 17 * N * 4 and ws_eval < 20 * N * 4
    d.flags = 0
    fwd = L.dc_photo_algorithmic_bytes(ctypes.byref(d), 0)
    bwd = L.dc_photo_algorithmic_bytes(ctypes.byref(d), 1)
    assert abs(fwd / N - 165.25) < 1e-6 and abs(bwd / N - 170.5625) < 1e-6
    assert L.dc_project3d_bwd_workspace(2, 64, 96) > 0 and L.dc_smooth_workspace(2, 64, 96) > 0


def test_ops_refuse_cpu_tensors():
    from depthcore import ops, _lib
    with pytest.raises(_lib.DepthcoreError):
        ops.disp_to_depth(torch.rand(1, 1, 4, 4), 0.1, 100.0)
    with pytest.raises(_lib.DepthcoreError):
        ops.ssim(torch.rand(1, 3, 8, 8), torch.rand(1, 3, 8, 8))


def test_facade_names_and_state_dict_layout():
    """The reference-shaped module API: names, parameter counts and state_dict keys."""
    import numpy as np
    import layers
    import networks
    for n in ("disp_to_depth", "transformation_from_parameters", "get_translation_matrix", "rot_from_axisangle",
              "ConvBlock", "Conv3x3", "BackprojectDepth", "Project3D", "upsample", "get_smooth_loss", "SSIM",
              "compute_depth_errors", "torch", "nn", "F", "np"):
        assert hasattr(layers, n), n
    enc = networks.ResnetEncoder(18, False)
    assert sum(p.numel() for p in enc.parameters()) == 11689512
    assert list(enc.num_ch_enc) == [64, 64, 128, 256, 512]
    k = list(enc.state_dict().keys())
    assert k[0] == "encoder.conv1.weight" and "encoder.layer4.1.bn2.running_var" in k and k[-1] == "encoder.fc.bias"
    assert networks.ResnetEncoder(18, False, 2).encoder.conv1.weight.shape == (64, 6, 7, 7)
    e50 = networks.ResnetEncoder(50, False)
    assert list(e50.num_ch_enc) == [64, 256, 512, 1024, 2048]
    dec = networks.DepthDecoder(enc.num_ch_enc)
    assert sum(p.numel() for p in dec.parameters()) == 3152724 and len(dec.state_dict()) == 28
    assert list(dec.state_dict())[0] == "decoder.0.conv.conv.weight" and list(dec.state_dict())[-1] == "decoder.13.conv.bias"
    pose = networks.PoseDecoder(enc.num_ch_enc, 1, 2)
    assert sum(p.numel() for p in pose.parameters()) == 1314572
    assert list(pose.state_dict()) == ["net.%d.%s" % (i, w) for i in range(4) for w in ("weight", "bias")]
    bp = layers.BackprojectDepth(1, 192, 640)
    pc = bp.pix_coords[0].numpy()
    assert np.array_equal(pc[:, 639:642], np.array([[639, 0, 1], [0, 1, 1], [1, 1, 1]], np.float32))


def test_wgrad_lanes_bookkeeping_without_gpu():
    """ops.WgradLanes host logic: uses are counted per parameter by the forwards, a lane is only taken for a parameter used
    exactly once whose .grad is still None (autograd sums the gradients of a multiply-used parameter on the backward's own
    stream), CPU tensors never switch streams, and leaving active() clears the step's bookkeeping."""
    import torch
    from depthcore import ops
    L = ops.WgradLanes
    p, q = torch.nn.Parameter(torch.zeros(3)), torch.nn.Parameter(torch.zeros(3))
    L.join()
    assert not L._on and L._uses == {}
    with L.active():
        assert L._on
        L.count_use(p)
        L.count_use(q)
        L.count_use(q)
        assert L._uses[id(p)] == 1 and L._uses[id(q)] == 2
        with L.lane(p, torch.zeros(2)):          # CPU tensor: no stream switch, nothing recorded
            pass
        assert not L._used
    assert not L._on and L._uses == {} and not L._used
    with L.active(False):
        assert not L._on
