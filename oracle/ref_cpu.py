"""CPU restatement of the reference's geometry + photometric-loss hot path.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Plain torch fp32 on the CPU,
written from the formulas rather than by calling the same high-level ATen ops
the reference calls (explicit gathers instead of `F.grid_sample`, explicit
shifted sums instead of `AvgPool2d`, ...), so that it is an independent
statement of the semantics in SURVEY.md Appendix B.  Every function cites the
reference lines it follows.  Autograd through these functions is the gradient
oracle for the HIP backward kernels.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------
# a6  disp_to_depth                                    reference layers.py:16-25
# --------------------------------------------------------------------------
def disp_to_depth(disp, min_depth, max_depth):
    lo = 1.0 / max_depth
    hi = 1.0 / min_depth
    scaled = lo + (hi - lo) * disp
    return scaled, 1.0 / scaled


# --------------------------------------------------------------------------
# a5  pose parameters -> 4x4                           reference layers.py:28-103
# --------------------------------------------------------------------------
def rot_from_axisangle(vec):
    """vec (B,1,3) -> (B,4,4) Rodrigues rotation.  layers.py:64-103."""
    B = vec.shape[0]
    v = vec.reshape(B, 3)
    angle = torch.sqrt((v * v).sum(1, keepdim=True))          # torch.norm(vec,2,2,True)
    axis = v / (angle + 1e-7)
    ca, sa = torch.cos(angle), torch.sin(angle)
    C = 1.0 - ca
    x, y, z = axis[:, 0:1], axis[:, 1:2], axis[:, 2:3]
    rows = [
        torch.cat([x * (x * C) + ca, x * (y * C) - z * sa, z * (x * C) + y * sa], 1),
        torch.cat([x * (y * C) + z * sa, y * (y * C) + ca, y * (z * C) - x * sa], 1),
        torch.cat([z * (x * C) - y * sa, y * (z * C) + x * sa, z * (z * C) + ca], 1),
    ]
    R3 = torch.stack(rows, 1)                                   # (B,3,3)
    R = torch.cat([torch.cat([R3, torch.zeros(B, 3, 1, dtype=vec.dtype)], 2),
                   torch.tensor([0.0, 0.0, 0.0, 1.0], dtype=vec.dtype).expand(B, 1, 4)], 1)
    return R


def get_translation_matrix(t):
    """t (B,1,3) or (B,3) -> (B,4,4).  layers.py:48-61."""
    B = t.shape[0]
    tv = t.reshape(B, 3, 1)
    top = torch.cat([torch.eye(3, dtype=t.dtype).expand(B, 3, 3), tv], 2)
    bot = torch.tensor([0.0, 0.0, 0.0, 1.0], dtype=t.dtype).expand(B, 1, 4)
    return torch.cat([top, bot], 1)


def transformation_from_parameters(axisangle, translation, invert=False):
    """layers.py:28-45: M = T @ R, or R^T @ T(-t) when invert."""
    R = rot_from_axisangle(axisangle)
    t = translation
    if invert:
        R = R.transpose(1, 2)
        t = -t
    T = get_translation_matrix(t)
    return torch.matmul(R, T) if invert else torch.matmul(T, R)


# --------------------------------------------------------------------------
# a7  BackprojectDepth                                reference layers.py:139-168
# --------------------------------------------------------------------------
def pix_coords(batch_size, height, width):
    """(B,3,H*W) homogeneous pixel grid [x; y; 1], x = i mod W, y = i div W.

    layers.py:150-161 builds it from np.meshgrid(range(W), range(H), 'xy');
    the values are exact small integers stored in fp32.
    """
    i = np.arange(height * width, dtype=np.int64)
    x = (i % width).astype(np.float32)
    y = (i // width).astype(np.float32)
    one = np.ones_like(x)
    pc = torch.from_numpy(np.stack([x, y, one], 0))
    return pc.unsqueeze(0).repeat(batch_size, 1, 1)


def backproject(depth, inv_K):
    """depth (B,1,H,W), inv_K (B,4,4) -> cam points (B,4,HW).  layers.py:163-168."""
    B, _, H, W = depth.shape
    pc = pix_coords(B, H, W).to(depth.dtype)
    rays = torch.matmul(inv_K[:, :3, :3], pc)
    cam = depth.reshape(B, 1, -1) * rays
    return torch.cat([cam, torch.ones(B, 1, H * W, dtype=depth.dtype)], 1)


# --------------------------------------------------------------------------
# a8  Project3D                                       reference layers.py:171-193
# --------------------------------------------------------------------------
def project3d(points, K, T, height, width, eps=1e-7):
    """points (B,4,HW) -> sampling grid (B,H,W,2) in [-1,1] 'W-1' convention."""
    B = points.shape[0]
    P = torch.matmul(K, T)[:, :3, :]
    cam = torch.matmul(P, points)
    z = cam[:, 2:3, :] + eps
    u = (cam[:, 0:1, :] / z) / (width - 1)
    v = (cam[:, 1:2, :] / z) / (height - 1)
    uv = torch.cat([u, v], 1).reshape(B, 2, height, width).permute(0, 2, 3, 1)
    return (uv - 0.5) * 2


# --------------------------------------------------------------------------
# a9  grid_sample(bilinear, border, align_corners=False)    trainer.py:508-511
# --------------------------------------------------------------------------
def _unnormalize_clip(g, size, align_corners):
    if align_corners:
        x = (g + 1) / 2 * (size - 1)
    else:
        x = ((g + 1) * size - 1) / 2
    # ATen clip_coordinates_set_grad: gradient is exactly zero where the
    # unclamped coordinate is <= 0 or >= size-1.
    inside = (x > 0) & (x < size - 1)
    xc = x.clamp(0, size - 1)
    return torch.where(inside, x, xc.detach())


def grid_sample_border(img, grid, align_corners=False):
    """img (B,C,H,W), grid (B,Ho,Wo,2) -> (B,C,Ho,Wo)."""
    B, C, H, W = img.shape
    Ho, Wo = grid.shape[1], grid.shape[2]
    x = _unnormalize_clip(grid[..., 0], W, align_corners)
    y = _unnormalize_clip(grid[..., 1], H, align_corners)
    x0 = torch.floor(x.detach())
    y0 = torch.floor(y.detach())
    wx1 = x - x0
    wy1 = y - y0
    wx0 = 1 - wx1
    wy0 = 1 - wy1
    x0 = x0.long()
    y0 = y0.long()
    flat = img.reshape(B, C, H * W)

    def tap(yy, xx):
        ok = ((xx >= 0) & (xx < W) & (yy >= 0) & (yy < H)).to(img.dtype)
        idx = (yy.clamp(0, H - 1) * W + xx.clamp(0, W - 1)).reshape(B, 1, Ho * Wo)
        val = torch.gather(flat, 2, idx.expand(B, C, Ho * Wo)).reshape(B, C, Ho, Wo)
        return val * ok.unsqueeze(1)

    out = (tap(y0, x0) * (wy0 * wx0).unsqueeze(1)
           + tap(y0, x0 + 1) * (wy0 * wx1).unsqueeze(1)
           + tap(y0 + 1, x0) * (wy1 * wx0).unsqueeze(1)
           + tap(y0 + 1, x0 + 1) * (wy1 * wx1).unsqueeze(1))
    return out


# --------------------------------------------------------------------------
# a10 F.interpolate(bilinear, align_corners=False)          trainer.py:474-475
# --------------------------------------------------------------------------
def _lin_taps(n_in, n_out, dtype):
    scale = n_in / n_out
    dst = torch.arange(n_out, dtype=dtype)
    src = ((dst + 0.5) * scale - 0.5).clamp(min=0)
    i0 = torch.floor(src).long().clamp(max=n_in - 1)
    i1 = (i0 + 1).clamp(max=n_in - 1)
    w1 = src - i0.to(dtype)
    return i0, i1, w1


def upsample_bilinear(x, height, width):
    """(B,C,h,w) -> (B,C,height,width); same-size call is the identity."""
    B, C, h, w = x.shape
    if (h, w) == (height, width):
        return x
    y0, y1, wy = _lin_taps(h, height, x.dtype)
    x0, x1, wx = _lin_taps(w, width, x.dtype)
    top = x[:, :, y0, :]
    bot = x[:, :, y1, :]
    wy = wy.view(1, 1, -1, 1)
    wx = wx.view(1, 1, 1, -1)
    # ATen upsample_bilinear2d: w0*(wx0*a + wx1*b) + w1*(wx0*c + wx1*d)
    t = top[:, :, :, x0] * (1 - wx) + top[:, :, :, x1] * wx
    b = bot[:, :, :, x0] * (1 - wx) + bot[:, :, :, x1] * wx
    return t * (1 - wy) + b * wy


def upsample_nearest2(x):
    """layers.py:196-199: out[y,x] = in[y//2, x//2]."""
    return x.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3)


# --------------------------------------------------------------------------
# a11 SSIM                                            reference layers.py:218-248
# --------------------------------------------------------------------------
def _reflect_pad1(x):
    x = torch.cat([x[:, :, 1:2, :], x, x[:, :, -2:-1, :]], 2)
    x = torch.cat([x[:, :, :, 1:2], x, x[:, :, :, -2:-1]], 3)
    return x


def _box3(xp):
    """3x3 mean over a (B,C,H+2,W+2) padded tensor -> (B,C,H,W)."""
    H, W = xp.shape[2] - 2, xp.shape[3] - 2
    acc = 0
    for dy in range(3):
        for dx in range(3):
            acc = acc + xp[:, :, dy:dy + H, dx:dx + W]
    return acc / 9.0


def ssim(x, y):
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    xp, yp = _reflect_pad1(x), _reflect_pad1(y)
    mu_x, mu_y = _box3(xp), _box3(yp)
    sig_x = _box3(xp * xp) - mu_x * mu_x
    sig_y = _box3(yp * yp) - mu_y * mu_y
    sig_xy = _box3(xp * yp) - mu_x * mu_y
    n = (2 * mu_x * mu_y + C1) * (2 * sig_xy + C2)
    d = (mu_x * mu_x + mu_y * mu_y + C1) * (sig_x + sig_y + C2)
    return torch.clamp((1 - n / d) / 2, 0, 1)


# --------------------------------------------------------------------------
# a12 compute_reprojection_loss                             trainer.py:517-529
# --------------------------------------------------------------------------
def reprojection_loss(pred, target, no_ssim=False):
    l1 = (target - pred).abs().mean(1, keepdim=True)
    if no_ssim:
        return l1
    return 0.85 * ssim(pred, target).mean(1, keepdim=True) + 0.15 * l1


# --------------------------------------------------------------------------
# a13 get_smooth_loss                                 reference layers.py:202-215
# --------------------------------------------------------------------------
def smooth_loss(disp, img):
    dx = (disp[:, :, :, :-1] - disp[:, :, :, 1:]).abs()
    dy = (disp[:, :, :-1, :] - disp[:, :, 1:, :]).abs()
    ix = (img[:, :, :, :-1] - img[:, :, :, 1:]).abs().mean(1, keepdim=True)
    iy = (img[:, :, :-1, :] - img[:, :, 1:, :]).abs().mean(1, keepdim=True)
    return (dx * torch.exp(-ix)).mean() + (dy * torch.exp(-iy)).mean()


# --------------------------------------------------------------------------
# a14 generate_images_pred                                  trainer.py:465-515
# --------------------------------------------------------------------------
class Opt:
    """Hot-path defaults of options.py:100-213 (only the fields the path reads)."""
    height = 192
    width = 640
    scales = (0, 1, 2, 3)
    min_depth = 0.1
    max_depth = 100.0
    disparity_smoothness = 1e-3
    frame_ids = (0, -1, 1)
    v1_multiscale = False
    avg_reprojection = False
    disable_automasking = False
    predictive_mask = False
    no_ssim = False
    align_corners = False      # installed-torch default of F.grid_sample
    pose_model_type = "separate_resnet"

    def __init__(self, **kw):
        for k, v in kw.items():
            setattr(self, k, v)


def generate_images_pred(inputs, outputs, opt):
    for s in opt.scales:
        disp = outputs[("disp", s)]
        if opt.v1_multiscale:
            src = s
            H, W = disp.shape[2], disp.shape[3]
        else:
            src = 0
            H, W = opt.height, opt.width
            disp = upsample_bilinear(disp, H, W)
        _, depth = disp_to_depth(disp, opt.min_depth, opt.max_depth)
        outputs[("depth", 0, s)] = depth
        for f in (-1, 1):
            T = outputs[("cam_T_cam", 0, f)]
            if opt.pose_model_type == "posecnn":
                # trainer.py:490-499: the translation is rescaled by this scale's mean inverse depth and T rebuilt
                mean_inv_depth = (1 / depth).mean(3, True).mean(2, True)
                T = transformation_from_parameters(
                    outputs[("axisangle", 0, f)][:, 0], outputs[("translation", 0, f)][:, 0] * mean_inv_depth[:, 0], f < 0)
            cam = backproject(depth, inputs[("inv_K", src)])
            grid = project3d(cam, inputs[("K", src)], T, H, W)
            outputs[("sample", f, s)] = grid
            outputs[("color", f, s)] = grid_sample_border(
                inputs[("color", f, src)], grid, opt.align_corners)
            if not opt.disable_automasking:
                outputs[("color_identity", f, s)] = inputs[("color", f, src)]


# --------------------------------------------------------------------------
# a15 compute_losses                                        trainer.py:531-622
# --------------------------------------------------------------------------
def compute_losses(inputs, outputs, opt, noise):
    """`noise[s]` (B,2,H,W) replaces the CPU `torch.randn` of trainer.py:594-595."""
    losses = {}
    total = 0
    for s in opt.scales:
        src = s if opt.v1_multiscale else 0
        disp = outputs[("disp", s)]
        color = inputs[("color", 0, s)]
        target = inputs[("color", 0, src)]
        reproj = torch.cat([reprojection_loss(outputs[("color", f, s)], target, opt.no_ssim)
                            for f in (-1, 1)], 1)
        loss = 0
        if not opt.disable_automasking:
            ident = torch.cat([reprojection_loss(inputs[("color", f, src)], target, opt.no_ssim)
                               for f in (-1, 1)], 1)
            if opt.avg_reprojection:
                ident = ident.mean(1, keepdim=True)
        elif opt.predictive_mask:
            # trainer.py:571-584: per-frame mask from a second decoder, pushed towards 1 by a BCE term (log clamped at -100)
            mask = outputs["predictive_mask"][("disp", s)]
            if not opt.v1_multiscale:
                mask = upsample_bilinear(mask, opt.height, opt.width)
            reproj = reproj * mask
            loss = loss + 0.2 * (-torch.clamp(torch.log(mask), min=-100.0)).mean()
        if opt.avg_reprojection:
            reproj = reproj.mean(1, keepdim=True)
        if not opt.disable_automasking:
            ident = ident + noise[s] * 0.00001
            combined = torch.cat([ident, reproj], 1)
        else:
            combined = reproj
        if combined.shape[1] == 1:
            to_opt = combined
        else:
            to_opt, idx = torch.min(combined, dim=1)
        if not opt.disable_automasking:
            outputs["identity_selection/{}".format(s)] = (idx > ident.shape[1] - 1).float()
        loss = loss + to_opt.mean()
        mean_disp = disp.mean(2, True).mean(3, True)
        norm_disp = disp / (mean_disp + 1e-7)
        loss = loss + opt.disparity_smoothness * smooth_loss(norm_disp, color) / (2 ** s)
        total = total + loss
        losses["loss/{}".format(s)] = loss
    losses["loss"] = total / len(opt.scales)
    return losses


# --------------------------------------------------------------------------
# a3  Conv3x3 / ConvBlock                             reference layers.py:106-136
# --------------------------------------------------------------------------
def conv3x3_reflect(x, weight, bias):
    return F.conv2d(_reflect_pad1(x), weight, bias)


def conv_block(x, weight, bias):
    return F.elu(conv3x3_reflect(x, weight, bias))


# --------------------------------------------------------------------------
# a2  DepthDecoder                       reference networks/depth_decoder.py:17-67
# --------------------------------------------------------------------------
def depth_decoder_layout(num_ch_enc, scales=(0, 1, 2, 3), num_output_channels=1, use_skips=True):
    """Module order of `self.decoder = nn.ModuleList(convs.values())`.

    Returns a list of (index, name, cin, cout) in state_dict order: ten
    ConvBlocks (`decoder.{i}.conv.conv.*`) then the dispconvs (`decoder.{i}.conv.*`).
    """
    dec = [16, 32, 64, 128, 256]
    out = []
    for i in range(4, -1, -1):
        cin = int(num_ch_enc[-1]) if i == 4 else dec[i + 1]
        out.append((("upconv", i, 0), cin, dec[i]))
        cin = dec[i] + (int(num_ch_enc[i - 1]) if (use_skips and i > 0) else 0)
        out.append((("upconv", i, 1), cin, dec[i]))
    for s in scales:
        out.append((("dispconv", s), dec[s], num_output_channels))
    return [(k, name, cin, cout) for k, (name, cin, cout) in enumerate(out)]


def depth_decoder_forward(state, features, num_ch_enc, scales=(0, 1, 2, 3), use_skips=True,
                          pre_disp=False):
    lay = {name: idx for idx, name, _, _ in depth_decoder_layout(num_ch_enc, scales, 1, use_skips)}

    def cb(name, x):
        i = lay[name]
        return conv_block(x, state["decoder.%d.conv.conv.weight" % i], state["decoder.%d.conv.conv.bias" % i])

    outputs = {}
    x = features[-1]
    for i in range(4, -1, -1):
        x = cb(("upconv", i, 0), x)
        x = upsample_nearest2(x)
        if use_skips and i > 0:
            x = torch.cat([x, features[i - 1]], 1)
        x = cb(("upconv", i, 1), x)
        if i in scales:
            if pre_disp:
                outputs[("disp", i)] = x
            else:
                j = lay[("dispconv", i)]
                outputs[("disp", i)] = torch.sigmoid(conv3x3_reflect(
                    x, state["decoder.%d.conv.weight" % j], state["decoder.%d.conv.bias" % j]))
    return outputs


# --------------------------------------------------------------------------
# a4  PoseDecoder                        reference networks/pose_decoder.py:14-54
# --------------------------------------------------------------------------
def pose_decoder_forward(state, input_features, num_frames_to_predict_for, kinks=None):
    """`kinks`: oracle.kinks.ForcedKinks to impose recorded ReLU decisions (default: plain F.relu)."""
    relu = kinks.relu if kinks is not None else F.relu
    last = [f[-1] for f in input_features]
    cat = torch.cat([relu(F.conv2d(f, state["net.0.weight"], state["net.0.bias"])) for f in last], 1)
    out = relu(F.conv2d(cat, state["net.1.weight"], state["net.1.bias"], padding=1))
    out = relu(F.conv2d(out, state["net.2.weight"], state["net.2.bias"], padding=1))
    out = F.conv2d(out, state["net.3.weight"], state["net.3.bias"])
    out = out.mean(3).mean(2)
    out = 0.01 * out.view(-1, num_frames_to_predict_for, 1, 6)
    return out[..., :3], out[..., 3:]


# --------------------------------------------------------------------------
# a16 predict_poses (pairs mode, separate_resnet)           trainer.py:378-442
# --------------------------------------------------------------------------
def predict_poses(inputs, pose_encoder_fn, pose_decoder_fn):
    outputs = {}
    for f in (-1, 1):
        pair = [inputs[("color_aug", f, 0)], inputs[("color_aug", 0, 0)]] if f < 0 else \
               [inputs[("color_aug", 0, 0)], inputs[("color_aug", f, 0)]]
        feats = [pose_encoder_fn(torch.cat(pair, 1))]
        axisangle, translation = pose_decoder_fn(feats)
        outputs[("axisangle", 0, f)] = axisangle
        outputs[("translation", 0, f)] = translation
        outputs[("cam_T_cam", 0, f)] = transformation_from_parameters(
            axisangle[:, 0], translation[:, 0], invert=(f < 0))
    return outputs


# --------------------------------------------------------------------------
# a16, the other pose modes                      networks/pose_cnn.py:14-53, trainer.py:378-442
# --------------------------------------------------------------------------
def pose_cnn_forward(state, x, num_input_frames):
    """networks/pose_cnn.py:40-53: seven stride-2 convolutions (7x7, 5x5, 3x3 x 5) + ReLU, a 1x1 head, spatial mean, x 0.01."""
    pads = (3, 2, 1, 1, 1, 1, 1)
    out = x
    for i in range(7):
        out = F.relu(F.conv2d(out, state["net.%d.weight" % i], state["net.%d.bias" % i], stride=2, padding=pads[i]))
    out = F.conv2d(out, state["pose_conv.weight"], state["pose_conv.bias"])
    out = out.mean(3).mean(2)
    out = 0.01 * out.view(-1, num_input_frames - 1, 1, 6)
    return out[..., :3], out[..., 3:]


def predict_poses_modes(inputs, features, pose_model_type, pose_model_input, frame_ids, pose_encoder_fn, pose_fn):
    """trainer.py:378-442 for every pose_model_type / pose_model_input.  `features`: {frame id: encoder feature list} ("shared").
    `pose_fn`: the pose network -- PoseDecoder on a list of feature lists ("separate_resnet", "shared") or PoseCNN on a
    tensor; `pose_encoder_fn`: the pose encoder ("separate_resnet").  Frames are always passed in temporal order in pairs mode
    (:396-399) and the pose to a PAST frame is inverted there (:417-419); "all" mode predicts every pose in one pass, in
    frame_ids order, without inversion (:435-440)."""
    outputs = {}
    if pose_model_input == "pairs":
        feats = {f: features[f] for f in frame_ids} if pose_model_type == "shared" else {f: inputs[("color_aug", f, 0)] for f in (-1, 0, 1)}
        for f in (-1, 1):
            pose_inputs = [feats[f], feats[0]] if f < 0 else [feats[0], feats[f]]
            if pose_model_type == "separate_resnet":
                pose_inputs = [pose_encoder_fn(torch.cat(pose_inputs, 1))]
            elif pose_model_type == "posecnn":
                pose_inputs = torch.cat(pose_inputs, 1)
            axisangle, translation = pose_fn(pose_inputs)
            outputs[("axisangle", 0, f)] = axisangle
            outputs[("translation", 0, f)] = translation
            outputs[("cam_T_cam", 0, f)] = transformation_from_parameters(axisangle[:, 0], translation[:, 0], invert=(f < 0))
        return outputs
    ids = [i for i in frame_ids if i != "s"]
    if pose_model_type in ("separate_resnet", "posecnn"):
        pose_inputs = torch.cat([inputs[("color_aug", i, 0)] for i in ids], 1)
        if pose_model_type == "separate_resnet":
            pose_inputs = [pose_encoder_fn(pose_inputs)]
    else:
        pose_inputs = [features[i] for i in ids]
    axisangle, translation = pose_fn(pose_inputs)
    for i, f in enumerate(frame_ids[1:]):
        if f != "s":
            outputs[("axisangle", 0, f)] = axisangle
            outputs[("translation", 0, f)] = translation
            outputs[("cam_T_cam", 0, f)] = transformation_from_parameters(axisangle[:, i], translation[:, i])
    return outputs


# --------------------------------------------------------------------------
# synthetic KITTI-shaped batch     datasets/mono_dataset.py:122-183
# --------------------------------------------------------------------------
KITTI_K = np.array([[0.58, 0, 0.5, 0],
                    [0, 1.92, 0.5, 0],
                    [0, 0, 1, 0],
                    [0, 0, 0, 1]], dtype=np.float32)       # datasets/kitti_dataset.py:25-28


def synthetic_inputs(batch, height, width, num_scales=4, frame_ids=(0, -1, 1), seed=0, smooth=True):
    g = torch.Generator().manual_seed(seed)
    inputs = {}
    for f in frame_ids:
        base = torch.rand(batch, 3, height, width, generator=g)
        if smooth:
            base = F.avg_pool2d(F.pad(base, (2, 2, 2, 2), mode="reflect"), 5, 1)
        for s in range(num_scales):
            img = base if s == 0 else F.avg_pool2d(base, 2 ** s)
            inputs[("color", f, s)] = img.contiguous()
            inputs[("color_aug", f, s)] = img.contiguous()
    for s in range(num_scales):
        K = KITTI_K.copy()
        K[0, :] *= width // (2 ** s)
        K[1, :] *= height // (2 ** s)
        inv_K = np.linalg.pinv(K)
        inputs[("K", s)] = torch.from_numpy(K).unsqueeze(0).repeat(batch, 1, 1)
        inputs[("inv_K", s)] = torch.from_numpy(inv_K).unsqueeze(0).repeat(batch, 1, 1)
    return inputs


def tiebreak_noise(batch, height, width, num_scales=4, seed=1234):
    """One CPU randn(B,2,H,W) per scale, in scale order (trainer.py:594-595)."""
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(batch, 2, height, width, generator=g) for _ in range(num_scales)]
