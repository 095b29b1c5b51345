"""CPU restatement of the reference's `Fusion_v3` attention-fusion front-end (BASELINE configs[4]).

TEST INFRASTRUCTURE (see oracle/__init__.py).  Functional, over a plain state dict with the reference's key names
(`fusion_block_{1..4}.resConfUnit{1,2,3}.atten{1,2}.{rel_h,rel_w,key_conv,query_conv,value_conv}...`), written from the
formulas with explicit shifted slices instead of `unfold` / `einsum`.  Pinned by tests/golden/fusion_v3.npz, which
tests/golden/make_golden_r2.py generates by running the reference's networks/fusion_v2.py (loaded by file path).

Reference lines followed: networks/fusion_v2.py:46-98 (AttentionConv), :101-137 (ResidualAttentionUnit), :226-236
(UpscalePS), :279-320 (FeatureFusionBlock_v3), :323-363 (Fusion_v3); layers.py:121-136 (Conv3x3).
"""
import torch
import torch.nn.functional as F


def attention_conv(x, st, p):
    """networks/fusion_v2.py:67-96 with kernel_size 3, stride 1, padding 1, groups 1, bias=True.

    q = Wq x + bq at the pixel; k, v = 1x1 convolutions of the ZERO-PADDED input, so an out-of-image tap carries
    k = bk (+ rel), v = bv.  First half of the channels: k += rel_h[dy]; second half: k += rel_w[dx].
    Per pixel and channel: softmax over the 9 taps of q * k, output = sum_t softmax_t * v_t."""
    B, C, H, W = x.shape
    wq, bq = st[p + "query_conv.weight"].reshape(C, C), st[p + "query_conv.bias"]
    wk, bk = st[p + "key_conv.weight"].reshape(C, C), st[p + "key_conv.bias"]
    wv, bv = st[p + "value_conv.weight"].reshape(C, C), st[p + "value_conv.bias"]
    rel_h, rel_w = st[p + "rel_h"].reshape(3), st[p + "rel_w"].reshape(3)
    xp = F.pad(x, (1, 1, 1, 1))
    q = torch.einsum("oc,bchw->bohw", wq, x) + bq.view(1, C, 1, 1)
    k = torch.einsum("oc,bchw->bohw", wk, xp) + bk.view(1, C, 1, 1)
    v = torch.einsum("oc,bchw->bohw", wv, xp) + bv.view(1, C, 1, 1)
    half = C // 2
    logits, vals = [], []
    for dy in range(3):
        for dx in range(3):
            kt = k[:, :, dy:dy + H, dx:dx + W]
            rel = torch.cat([rel_h[dy].expand(half), rel_w[dx].expand(C - half)]).view(1, C, 1, 1)
            logits.append(q * (kt + rel))
            vals.append(v[:, :, dy:dy + H, dx:dx + W])
    logits = torch.stack(logits, -1)                       # (B,C,H,W,9), tap index dy*3+dx as in the reference's view
    a = torch.softmax(logits, dim=-1)
    return (a * torch.stack(vals, -1)).sum(-1)


def residual_attention_unit(x, st, p):
    """networks/fusion_v2.py:101-137.  `nn.ReLU(inplace=True)` rewrites the unit's input, so the skip connection adds
    relu(x), not x."""
    r = F.relu(x)
    out = attention_conv(r, st, p + "atten1.")
    out = attention_conv(F.relu(out), st, p + "atten2.")
    return out + r


def residual_conv_unit(x, st, p):
    """networks/fusion_v2.py:11-43 (`attention=False`): conv2(relu(conv1(relu(x)))) + relu(x) (in-place ReLU, as above)."""
    r = F.relu(x)
    out = F.conv2d(r, st[p + "conv1.weight"], st[p + "conv1.bias"], padding=1)
    out = F.conv2d(F.relu(out), st[p + "conv2.weight"], st[p + "conv2.bias"], padding=1)
    return out + r


def conv3x3_reflect(x, w, b):
    """layers.py:121-136 (ReflectionPad2d(1) + Conv2d(3))."""
    return F.conv2d(F.pad(x, (1, 1, 1, 1), mode="reflect"), w, b)


def upscale_ps(x, w, b, scale=2):
    """networks/fusion_v2.py:226-236: PixelShuffle(tanh(conv3x3 zero-pad))."""
    y = torch.tanh(F.conv2d(x, w, b, padding=1))
    B, C, H, W = y.shape
    c = C // (scale * scale)
    y = y.view(B, c, scale, scale, H, W).permute(0, 1, 4, 2, 5, 3)      # out[b,c,h*s+i,w*s+j] = y[b,c*s*s+i*s+j,h,w]
    return y.reshape(B, c, H * scale, W * scale)


def upscale_ps_shuffle_only(y, scale=2):
    """nn.PixelShuffle(scale) alone (the second half of UpscalePS)."""
    B, C, H, W = y.shape
    c = C // (scale * scale)
    return y.view(B, c, scale, scale, H, W).permute(0, 1, 4, 2, 5, 3).reshape(B, c, H * scale, W * scale)


def fusion_block_v3(dt, upt, dt_1, dt_2, st, p, init_scale, attention=True):
    """networks/fusion_v2.py:304-320 (units per :294-302)."""
    residual_attention_unit = globals()["residual_attention_unit"] if attention else residual_conv_unit
    if init_scale:
        dt_upt = F.conv2d(dt, st[p + "conv_1.weight"], st[p + "conv_1.bias"], padding=1)
    else:
        dt_upt = torch.cat([dt, upt], 1)
    context = torch.cat([dt_1, dt_2], 1)
    out = torch.cat([residual_attention_unit(dt_upt, st, p + "resConfUnit1."),
                     residual_attention_unit(context, st, p + "resConfUnit2.")], 1)
    out = residual_attention_unit(out, st, p + "resConfUnit3.")
    depth = conv3x3_reflect(out, st[p + "conv3x3.conv.weight"], st[p + "conv3x3.conv.bias"])
    up = upscale_ps(out, st[p + "upscale.conv.weight"], st[p + "upscale.conv.bias"])
    return depth, up


def fusion_v3_forward(st, depth_dec_outputs, attention=True):
    """networks/fusion_v2.py:335-363.  Every decoder output (3B, 1, h, w) is split into three chunks along the batch;
    chunk 0 plays `dt`, chunks 1 and 2 the context (the caller stacks frames [-2, -1, 0], trainer_fusion_v3.py:319,
    so chunk 0 is frame -2: kept as is).  No sigmoid on the outputs."""
    cur, t1, t2 = {}, {}, {}
    for k, v in depth_dec_outputs.items():
        n = v.shape[0] // 3
        cur[k], t1[k], t2[k] = v[:n], v[n:2 * n], v[2 * n:3 * n]
    outputs, up = {}, None
    for i, s in enumerate((3, 2, 1, 0)):
        outputs[("disp", s)], up = fusion_block_v3(cur[("disp", s)], up, t1[("disp", s)], t2[("disp", s)], st,
                                                   "fusion_block_%d." % (i + 1), init_scale=(i == 0), attention=attention)
    return outputs


def fusion_v3_layout():
    """(key, shape) of the reference state dict, in registration order (210 tensors, 1,672 parameters)."""
    out = []
    for i in range(1, 5):
        p = "fusion_block_%d." % i
        if i == 1:
            out += [(p + "conv_1.weight", (2, 1, 3, 3)), (p + "conv_1.bias", (2,))]
        for u, c in (("resConfUnit1.", 2), ("resConfUnit2.", 2), ("resConfUnit3.", 4)):
            for a in ("atten1.", "atten2."):
                q = p + u + a
                out += [(q + "rel_h", (1, 1, 1, 3, 1)), (q + "rel_w", (1, 1, 1, 1, 3))]
                for n in ("key_conv", "query_conv", "value_conv"):
                    out += [(q + n + ".weight", (c, c, 1, 1)), (q + n + ".bias", (c,))]
        out += [(p + "conv3x3.conv.weight", (1, 4, 3, 3)), (p + "conv3x3.conv.bias", (1,)),
                (p + "upscale.conv.weight", (4, 4, 3, 3)), (p + "upscale.conv.bias", (4,))]
    return out
