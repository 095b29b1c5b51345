"""f1: the Fusion_v3 front-end on the HIP path (dc_attnconv_fwd/bwd, the fused conv block with tanh) against the
reference's fixtures (tests/golden/fusion_v3.npz) and the CPU oracle (oracle/fusion_ref.py)."""
import ctypes

import pytest
import torch

from oracle import fusion_ref as FR
import make_golden_r2 as MG2
from helpers import T, close, rel_l2
from test_fusion_oracle import attn_state, fusion_state

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _order(st, p=""):
    """oracle-keyed parameters -> the (rel_h, rel_w, wk, bk, wq, bq, wv, bv) tuple of depthcore.ops"""
    return tuple(st[p + k] for k in ("rel_h", "rel_w", "key_conv.weight", "key_conv.bias", "query_conv.weight",
                                     "query_conv.bias", "value_conv.weight", "value_conv.bias"))


def test_attention_conv_kernel_vs_reference_fixture(golden):
    """One bare AttentionConv (4 channels, 9x14 map: ragged against the 16x32 / 6x30 tiles) through the C ABI."""
    from depthcore import _lib, ops
    L = _lib.lib()
    g = golden["fusion_v3"]
    st = attn_state()
    ps = [t.to(DEV).contiguous() for t in _order(st)]
    x, _ = MG2.attn_case()
    x = x.to(DEV)
    B, C, H, W = x.shape
    xm, _ = ops._attn_map([x], [ops.PLAIN], H, W)
    ap = ops._attn_params(ps)
    y = torch.empty_like(x)
    _lib.check(L.dc_attnconv_fwd(ctypes.byref(xm), ctypes.byref(ap), None, _lib.ptr(y), B, C, H, W, 0, 0, _lib.stream(x)), "fwd")
    close(y, g["ac_y"], rtol=1e-4, atol=1e-5)
    gy = T(g["ac_cot"]).to(DEV)
    dx = torch.full_like(x, float("nan"))
    dxm, _ = ops._attn_map([dx], [ops.PLAIN], H, W)
    dp = torch.empty(L.dc_attnconv_param_count(C), device=DEV)
    ws = torch.empty(L.dc_attnconv_bwd_workspace(B, C, H, W), dtype=torch.uint8, device=DEV)
    _lib.check(L.dc_attnconv_bwd(ctypes.byref(xm), ctypes.byref(ap), None, _lib.ptr(gy), ctypes.byref(dxm), None, None, _lib.ptr(dp),
                                 ws.data_ptr(), B, C, H, W, 0, 0, _lib.stream(x)), "bwd")
    close(dx, g["ac_gx"], rtol=1e-3, atol=1e-5)
    grads = ops._attn_param_grads(dp, C, ps)
    for k, got in zip(("rel_h", "rel_w", "key_conv.weight", "key_conv.bias", "query_conv.weight", "query_conv.bias",
                       "value_conv.weight", "value_conv.bias"), grads):
        close(got, g["ac_g_" + k], rtol=1e-3, atol=5e-5, msg=k)


@pytest.mark.parametrize("case", ["c2_plain_ps2", "c4_two_plain", "c2_two_chunks"])
def test_residual_attention_unit_vs_oracle(case):
    """The unit with its gathered inputs: [plain 1ch, pixel-shuffled 4ch] (block 2-4 `cat([dt, upt])`), two 2-channel
    tensors (`cat([unit1, unit2])`), two chunks of one tensor (`cat([dt_1, dt_2])`); forward and every gradient."""
    from depthcore import ops
    g = torch.Generator().manual_seed(5)
    B, H, W = 2, 20, 34
    if case == "c2_plain_ps2":
        srcs = [torch.randn(B, 1, H, W, generator=g), torch.randn(B, 4, H // 2, W // 2, generator=g)]
        kinds = [ops.PLAIN, ops.PIXEL_SHUFFLE2]
        C = 2
    elif case == "c4_two_plain":
        srcs = [torch.randn(B, 2, H, W, generator=g), torch.randn(B, 2, H, W, generator=g)]
        kinds = [ops.PLAIN, ops.PLAIN]
        C = 4
    else:
        full = torch.randn(3 * B, 1, H, W, generator=g)
        srcs = [full[B:2 * B], full[2 * B:]]
        kinds = [ops.PLAIN, ops.PLAIN]
        C = 2
    st = {}
    for a in ("atten1.", "atten2."):
        for k, v in attn_state(C, seed=11 + len(st)).items():
            st[a + k] = v
    cot = torch.randn(B, C, H, W, generator=g)
    # oracle
    so = [t.clone().requires_grad_() for t in srcs]
    po = {k: v.clone().requires_grad_() for k, v in st.items()}
    parts = [FR.upscale_ps_shuffle_only(t) if k == ops.PIXEL_SHUFFLE2 else t for t, k in zip(so, kinds)]
    yo = FR.residual_attention_unit(torch.cat(parts, 1), po, "")
    go = torch.autograd.grad((yo * cot).sum(), so + list(po.values()))
    # HIP
    sh = [t.to(DEV).contiguous().requires_grad_() for t in srcs]
    ph = {k: v.to(DEV).requires_grad_() for k, v in st.items()}
    yh = ops.residual_attention_unit(sh, kinds, _order(ph, "atten1."), _order(ph, "atten2."))
    gh = torch.autograd.grad((yh * cot.to(DEV)).sum(), sh + list(ph.values()))
    close(yh, yo, rtol=1e-4, atol=1e-5)
    names = ["src%d" % i for i in range(len(srcs))] + list(st)
    scale = max(float(b.abs().max()) for b in go)
    for n, a, b in zip(names, gh, go):
        if "key_conv.bias" in n:      # analytically zero (softmax is invariant to a shift of the logits' k offset times q ... sum_t dlogit_t = 0)
            assert float(a.abs().max()) <= 1e-5 * scale and float(b.abs().max()) <= 1e-5 * scale, n
            continue
        assert rel_l2(a, b) < 2e-4, (n, rel_l2(a, b))


def test_fusion_v3_module_vs_reference_fixture(golden):
    import networks
    g = golden["fusion_v3"]
    fu = networks.Fusion_v3(attention=True).to(DEV)
    assert list(fu.state_dict().keys()) == list(g["keys"])
    fu.load_state_dict(fusion_state())
    inp, _ = MG2.fusion_inputs()
    inp = {k: v.to(DEV).requires_grad_() for k, v in inp.items()}
    o = fu(inp)
    tot = 0
    for s in range(4):
        close(o[("disp", s)], g["disp%d" % s], rtol=1e-3, atol=1e-5, msg="disp%d" % s)
        tot = tot + (o[("disp", s)] * T(g["cot%d" % s]).to(DEV)).sum()
    params = dict(fu.named_parameters())
    names = list(params)
    gr = torch.autograd.grad(tot, [inp[("disp", s)] for s in range(4)] + [params[k] for k in names], allow_unused=True)
    for s in range(4):
        assert rel_l2(gr[s], g["gin%d" % s]) < 1e-3, ("gin%d" % s, rel_l2(gr[s], g["gin%d" % s]))
    for j, k in enumerate(names):
        want = g["g_" + k]
        if want.size == 0:
            assert gr[4 + j] is None                     # fusion_block_4.upscale: unused in the reference as well
        else:
            close(gr[4 + j], want, rtol=2e-3, atol=3e-4, msg=k)


def test_fusion_v3_without_attention_vs_reference_fixture(golden):
    """`Fusion_v3(attention=False)` (reference --disable_attention; ResidualConvUnit, networks/fusion_v2.py:11-43,294-302):
    both convolutions of every unit on the fused conv block, against the reference's own outputs and gradients."""
    import networks
    from test_fusion_oracle import noattn_state
    g = golden["fusion_v3_noattn"]
    fu = networks.Fusion_v3(attention=False).to(DEV)
    assert list(fu.state_dict().keys()) == list(g["keys"])
    fu.load_state_dict(noattn_state(g["keys"], g["shapes"]))
    inp, _ = MG2.fusion_inputs()
    inp = {k: v.to(DEV).requires_grad_() for k, v in inp.items()}
    o = fu(inp)
    tot = 0
    for s in range(4):
        close(o[("disp", s)], g["disp%d" % s], rtol=1e-3, atol=1e-4, msg="disp%d" % s)
        tot = tot + (o[("disp", s)] * T(g["cot%d" % s]).to(DEV)).sum()
    params = dict(fu.named_parameters())
    names = list(params)
    gr = torch.autograd.grad(tot, [inp[("disp", s)] for s in range(4)] + [params[k] for k in names], allow_unused=True)
    for s in range(4):
        assert rel_l2(gr[s], g["gin%d" % s]) < 1e-3, ("gin%d" % s, rel_l2(gr[s], g["gin%d" % s]))
    for j, k in enumerate(names):
        want = g["g_" + k]
        if want.size == 0:
            assert gr[4 + j] is None
        else:
            assert rel_l2(gr[4 + j], want) < 1e-3, (k, rel_l2(gr[4 + j], want))


def test_fusion_v3_full_size_properties():
    """BASELINE configs[4] size (B=12, 192x640, 3 stacked frames): deterministic (bitwise) forward and gradients, finite."""
    import networks
    torch.manual_seed(0)
    fu = networks.Fusion_v3().to(DEV)
    g = torch.Generator(device=DEV).manual_seed(1)
    inp = {("disp", s): torch.rand(36, 1, 192 >> s, 640 >> s, device=DEV, generator=g).requires_grad_() for s in range(4)}
    res = []
    for _ in range(2):
        o = fu(inp)
        tot = sum(o[("disp", s)].square().sum() for s in range(4))
        gr = torch.autograd.grad(tot, list(inp.values()) + [p for n, p in fu.named_parameters() if "fusion_block_4.upscale" not in n])
        res.append([o[("disp", s)].clone() for s in range(4)] + [x.clone() for x in gr])
    for a, b in zip(*res):
        assert torch.equal(a, b) and torch.isfinite(a).all()
    assert o[("disp", 0)].shape == (12, 1, 192, 640)
